#!/bin/bash
# whole-step A/B on one box: bench.py per library build, alternating (ACATTN_LIB; "default" = the in-tree build)
# usage: tools/gpu_ab.sh <rounds> <lib|default> [<lib|default> ...]
rounds=$1; shift
for r in $(seq "$rounds"); do
  for lib in "$@"; do
    if [ "$lib" != "default" ]; then export ACATTN_LIB=$PWD/$lib; else unset ACATTN_LIB; fi
    timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/ab_bench.log 2> gpurun_out/ab_bench.err || { echo "$lib failed"; tail -n 3 gpurun_out/ab_bench.err; exit 1; }
    echo "$lib: $(tail -n 1 gpurun_out/ab_bench.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
