#!/bin/bash
# whole-step A/B on one box (headline shape, short): usage tools/gpu_r4_ab.sh <rounds> <lib|default> ...
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r4
rounds=$1; shift
for r in $(seq "$rounds"); do
  for lib in "$@"; do
    if [ "$lib" != "default" ]; then export ACATTN_LIB=$PWD/$lib; else unset ACATTN_LIB; fi
    timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 10 > gpurun_out/r4/ab.log 2> gpurun_out/r4/ab.err || { echo "$lib failed"; tail -n 3 gpurun_out/r4/ab.err; exit 1; }
    echo "$lib: $(tail -n 1 gpurun_out/r4/ab.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"])')"
  done
done
