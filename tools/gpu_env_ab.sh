#!/bin/bash
# whole-step A/B of environment settings on one box: tools/gpu_env_ab.sh <rounds> "VAR=1 VAR2=x" "" ...  ("" = defaults)
rounds=$1; shift
for r in $(seq "$rounds"); do
  for setting in "$@"; do
    env $setting timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/ab_bench.log 2> gpurun_out/ab_bench.err || { echo "[$setting] failed"; tail -n 3 gpurun_out/ab_bench.err; exit 1; }
    echo "[$setting]: $(tail -n 1 gpurun_out/ab_bench.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
