#!/bin/bash
# whole-step A/B of an environment switch on one box: usage tools/gpu_r4_env_ab.sh <rounds> <VAR=value|none> ...
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r4
rounds=$1; shift
for r in $(seq "$rounds"); do
  for kv in "$@"; do
    ( [ "$kv" != "none" ] && export "$kv"
      timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 10 > gpurun_out/r4/ab.log 2> gpurun_out/r4/ab.err || { echo "$kv failed"; tail -n 3 gpurun_out/r4/ab.err; exit 1; }
      echo "$kv: $(tail -n 1 gpurun_out/r4/ab.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"], d["config"]["final_losses"])')" )
  done
done
