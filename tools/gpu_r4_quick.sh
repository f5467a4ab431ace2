#!/bin/bash
# quick check after a small kernel change: the loss / CE / model suites, then the headline and L = 200 step times
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_ce.py tests/test_hip_backward.py tests/test_hip_model_surface.py tests/test_hip_combined.py tests/test_hip_bert4rec.py -x -q 2>&1 | tail -2 || exit 1
for spec in "head:" "l200:--seq-len 200"; do tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 10 --steps 100 --warmup 10 > $O/q.json 2> $O/q.err || tail -3 $O/q.err
  python3 -c "import json; d=json.loads(open('$O/q.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'], d['config']['final_losses'])"; done
