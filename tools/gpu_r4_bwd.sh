#!/bin/bash
# Round 4: GPU suite, then step timings of the named shapes (two-walk) after the backward changes
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu2.log 2>&1
rc=$?
tail -n 4 $O/pytest_gpu2.log
if [ $rc -ne 0 ]; then grep -n "^E \|Error" $O/pytest_gpu2.log | head -20; exit $rc; fi
for spec in "head:" "l200:--seq-len 200" "cfg4:--config cfg4" "cfg5:--config cfg5"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20 --steps 30 > $O/bench2_${tag}.json 2> $O/bench2_${tag}.err || tail -3 $O/bench2_${tag}.err
  python3 -c "import json,sys; d=json.loads(open('$O/bench2_${tag}.json').read().strip().splitlines()[-1]); print('$tag', 'ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'losses', d['config']['final_losses'])"
done
