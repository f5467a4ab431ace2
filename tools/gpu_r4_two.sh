#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_combined.py tests/test_hip_onehop.py -x -q > $O/pytest_two.log 2>&1
rc=$?; tail -n 4 $O/pytest_two.log
if [ $rc -ne 0 ]; then grep -n "^E \|Error" $O/pytest_two.log | head -20; exit $rc; fi
for spec in "head:" "l200:--seq-len 200" "cfg4:--config cfg4"; do
  tag=${spec%%:*}; args=${spec#*:}
  for mode in "" "--combined-backward"; do
    m=two; [ -n "$mode" ] && m=one
    timeout -k 10 300 python bench.py $args $mode --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20 --steps 30 > $O/bench3_${tag}_${m}walk.json 2> $O/bench3_${tag}_${m}walk.err || tail -3 $O/bench3_${tag}_${m}walk.err
    python3 -c "import json,sys; d=json.loads(open('$O/bench3_${tag}_${m}walk.json').read().strip().splitlines()[-1]); print('$tag', '$m', 'walk(s): ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'losses', d['config']['final_losses'])"
  done
done
