#!/bin/bash
# Round 4: census of the headline step + the opt-in combined-backward mode timed at the headline shape, L = 200 and configs[3]
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4
mkdir -p $O
bash tools/gpu_r4_census.sh head > $O/census_head.log 2>&1; tail -3 $O/census_head.log
for spec in "head:" "l200:--seq-len 200" "cfg4:--config cfg4"; do
  tag=${spec%%:*}; args=${spec#*:}
  for mode in "" "--combined-backward"; do
    m=two; [ -n "$mode" ] && m=one
    timeout -k 10 300 python bench.py $args $mode --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20 --steps 30 > $O/bench_${tag}_${m}walk.json 2> $O/bench_${tag}_${m}walk.err || tail -3 $O/bench_${tag}_${m}walk.err
    python3 -c "import json,sys; d=json.loads(open('$O/bench_${tag}_${m}walk.json').read().strip().splitlines()[-1]); print('$tag', '$m', 'walk(s): ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'losses', d['config']['final_losses'])"
  done
done
