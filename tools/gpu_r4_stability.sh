#!/bin/bash
# Round 4: graph-replay training runs with the losses and parameter finiteness checked every step (tools/nan_probe_graph.py),
# both backward modes; then the embedding / AcBERT4Rec tests and the configs[4] step after the hot-row change
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_embed.py tests/test_hip_bert4rec.py tests/test_hip_combined.py -x -q > $O/pytest_embed.log 2>&1; tail -n 3 $O/pytest_embed.log
{
for spec in "headline:3000:" "l200:600:--seq-len 200" "cfg4:600:--config cfg4" "cfg5:200:--config cfg5"; do
  tag=${spec%%:*}; rest=${spec#*:}; n=${rest%%:*}; args=${rest#*:}
  for mode in "" "--combined-backward"; do
    echo "== $tag ${mode:-two walks}"
    PROBE_STEPS=$n timeout -k 10 400 python tools/nan_probe_graph.py $args $mode 2>/dev/null | awk 'NR<=3 || /clean/'
  done
done
} > $O/stability_runs.txt 2>&1
cat $O/stability_runs.txt
timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20 --steps 20 > $O/bench4_cfg5.json 2> $O/bench4_cfg5.err
python3 -c "import json; d=json.loads(open('$O/bench4_cfg5.json').read().strip().splitlines()[-1]); print('cfg5 ms/step', d['ms_per_step'], d['config']['final_losses'])"
