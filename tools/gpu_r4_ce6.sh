#!/bin/bash
# Round 4: the split-product cross-entropy sweeps -- tests, phase stamps (after tools/gpu_ce6_stamps.sh build), step time.
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r4; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_hip_ce.py -x -q 2>&1 | tail -3 || exit 1
[ -f tools/tmp_libs/libacattn_ce6stamps.so ] && bash tools/gpu_ce6_stamps.sh
for m in ${CE6_MODES:-default}; do
  ACATTN_CE_PRODUCTS=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 10 --steps 50 > $O/ce6_b.json 2> $O/ce6_b.err || tail -3 $O/ce6_b.err
  python3 -c "import json; d=json.loads(open('$O/ce6_b.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['value'], d['config']['final_losses'])"
done
