#!/bin/bash
# Round 4: the cross-entropy files under profiles/ -- accuracy of the split products against fp64 next to the fp32 kernels',
# per-phase stamps (needs tools/gpu_ce6_stamps.sh build beforehand), fp32 vs split step time on one box.
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r4; mkdir -p $O; export PYTHONPATH=$PWD
timeout -k 10 300 python tools/dbg/ce_split_dbg.py 2>&1 | grep -v amdgpu.ids > $O/ce_split_accuracy.txt; tail -3 $O/ce_split_accuracy.txt
bash tools/gpu_ce6_stamps.sh > /dev/null; cp gpurun_out/ce6_stamps.txt $O/ce6_stamps.txt; head -3 $O/ce6_stamps.txt
for m in fp32 default; do
  ACATTN_CE_PRODUCTS=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 10 --steps 50 > $O/ce_ab_$m.json 2> $O/ce_ab_$m.err || tail -3 $O/ce_ab_$m.err
  python3 -c "import json; d=json.loads(open('$O/ce_ab_$m.json').read().strip().splitlines()[-1]); print('ACATTN_CE_PRODUCTS=$m', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'final_losses', d['config']['final_losses'], d['config']['ce_products'])" | tee -a $O/ce_ab.txt
done
