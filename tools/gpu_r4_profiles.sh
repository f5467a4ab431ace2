#!/bin/bash
# Round 4: the profiles the documents quote -- per-step census + rocprofv3 kernel stats of the four named shapes (timed,
# pruned-schedule steps), the same for the one-walk mode at L = 200, the forward kernel alone, its PMC passes.
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
O=gpurun_out/r4
mkdir -p $O
bash tools/gpu_r4_census.sh head > $O/c_head.log 2>&1; tail -2 $O/c_head.log | head -1
bash tools/gpu_r4_census.sh l200 --seq-len 200 > $O/c_l200.log 2>&1
bash tools/gpu_r4_census.sh cfg4 --config cfg4 > $O/c_cfg4.log 2>&1
bash tools/gpu_r4_census.sh cfg5 --config cfg5 > $O/c_cfg5.log 2>&1
bash tools/gpu_r4_census.sh l200_onewalk --seq-len 200 --combined-backward > $O/c_l200_1w.log 2>&1
bash tools/gpu_r4_census.sh cfg4_onewalk --config cfg4 --combined-backward > $O/c_cfg4_1w.log 2>&1
for t in head l200 cfg4 cfg5 l200_onewalk cfg4_onewalk; do echo "== $t"; head -1 $O/census_$t/census.txt; done
# forward kernel alone (ragged + full + spatial), rocprofv3 stats
mkdir -p $O/fwd_only
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/fwd_only -o fwd -- python3 $R/bench.py --kernel-only --kernel-iters 300 > $R/$O/fwd_only/run.log 2>&1 )
cp $(find $O/fwd_only -name "*kernel_stats.csv" | head -1) $O/fwd_only/kernel_stats.csv; head -5 $O/fwd_only/kernel_stats.csv | cut -c1-160
# PMC passes of the forward kernel (ragged + spatial; then full length)
KINDS=ragged,spatial bash tools/gpu_pmc.sh r4/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
KINDS=full bash tools/gpu_pmc.sh r4/pmc_full > $O/pmc_full.log 2>&1; tail -2 $O/pmc_full.log
