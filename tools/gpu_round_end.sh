#!/bin/bash
# the measurements the round's documents quote, in one call: bench line, rocprofv3 stats + census of the step, per-wave
# counters, backward stamps, LDS atomic probe.  Outputs under gpurun_out/end/ (copy what is quoted into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/end
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -n 5 $O/bench.err; exit 1; }
tail -c 400 $O/bench_line.json; echo
tools/gpu_profile.sh end_stats --steps 40 --warmup 5 --no-cpu-baseline > $O/profile.txt 2>&1 || exit 1
python3 tools/step_census.py $(find gpurun_out/end_stats -name "*kernel_trace.csv" | head -1) > $O/step_census.txt || exit 1
head -n 3 $O/step_census.txt
tools/gpu_pmc_all.sh end_pmc > $O/pmc.log 2>&1 || exit 1
python3 tools/pmc_waves.py gpurun_out/end_pmc > $O/pmc_waves.txt
tools/gpu_bwd_stamps.sh > /dev/null 2>&1; cp gpurun_out/bwd_stamps.txt $O/bwd_stamps.txt
timeout -k 10 100 tools/probe/lds_atomic > $O/lds_atomic.txt 2>&1
echo done
