#!/bin/bash
# Round 4: rocprofv3 kernel trace of the TIMED (pruned-schedule, hipGraph) steps of the named shapes -> per-step census.
# usage: gpu_r4_census.sh <tag> <bench args...>   (--no-full-schedule is added: round 3's L = 200 census had picked up the
# reference-schedule steps that bench.py runs after the timed ones)
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4/census_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $tag -- python3 $R/bench.py "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-kinds ragged --kernel-iters 70 > $O/run.log 2>&1
rc=$?
echo "rocprof rc=$rc"
grep "^{" $O/run.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'fwd', d['roofline']['avg_launch_us'], d['roofline']['frac'])"
python3 $R/tools/step_census.py $(find $O -name "*kernel_trace.csv" | head -1) --sequence > $O/census.txt
head -n 30 $O/census.txt
cp $(find $O -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
exit $rc
