#!/bin/bash
# round 3, final measurements: the GPU suite, the default bench line, then tools/gpu_r3_prof.sh (counters + kernel stats),
# then the per-step kernel stats of configs[3] / [4] and of L = 200
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r3f
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest_gpu.log 2>&1
rc=$?
tail -n 4 gpurun_out/r3f/pytest_gpu.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r3f/bench_line.txt 2> gpurun_out/r3f/bench_err.txt
echo "bench rc=$?"
grep "^{" gpurun_out/r3f/bench_line.txt | cut -c1-400
