#!/bin/bash
# per-step kernel census of the timed steps: rocprofv3 kernel trace of bench.py + tools/step_census.py
# usage: tools/gpu_census.sh <name> [bench.py args]; env passes through (ACATTN_* measurement hooks)
set -u
name=${1:-census}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$name -o $name -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/$name/run.log 2>&1 || { echo "rocprof failed"; tail -n 5 $R/gpurun_out/$name/run.log; exit 1; }
python3 $R/tools/step_census.py $(find $R/gpurun_out/$name -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/$name/census.txt
head -n 24 $R/gpurun_out/$name/census.txt
