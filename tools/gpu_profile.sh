#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py run; summaries land in gpurun_out/<name>/
set -u
name=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o $name -- python3 $R/bench.py "$@" > $R/gpurun_out/$name/run.log 2>&1
echo "rocprof rc=$?"
tail -n 3 $R/gpurun_out/$name/run.log
find $R/gpurun_out/$name -name "*kernel_stats.csv" | head -1 | xargs -r head -n 40
