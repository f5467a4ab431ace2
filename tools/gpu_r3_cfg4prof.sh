#!/bin/bash
# cfg4 (and cfg5) step profile: rocprofv3 kernel stats of the graph-replayed step
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3q
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg4 cfg5; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r3q/$cfg" -o $cfg --output-format csv -- python3 "$R/bench.py" --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --kernel-kinds ragged --kernel-iters 20 > "$R/gpurun_out/r3q/$cfg.log" 2>&1
done
echo done
