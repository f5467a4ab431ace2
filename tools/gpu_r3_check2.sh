#!/bin/bash
set -u
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest_gpu.log 2>&1
rc=$?
tail -n 6 gpurun_out/r3/pytest_gpu.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
ACATTN_ZERO_MEMSET=1 timeout -k 10 300 python tools/memset_graph_probe.py gpurun_out/r3/graph_memset > gpurun_out/r3/graph_memset.txt 2>&1
echo "memset probe rc=$?"; tail -n 12 gpurun_out/r3/graph_memset.txt
timeout -k 10 300 python tools/memset_graph_probe.py gpurun_out/r3/graph_kernel > gpurun_out/r3/graph_kernel.txt 2>&1
echo "kernel probe rc=$?"; tail -n 3 gpurun_out/r3/graph_kernel.txt
echo skip bench

exit $rc
