#!/bin/bash
# GPU parity suite, then the end-to-end bench (no CPU baseline) -- the quick loop while tuning the training step.
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 12 gpurun_out/pytest_gpu.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/step_bench.log 2>&1 || { tail -n 20 gpurun_out/step_bench.log; exit 1; }
tail -n 1 gpurun_out/step_bench.log | cut -c1-400
