#!/bin/bash
# Run on the GPU box via gpurun: GPU parity tests, then the kernel roofline micro-bench.
# A timed-out / killed step stops the chain (never start another GPU step after a hang).
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 30 gpurun_out/pytest_gpu.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then echo "pytest timed out or was killed: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --kernel-only > gpurun_out/kernel_bench.log 2>&1
rc2=$?
cat gpurun_out/kernel_bench.log | tail -n 5
echo "kernel bench rc=$rc2"
exit $rc
