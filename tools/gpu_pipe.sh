#!/bin/bash
# Pipelined forward kernel: parity against the general kernel, then the kernel micro-bench with and without it.
set -u
mkdir -p gpurun_out
ACATTN_PIPE=1 timeout -k 10 300 python -m pytest tests/test_hip_forward.py -m gpu -x -q -k "fast_training_kernel" > gpurun_out/pipe_pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/pipe_pytest.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for items in 0 2 4; do
  if [ $items -eq 0 ]; then export ACATTN_PIPE=0; else export ACATTN_PIPE=1 ACATTN_PIPE_ITEMS=$items; fi
  timeout -k 10 200 python bench.py --kernel-only > gpurun_out/pipe_bench_$items.log 2>&1 || { echo "bench $items failed"; tail -n 5 gpurun_out/pipe_bench_$items.log; exit 1; }
  echo "items=$items: $(tail -n 1 gpurun_out/pipe_bench_$items.log)"
done
