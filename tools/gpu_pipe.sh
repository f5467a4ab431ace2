#!/bin/bash
# DMA-staged forward kernel: parity against the general kernel, then the kernel micro-bench with and without it.
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_forward.py -m gpu -x -q > gpurun_out/dma_pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/dma_pytest.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 0 1; do
  ACATTN_DMA=$v timeout -k 10 200 python bench.py --kernel-only > gpurun_out/dma_bench_$v.log 2>&1 || { echo "bench $v failed"; tail -n 5 gpurun_out/dma_bench_$v.log; exit 1; }
  echo "dma=$v: $(tail -n 1 gpurun_out/dma_bench_$v.log | cut -c1-420)"
done
