#!/bin/bash
# Cycles per phase of the split-product cross-entropy sweeps (acattn_ce_bf16.hip) at the benchmark shape.
# Build step (in the container, before gpurun):  tools/gpu_ce6_stamps.sh build
# On the GPU box:                                tools/gpu_ce6_stamps.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "${1:-}" = "build" ]; then
  set -e
  cd $R/ac_tsr_amd/csrc
  make -j8 > /dev/null
  mkdir -p $R/tools/tmp_libs
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -DACATTN_CE_STAMPS ${CE6_STAMP_LEFT:+-DCE6_STAMP_LEFT} -DCE6_PIPE=${CE6_PIPE:-0} -c acattn_ce_bf16.hip -o /tmp/ce6_stamps.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v '^acattn_ce_bf16.o$') /tmp/ce6_stamps.o -o $R/tools/tmp_libs/libacattn_ce6stamps.so
  echo built $R/tools/tmp_libs/libacattn_ce6stamps.so
  exit 0
fi
mkdir -p $R/gpurun_out
ACATTN_LIB=$R/tools/tmp_libs/libacattn_ce6stamps.so timeout -k 10 200 python $R/tools/ce6_stamps.py 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/ce6_stamps.txt
