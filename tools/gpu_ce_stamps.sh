#!/bin/bash
# Cycles per phase of the cross-entropy backward's row-block loop at the benchmark shape.
# Build step (in the container, before gpurun):  tools/gpu_ce_stamps.sh build
# On the GPU box:                                tools/gpu_ce_stamps.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "${1:-}" = "build" ]; then
  set -e
  cd $R/ac_tsr_amd/csrc
  make -j8 > /dev/null
  mkdir -p $R/tools/tmp_libs
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -DACATTN_CE_STAMPS -c acattn_ce.hip -o /tmp/ce_stamps.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v '^acattn_ce.o$') /tmp/ce_stamps.o -o $R/tools/tmp_libs/libacattn_cestamps.so
  echo built $R/tools/tmp_libs/libacattn_cestamps.so
  exit 0
fi
ACATTN_LIB=$R/tools/tmp_libs/libacattn_cestamps.so timeout -k 10 200 python $R/tools/ce_stamps.py 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/ce_stamps.txt
