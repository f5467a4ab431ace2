#!/bin/bash
# Per-phase cycle distribution of the row-resident attention backward at the benchmark shape.
# Build step (in the container, before gpurun):  tools/gpu_bwd_stamps.sh build
# On the GPU box:                                tools/gpu_bwd_stamps.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "${1:-}" = "build" ]; then
  set -e
  cd $R/ac_tsr_amd/csrc
  make -j8 > /dev/null
  mkdir -p $R/tools/tmp_libs
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -DACATTN_BWD_STAMPS -c acattn_bwd_fast.hip -o /tmp/bwd_fast_stamps.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v '^acattn_bwd_fast.o$') /tmp/bwd_fast_stamps.o -o $R/tools/tmp_libs/libacattn_stamps.so
  echo built $R/tools/tmp_libs/libacattn_stamps.so
  exit 0
fi
ACATTN_LIB=$R/tools/tmp_libs/libacattn_stamps.so timeout -k 10 300 python $R/tools/bwd_stamps.py > $R/gpurun_out/bwd_stamps.txt 2>&1
tail -n 50 $R/gpurun_out/bwd_stamps.txt
