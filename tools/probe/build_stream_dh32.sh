#!/bin/bash
# build_stream_dh32.sh <name> [extra hipcc flags...]: the streaming forward at head size 32 ONLY (dispatcher + one
# translation unit, the other head sizes stubbed out) into tools/tmp_libs/libfwd_<name>.so for tools/probe/fwd_probe.
# Round 4: variants for the "spilling build computes wrong tiles" question (NANFLAG=-fno-honor-nans for the shipped flags).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/tools/tmp_libs
C=$R/ac_tsr_amd/csrc
S=$R/tools/tmp_libs/stub_$name.hip
cat > $S <<'EOS'
#include "acattn_common.h"
int acattn_launch_fwd_stream_dh16(const acattn_problem&, const acattn_fwd_out&, int, hipStream_t) { return -100; }
int acattn_launch_fwd_stream_dh64(const acattn_problem&, const acattn_fwd_out&, int, hipStream_t) { return -100; }
int acattn_launch_fwd_stream_dh128(const acattn_problem&, const acattn_fwd_out&, int, hipStream_t) { return -100; }
EOS
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I$R/include -I$C -Wno-unused-result -Wno-pass-failed $NANFLAG "$@" \
  $C/acattn_fwd_stream.hip $C/acattn_fwd_stream_dh32.hip $S -o $R/tools/tmp_libs/libfwd_$name.so
rm -f $S
echo built $R/tools/tmp_libs/libfwd_$name.so
