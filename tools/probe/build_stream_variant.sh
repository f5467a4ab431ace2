#!/bin/bash
# build_stream_variant.sh <name> [extra hipcc flags...]: builds the streaming forward kernel (dispatcher + the three
# head-size translation units) into tools/tmp_libs/libfwd_<name>.so for tools/probe/fwd_probe.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/tools/tmp_libs
C=$R/ac_tsr_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I$R/include -I$C -Wno-unused-result $NANFLAG "$@" \
  $C/acattn_fwd_stream.hip $C/acattn_fwd_stream_dh16.hip $C/acattn_fwd_stream_dh32.hip $C/acattn_fwd_stream_dh64.hip \
  -o $R/tools/tmp_libs/libfwd_$name.so
echo built $R/tools/tmp_libs/libfwd_$name.so
