#!/bin/bash
# build_variant.sh <name> [extra hipcc flags...]: builds ac_tsr_amd/csrc/acattn_fwd_dma.hip (or $SRC) alone into
# tools/tmp_libs/libfwd_<name>.so for tools/probe/fwd_probe.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/tools/tmp_libs
SRC=${SRC:-$R/ac_tsr_amd/csrc/acattn_fwd_dma.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -I$R/include -I$R/ac_tsr_amd/csrc -Wno-unused-result -Wno-inline-asm "$@" $SRC -o $R/tools/tmp_libs/libfwd_$name.so
echo built $R/tools/tmp_libs/libfwd_$name.so
