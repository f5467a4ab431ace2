#!/bin/bash
# build_proj_variant.sh <name> [-D...]: libacattn with acattn_proj.hip rebuilt under probe macros ->
# tools/tmp_libs/libacattn_<name>.so (run with ACATTN_LIB=<that path>; timing only)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/tools/tmp_libs
C=$R/ac_tsr_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C "$@" -c $C/acattn_proj.hip -o $R/tools/tmp_libs/proj_$name.o 2>/dev/null
objs=$(ls $C/*.o | grep -v acattn_proj.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs $R/tools/tmp_libs/proj_$name.o -o $R/tools/tmp_libs/libacattn_$name.so
echo built libacattn_$name.so
