#!/bin/bash
# rebuild_stream.sh: recompile only the streaming-forward translation units (and relink) -- `make` rebuilds every object
# when an .inc changes (25 translation units, minutes); this is the edit loop of acattn_fwd_stream.inc.
set -e
cd "$(dirname "$0")/../ac_tsr_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-variable -Wno-unused-but-set-variable"
pids=()
for f in acattn_fwd_stream_dh16 acattn_fwd_stream_dh32 acattn_fwd_stream_dh64 acattn_fwd_stream_dh128; do
  /opt/rocm/bin/hipcc $F -fno-honor-nans "$@" -c $f.hip -o $f.o & pids+=($!)
done
/opt/rocm/bin/hipcc $F -c acattn_fwd_stream.hip -o acattn_fwd_stream.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
make -t > /dev/null
rm -f libacattn.so
make libacattn.so > /dev/null
ls -la libacattn.so
