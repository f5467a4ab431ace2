#!/bin/bash
cd $GRAFT_REPO_ROOT
P=tools/probe/fwd_probe
T=tools/tmp_libs
V="$T/libfwd_stream_r2.so $T/libfwd_r3c.so $T/libfwd_r3c.so+pre $T/libfwd_r3a.so"
for args in "-H 256 -h 4 -L 200 -B 64 -causal 0 -sets 2" "-H 256 -h 4 -L 200 -B 64 -causal 0 -sets 1" "-H 256 -h 4 -L 200 -B 32 -causal 0 -sets 1" "-H 256 -h 4 -L 200 -B 64 -causal 1 -sets 1" "-H 256 -h 4 -L 130 -B 64 -causal 0 -sets 1"; do
  echo "== $args"; timeout -k 10 100 $P $V $args -rounds 1 -iters 3 2>&1 | grep "ctx_cal" | cut -c1-120
done
