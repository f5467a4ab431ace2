#!/bin/bash
cd $GRAFT_REPO_ROOT
P=tools/probe/fwd_probe
T=tools/tmp_libs
for V in "$T/libfwd_stream_r2.so $T/libfwd_r3d.so" "$T/libfwd_stream_r2.so $T/libfwd_r3d_w2.so $T/libfwd_r3d_w3.so $T/libfwd_r3d_nn.so $T/libfwd_r3d.so" "$T/libfwd_stream_r2.so $T/libfwd_r3c.so"; do
for args in "-B 512" "-B 256" "-B 128"; do
  echo "== $args : $V"; timeout -k 10 100 $P $V $args -rounds 1 -iters 3 2>&1 | grep "ctx_cal" | cut -c1-120
done; done
