#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_${1:-f}
shift
V="$T/libfwd_stream_r2.so"
for n in "$@"; do V="$V $T/libfwd_$n"; done
timeout -k 10 200 $P $V -rounds 8 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 8 -full 1 > ${O}_full.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -L 37 > ${O}_L37.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -causal 0 > ${O}_bidir.txt 2>&1 && \
timeout -k 10 200 $P $V -L 200 -rounds 4 -iters 10 -sets 2 > ${O}_L200.txt 2>&1 && \
timeout -k 10 200 $P $V -L 200 -H 128 -h 4 -B 128 -rounds 3 -iters 10 -sets 2 > ${O}_cfg4.txt 2>&1
echo rc=$?
for f in ragged full L37 bidir L200 cfg4; do echo "== $f"; grep -h "TIME\|ctx_cal" ${O}_$f.txt | cut -c1-130; done
