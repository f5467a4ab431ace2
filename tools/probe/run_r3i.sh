#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_i
V="$T/libfwd_stream_r2.so $T/libfwd_r3f.so+pre $T/libfwd_r3g.so+pre $T/libfwd_r3f_z127.so+pre"
timeout -k 10 200 $P $V -rounds 8 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 8 -full 1 > ${O}_full.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -causal 0 > ${O}_bidir.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -adv 0 > ${O}_spatial.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -pdrop 0.3 > ${O}_p03.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -L 36 > ${O}_L36.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -L 17 -B 64 > ${O}_L17.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -L 51 -wscale 0.3 > ${O}_L51.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -H 128 -h 2 -B 128 > ${O}_dh64.txt 2>&1
echo rc=$?
for f in ragged full bidir spatial p03 L36 L17 L51 dh64; do echo "== $f"; grep -h "TIME\|ctx_cal\|ctx_att" ${O}_$f.txt | grep -v "variant [03]" | cut -c1-125; done
