#!/bin/bash
# L = 200: what the K / Ka / V re-reads of every query-block wave cost the streaming forward (ACATTN_ZERO: bit 1 K/Ka, 2 V, 5 M)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_j
V="$T/libfwd_cur.so+pre $T/libfwd_z6.so+pre $T/libfwd_z32.so+pre $T/libfwd_z38.so+pre $T/libfwd_z127.so+pre"
timeout -k 10 200 $P $V -rounds 4 -L 200 > ${O}_L200.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -L 200 -full 1 > ${O}_L200_full.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -L 200 -H 128 -h 4 > ${O}_cfg4.txt 2>&1
echo rc=$?
for f in L200 L200_full cfg4; do echo "== $f"; grep -h "TIME" ${O}_$f.txt | cut -c1-140; done
