#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_h
V="$T/libfwd_r3e.so+pre $T/libfwd_r3e_notail.so+pre $T/libfwd_r3e_z127.so+pre $T/libfwd_r3e_notail_z127.so+pre"
timeout -k 10 200 $P $V -rounds 8 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 8 -full 1 > ${O}_full.txt 2>&1
echo rc=$?
for f in ragged full; do echo "== $f"; grep -h "TIME" ${O}_$f.txt | cut -c1-120; done
