#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_d
V="$T/libfwd_r3b.so+pre $T/libfwd_r3b_z127.so+pre $T/libfwd_r3b_z96.so+pre $T/libfwd_r3b_z31.so+pre $T/libfwd_r3b_z1.so+pre $T/libfwd_r3b_z2.so+pre $T/libfwd_r3b_z4.so+pre $T/libfwd_r3b_z24.so+pre $T/libfwd_r3b_z32.so+pre"
timeout -k 10 200 $P $V -rounds 8 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 8 -full 1 > ${O}_full.txt 2>&1
echo rc=$?
