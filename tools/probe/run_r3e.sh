#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_e
V="$T/libfwd_r3b_z127.so+pre $T/libfwd_r3b_z255.so+pre $T/libfwd_r3b_z383.so+pre $T/libfwd_r3b_z639.so+pre"
timeout -k 10 200 $P $V -rounds 8 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 8 -full 1 > ${O}_full.txt 2>&1
echo rc=$?
