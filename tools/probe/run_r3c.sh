#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_c
timeout -k 10 120 $P $T/libfwd_stream_r2.so $T/libfwd_r3a.so+pre $T/libfwd_r3b.so+pre $T/libfwd_r3b_up.so+pre > ${O}_ragged.txt 2>&1 && \
timeout -k 10 120 $P $T/libfwd_stream_r2.so $T/libfwd_r3a.so+pre $T/libfwd_r3b.so+pre $T/libfwd_r3b_up.so+pre -full 1 > ${O}_full.txt 2>&1 && \
timeout -k 10 120 $P $T/libfwd_r3b_up.so+pre $T/libfwd_r3b_up_st.so+pre -stamps 8 -rounds 4 > ${O}_upst_ragged.txt 2>&1 && \
timeout -k 10 120 $P $T/libfwd_r3b_up.so+pre $T/libfwd_r3b_up_st.so+pre -stamps 8 -rounds 4 -full 1 > ${O}_upst_full.txt 2>&1
echo rc=$?
