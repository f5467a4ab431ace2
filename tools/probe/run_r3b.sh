#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r3/probe_b
timeout -k 10 120 $P $T/libfwd_r3a.so+pre $T/libfwd_r3a_st.so+pre -stamps 8 -rounds 4 > ${O}_st_ragged.txt 2>&1 && \
timeout -k 10 120 $P $T/libfwd_r3a.so+pre $T/libfwd_r3a_stw.so+pre -stamps 8 -rounds 4 > ${O}_stw_ragged.txt 2>&1 && \
timeout -k 10 120 $P $T/libfwd_r3a.so+pre $T/libfwd_r3a_st.so+pre -stamps 8 -rounds 4 -full 1 > ${O}_st_full.txt 2>&1 && \
timeout -k 10 120 $P $T/libfwd_r3a.so+pre $T/libfwd_r3a_stw.so+pre -stamps 8 -rounds 4 -full 1 > ${O}_stw_full.txt 2>&1
echo rc=$?
