#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r4/probe_e
timeout -k 10 200 $P $T/libfwd_now.so+pre $T/libfwd_st8.so+pre -rounds 4 -stamps 8 > ${O}_stamps_ragged.txt 2>&1 && \
timeout -k 10 200 $P $T/libfwd_now.so+pre $T/libfwd_st8w.so+pre -rounds 4 -stamps 8 > ${O}_stampsw_ragged.txt 2>&1 && \
timeout -k 10 200 $P $T/libfwd_now.so+pre $T/libfwd_st8.so+pre -rounds 4 -stamps 8 -full 1 > ${O}_stamps_full.txt 2>&1
echo rc=$?
for f in stamps_ragged stampsw_ragged stamps_full; do echo "== $f"; grep -h "TIME\|stamp\|rank\|wave" ${O}_$f.txt | cut -c1-170; done
