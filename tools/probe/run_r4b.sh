#!/bin/bash
# Round 4: the round-3 intermediate source (commit 3ab3039: two K / Ka register sets) whose 4-waves-per-SIMD PRE = false
# instantiation spilled 5 registers -- is it still wrong at B >= 256, and where?  old_w3 = the same source at 3 waves per
# SIMD (spill-free), cur = today's kernel (in-kernel affines: no +pre).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r4/probe_b
V="$T/libfwd_cur.so $T/libfwd_old_w3.so $T/libfwd_old_w4.so $T/libfwd_old_w4_pat.so $T/libfwd_old_w4_nonan.so"
timeout -k 10 200 $P $V -rounds 2 -where 1 -B 128 > ${O}_B128.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -where 1 -B 256 > ${O}_B256.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -where 1 -B 512 > ${O}_B512.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -where 1 -B 512 -full 1 > ${O}_B512_full.txt 2>&1
echo rc=$?
for f in B128 B256 B512 B512_full; do echo "== $f"; grep -v "^    M\[\|^    ctx" ${O}_$f.txt | cut -c1-200; done
