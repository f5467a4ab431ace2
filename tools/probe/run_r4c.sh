#!/bin/bash
# Round 4: does the dead-row quantisation (wave-uniform blocks in passes 1-3) cost the hot path anything?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r4/probe_c
V="$T/libfwd_cur.so+pre $T/libfwd_deadfix.so+pre $T/libfwd_cur.so $T/libfwd_deadfix.so"
timeout -k 10 200 $P $V -rounds 8 -where 1 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 8 -where 1 -full 1 > ${O}_full.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -where 1 -L 200 > ${O}_L200.txt 2>&1
echo rc=$?
for f in ragged full L200; do echo "== $f"; grep -h "^variant .*ctx_cal\|TIME" ${O}_$f.txt | cut -c1-170; done
