#!/bin/bash
# Round 4: M stores issued before passes 2-3 (pass 1.5) against round 3's order (-DACATTN_LATE_M)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r4/probe_${TAG:-d}
V=""
for n in ${VARIANTS:-latem earlym}; do V="$V $T/libfwd_$n.so+pre"; done
timeout -k 10 200 $P $V -rounds 10 -where 1 > ${O}_ragged.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 10 -where 1 -full 1 > ${O}_full.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -where 1 -L 200 > ${O}_L200.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 4 -where 1 -L 200 -H 128 -h 4 > ${O}_cfg4.txt 2>&1
echo rc=$?
for f in ragged full L200 cfg4; do echo "== $f"; grep -h "^variant .*ctx_cal\|TIME" ${O}_$f.txt | cut -c1-170; done
