#!/bin/bash
# Round 4: do spilling builds of the streaming forward still compute wrong tiles, and what flips it?
# (cur = shipped flags; w5 / w6 = ACATTN_WAVES forced to 5 / 6: 85+ spilled VGPRs; _nan = without -fno-honor-nans;
#  _nopair = without the register-pair asm barriers; _pat = -ftrivial-auto-var-init=pattern; _notail = no tail-block path)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
P=tools/probe/fwd_probe
T=tools/tmp_libs
O=gpurun_out/r4/probe_a
V=""
for n in cur w5 w5_nan w5_nopair w6 cur_pat w5_pat w5_notail; do V="$V $T/libfwd_$n.so+pre"; done
V2=""
for n in cur w5 w5_nan w5_nopair w6 cur_pat w5_pat; do V2="$V2 $T/libfwd_$n.so"; done
timeout -k 10 200 $P $V -rounds 2 -where 1 -B 128 > ${O}_pre_B128.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 3 -where 1 -B 512 > ${O}_pre_B512.txt 2>&1 && \
timeout -k 10 200 $P $V -rounds 2 -where 1 -B 512 -full 1 > ${O}_pre_B512_full.txt 2>&1 && \
timeout -k 10 200 $P $V2 -rounds 2 -where 1 -B 128 > ${O}_nopre_B128.txt 2>&1 && \
timeout -k 10 200 $P $V2 -rounds 2 -where 1 -B 512 > ${O}_nopre_B512.txt 2>&1
echo rc=$?
for f in pre_B128 pre_B512 pre_B512_full nopre_B128 nopre_B512; do echo "== $f"; grep -h "^variant .*ctx_cal\|TIME\|differ" ${O}_$f.txt | cut -c1-170; done
