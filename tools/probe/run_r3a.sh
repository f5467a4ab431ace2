#!/bin/bash
# round 3, first probe of the rewritten streaming forward: correctness vs the round-2 kernel + timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
P=tools/probe/fwd_probe
A=tools/tmp_libs/libfwd_stream_r2.so
N=tools/tmp_libs/libfwd_${1:-r3a}.so
O=gpurun_out/r3/${2:-probe_a}
timeout -k 10 120 $P $A $N $N+pre > ${O}_ragged.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -full 1 > ${O}_full.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -causal 0 -rounds 6 > ${O}_bidir.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -pdrop 0.3 -rounds 4 > ${O}_p03.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -pdrop 0 -rounds 4 > ${O}_p0.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -adv 0 -rounds 4 > ${O}_spatial.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -wscale 0.3 -rounds 2 > ${O}_stress.txt 2>&1 && \
timeout -k 10 120 $P $A $N $N+pre -L 37 -rounds 2 > ${O}_L37.txt 2>&1 && \
timeout -k 10 200 $P $A $N $N+pre -L 200 -rounds 4 -iters 10 -sets 2 > ${O}_L200.txt 2>&1 && \
timeout -k 10 200 $P $A $N $N+pre -L 200 -H 128 -h 4 -B 128 -rounds 3 -iters 10 -sets 2 > ${O}_cfg4.txt 2>&1 && \
timeout -k 10 200 $P $A $N $N+pre -L 200 -H 256 -h 4 -B 64 -causal 0 -rounds 3 -iters 10 -sets 2 > ${O}_cfg5.txt 2>&1
echo rc=$?
grep -h "TIME\|variant" ${O}_*.txt | cut -c1-200
