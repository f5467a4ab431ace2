#!/bin/bash
# first GPU probe of round 2: issue-rate table, then base vs stamped build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 120 tools/probe/valu_rate > gpurun_out/r2/valu_rate.txt 2>&1 && \
timeout -k 10 200 tools/probe/fwd_probe tools/tmp_libs/libfwd_base.so tools/tmp_libs/libfwd_stamps.so -stamps 11 > gpurun_out/r2/probe1.txt 2>&1 && \
timeout -k 10 100 tools/probe/fwd_probe tools/tmp_libs/libfwd_base.so -full 1 > gpurun_out/r2/probe1_full.txt 2>&1
echo rc=$?
