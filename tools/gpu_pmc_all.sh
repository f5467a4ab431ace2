#!/bin/bash
# SQ counter passes over a few eager training steps, every kernel (raw CSVs under gpurun_out/<name>/<pass>/; fold with
# tools/pmc_fold.py).  One rocprofv3 run per pass, --kernel-trace only.
set -u
name=${1:-pmc_all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  mkdir -p $R/gpurun_out/$name/$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name/$tag -o $tag -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --kernel-iters 5 > $R/gpurun_out/$name/$tag/run.log 2>&1
  echo "pass $tag rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA || exit 1
pass sq4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_CYCLES || exit 1
find $R/gpurun_out/$name -name "*.csv" ! -name "*counter_collection.csv" -delete
