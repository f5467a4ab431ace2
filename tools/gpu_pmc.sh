#!/bin/bash
# PMC passes over the kernel-only bench (few launches). Each pass is its own rocprofv3 run (no tracing domains
# besides --kernel-trace). Usage: gpu_pmc.sh <name> ; results in gpurun_out/<name>/pass*/
set -u
name=${1:-pmc}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  mkdir -p $R/gpurun_out/$name/$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name/$tag -o $tag -- python3 $R/bench.py --kernel-only --kernel-iters 12 --kernel-kinds ${KINDS:-ragged,spatial} > $R/gpurun_out/$name/$tag/run.log 2>&1
  echo "pass $tag rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
pass sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES || exit 1
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA || exit 1
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE GRBM_GUI_ACTIVE || exit 1
python3 $R/tools/pmc_to_json.py $R/gpurun_out/$name $R/gpurun_out/$name/pmc.json
