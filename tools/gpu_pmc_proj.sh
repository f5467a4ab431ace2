#!/bin/bash
# PMC passes over tools/proj_time.py (hidden 128 projections): where do the waves of proj_wide_fwd_kernel wait?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=${1:-pmc_proj}
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  mkdir -p $R/gpurun_out/$name/$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name/$tag -o $tag -- python3 $R/tools/proj_time.py --iters 4 > $R/gpurun_out/$name/$tag/run.log 2>&1
  echo "pass $tag rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
pass sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC || exit 1
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum || echo "tcp pass failed"
python3 - <<PY
import csv, glob, collections
for tag in ("sq1","sq3","sq2","tcp"):
    for f in glob.glob("$R/gpurun_out/$name/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in agg.items():
            if "proj_wide" in k:
                print(tag, k, {a: b for a, b in v.items()})
PY
