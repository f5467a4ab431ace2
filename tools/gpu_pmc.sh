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
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$name/$tag -o $tag -- python3 $R/bench.py --kernel-only --kernel-iters 12 > $R/gpurun_out/$name/$tag/run.log 2>&1
  echo "pass $tag rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
pass sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES || exit 1
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA || exit 1
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE GRBM_GUI_ACTIVE || exit 1
python3 - <<PY
import csv, glob, collections
for tag in ("sq1","sq2","sq3","fetch","write"):
    files = glob.glob("$R/gpurun_out/$name/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not files:
        print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "acattn" not in k: continue
        short = "fast_adv" if "fast_kernel" in k and "true" in k else ("fast_spatial" if "fast_kernel" in k else k[:60])
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
