#!/bin/bash
# PMC pass over a few L = 200, d = 64 training steps: instruction counts and matrix-pipe time of the streaming backward pair
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_l200
for tag in sq1 sq2; do
  if [ $tag = sq1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_INST_ANY"; else C="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; fi
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l200/$tag -o $tag -- python3 $R/bench.py --seq-len 200 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-other-configs --kernel-kinds ragged --kernel-iters 3 > $R/gpurun_out/pmc_l200/$tag.log 2>&1
  echo "pass $tag rc=$?"
done
python3 - <<PY
import csv, glob, collections
for tag in ("sq1","sq2"):
    for f in glob.glob("$R/gpurun_out/pmc_l200/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add((k, r["Dispatch_Id"]))
        for k,_ in seen: n[k]+=1
        for k, v in agg.items():
            if "bwd_row" in k or "bwd_key" in k or "fwd_stream" in k:
                print(tag, k, "launches", n[k], {a: round(b / n[k]) for a, b in v.items()})
PY
