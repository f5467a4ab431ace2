#!/bin/bash
# kernel durations of the fused cross-entropy (rocprofv3 kernel trace of tools/ce_bench.py); optional ACATTN_LIB list
export PYTHONPATH=$PWD; R=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
if [ "$lib" != "default" ]; then export ACATTN_LIB=$R/$lib; else unset ACATTN_LIB; fi
rm -rf $R/gpurun_out/cep
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cep -o cep -- python3 $R/tools/ce_bench.py > $R/gpurun_out/ce_bench.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/cep/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "ce_" in n:
        d[n[27:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("$lib", {k: round(sum(v)/len(v),1) for k,v in d.items()})
PY
done
