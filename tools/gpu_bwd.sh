#!/bin/bash
# average duration of the attention backward kernel over a few eager steps, per library build (ACATTN_LIB list)
export PYTHONPATH=$PWD; R=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
if [ "$lib" != "default" ]; then export ACATTN_LIB=$R/$lib; else unset ACATTN_LIB; fi
rm -rf $R/gpurun_out/bwdp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bwdp -o bwdp -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --kernel-iters 5 > $R/gpurun_out/bwd_bench.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/bwdp/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "acattn_bwd" in n:
        d[n[27:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("$lib", {k: (len(v), round(sum(v)/len(v),1)) for k,v in d.items()})
PY
done
