#!/bin/bash
# kernel durations of acattn_linear_wgrad's two stages per shape (rocprofv3 kernel trace of tools/wgrad_bench.py)
export PYTHONPATH=$PWD; R=$PWD
cd /tmp && export TMPDIR=/tmp
for wgs in "$@"; do
export ACATTN_WGRAD_WGS=$wgs
rm -rf $R/gpurun_out/wg
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wg -o wg -- python3 $R/tools/wgrad_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/wg/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "wgrad" in n:
        d[(n[27:50],r.get("Grid_Size_X"), r.get("Grid_Size_Y"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("WGS=$wgs", {k: round(sum(v)/len(v),2) for k,v in d.items()})
PY
done
