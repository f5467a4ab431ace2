#!/bin/bash
export PYTHONPATH=$PWD; R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/wgg
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wgg -o wgg -- python3 $R/tools/wgrad_group_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/wgg/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "wgrad" in n:
        d[(n[27:48],r.get("Grid_Size_X"),r.get("Grid_Size_Y"),r.get("Grid_Size_Z"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items()): print(k, len(v), round(sum(v)/len(v),2))
PY
