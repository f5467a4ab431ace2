#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/wgt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wgt -o wgt -- python3 $R/tools/wgrad_tail_bench.py > $R/gpurun_out/wgt.log 2>&1 || { tail -n 5 $R/gpurun_out/wgt.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/wgt/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "wgrad_partial" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for i,name in enumerate(["dense+ff1+ff2","dense","ff1","ff2","ff1+ff2"]):
    seg=rows[20*i+5:20*i+20]
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in seg]
    print(name, "grid", seg[0]["Grid_Size_X"], seg[0]["Grid_Size_Y"], seg[0]["Grid_Size_Z"], "avg us", round(sum(d)/len(d),1))
PY
