#!/bin/bash
# kernel durations of the layer tail at several sizes / rows-per-wave settings (rocprofv3 kernel trace of tools/tail_bench.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2/tailp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "25600 1" "25600 2" "512 1" "512 0" "4096 0" "25600 0 unfused" "512 0 unfused"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o t -- python3 $R/tools/tail_bench.py $cfg > $O/$tag.log 2>&1 || exit 1
  echo "== $cfg"
  python3 - $O/$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n, c, a = r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3
    tot += float(r["TotalDurationNs"]) / 1e3
    if "tail_" in n or c >= 20:
        print(f"  {n[:70]:70s} calls={c:4d} avg={a:8.2f} us")
print(f"  total kernel time per iteration: {tot / 20:8.1f} us")
PY
done
