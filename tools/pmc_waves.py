#!/usr/bin/env python3
"""pmc_waves.py <dir>: per kernel and WAVE, from the rocprofv3 --pmc passes of tools/gpu_pmc_all.sh: lifetime, cycles the
matrix pipe worked for the wave, cycles in s_waitcnt, VALU instructions (SQ_WAVE_CYCLES / SQ_WAIT_ANY count in units of 4
cycles; the launch overhead of a profiled kernel is ~28k cycles of GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':46s} {'waves':>6s} {'life/wave':>10s} {'mfma/wave':>10s} {'waitcnt/wave':>12s} {'valu insts/wave':>15s} {'kernel cycles':>13s}")
for n, e in sorted(acc.items()):
    a = {c: sum(v) / len(v) for c, v in e.items()}
    if a.get("SQ_WAVES", 0) < 1 or "SQ_WAVE_CYCLES" not in a:
        continue
    w = a["SQ_WAVES"]
    print(f"{n:46s} {w:6.0f} {4 * a['SQ_WAVE_CYCLES'] / w:10.0f} {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:10.0f} "
          f"{4 * a.get('SQ_WAIT_ANY', 0) / w:12.0f} {a.get('SQ_INSTS_VALU', 0) / w:15.0f} {a.get('GRBM_GUI_ACTIVE', 0) / 8:13.0f}")
