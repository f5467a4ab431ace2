#!/usr/bin/env python3
"""pmc_fold.py <dir> [substring ...]: per-kernel averages of every counter in the rocprofv3 --pmc CSVs under dir."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        n = n.split("(")[0][:48]
        acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
pats = sys.argv[2:]
for k in sorted(acc):
    if pats and not any(p in k for p in pats):
        continue
    e = {c: sum(v) / len(v) for c, v in acc[k].items()}
    print(k, "launches", len(next(iter(acc[k].values()))))
    print("   " + "  ".join(f"{c}={e[c]:.3g}" for c in sorted(e)))
    if "SQ_BUSY_CYCLES" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
        print(f"   mfma busy / busy = {e['SQ_VALU_MFMA_BUSY_CYCLES'] / e['SQ_BUSY_CYCLES']:.3f}"
              f"   wait_inst_any / wave_cycles = {e.get('SQ_WAIT_INST_ANY', 0) / max(e.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
