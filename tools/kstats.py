#!/usr/bin/env python3
"""kstats.py <dir> [substring]: kernel name, calls, average us from the rocprofv3 *kernel_stats.csv under dir."""
import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)):
    print(f.replace(sys.argv[1], "").split("/")[1] if "/" in f.replace(sys.argv[1], "") else f)
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            n = r["Name"].replace("void (anonymous namespace)::", "")[:64]
            print(f"    {n:64s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:9.2f} us  min={float(r['MinNs'])/1e3:8.2f}")
