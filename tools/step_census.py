#!/usr/bin/env python3
"""step_census.py <kernel_trace.csv> [--sequence]: per-step kernel census from a rocprofv3 --kernel-trace of bench.py
(--sequence: the launches of one timed step in order, with their durations).
A step is delimited by adam_step_kernel (one launch per step).  bench.py runs the timed (full-schedule, graph) steps and then
the plain trainer's; the census averages the steps that have the most common kernel count, i.e. the timed ones."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want_sequence = "--sequence" in sys.argv[2:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
# one Adam launch per step ([r4]: the cross-entropy forward is no step marker any more -- bench.py's roofline_ce launches it
# 40 times in a row)
marks = [i for i, r in enumerate(rows) if "adam_step_kernel" in r["Kernel_Name"]]
if len(marks) < 3:
    marks = [i for i, r in enumerate(rows) if re.search(r"(^|[ :])ce_fwd_kernel<", r["Kernel_Name"])]
spans = list(zip(marks, marks[1:]))
mode = collections.Counter(b - a for a, b in spans).most_common(1)[0][0]
spans = [(a, b) for a, b in spans if b - a == mode]
steps = len(spans)
seg = [r for a, b in spans for r in rows[a:b]]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:72]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = sum(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) for a, b in spans) / steps / 1e3
print(f"{steps} steps   kernels/step {sum(v[0] for v in agg.values()) / steps:.1f}   sum of durations {sum(v[1] for v in agg.values()) / steps / 1e3:.1f} us"
      f"   wall span {span:.1f} us/step")
for n, v in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n:74s} {v[0] / steps:5.1f} x {v[1] / v[0] / 1e3:8.1f} = {v[1] / steps / 1e3:8.1f} us")
if want_sequence:
    a, b = spans[len(spans) // 2]
    seg = rows[a:b]
    first = [i for i, r in enumerate(seg) if "embed_ln_fwd" in r["Kernel_Name"]]
    if first:
        seg = seg[first[0]:] + seg[:first[0]]  # start at the step's first launch
    print()
    print("one step, launch order (us, kernel, grid x, workgroup x):")
    for r in seg:
        n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{d:7.1f}  {n.split('(')[0][:72]:72s} {r.get('Grid_Size_X', '')} {r.get('Workgroup_Size_X', '')}")
