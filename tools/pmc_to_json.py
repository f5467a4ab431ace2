"""Fold the per-pass rocprofv3 --pmc CSVs written by tools/gpu_pmc.sh into profiles/<out>.json (per kernel, averaged
over launches) and add the corrected HBM byte count bench.py reports as roofline.traffic.
Usage: python tools/pmc_to_json.py gpurun_out/<name> profiles/<out>.json"""
import collections
import csv
import glob
import json
import re
import sys

src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(acattn_\w+_kernel<[^>]*>)", row["Kernel_Name"])
        if m:
            acc[m.group(1).replace(" ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {}
for k, d in acc.items():
    e = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch_corrected"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    kernels[k] = e
json.dump({
    "source": "rocprofv3 --pmc, separate passes (tools/gpu_pmc.sh), bench.py --kernel-only, B=512 L=50 H=64 h=2, "
              "averaged over launches",
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half "
                  "of a wide coalesced read)",
    "kernels": kernels}, open(out, "w"), indent=1)
for k, e in kernels.items():
    print(k, e.get("hbm_bytes_per_launch_corrected"), {c: e[c] for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES") if c in e})
