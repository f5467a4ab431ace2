#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S listing: per basic block, counts by class.
usage: isa_count.py file.s <kernel-symbol-substring> [--blocks]"""
import re, sys, collections
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_pk_"): return "vpk"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64", "v_mad_i64")): return "vmul32"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")): return op.split()[0]
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"
def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
    blocks, cur, name = [], collections.Counter(), "entry"
    tot = collections.Counter()
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"\.LBB\d+_\d+:", s): continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur)); cur = collections.Counter(); name = m.group(1); continue
        op = s.split()[0]
        c = cls(op); cur[c] += 1; tot[c] += 1
        if c in ("valu", "vpk", "trans"): cur["op:" + op] += 1
    blocks.append((name, cur))
    print("TOTAL", dict(tot))
    if "--blocks" in sys.argv:
        for n, c in blocks:
            k = {a: b for a, b in c.items() if not a.startswith("op:")}
            if sum(k.values()) >= 20: print(n, k)
    if "--ops" in sys.argv:
        want = sys.argv[sys.argv.index("--ops") + 1].split(",")
        for n, c in blocks:
            if n in want:
                print(n, sorted(((b, a[3:]) for a, b in c.items() if a.startswith("op:")), reverse=True))
main()
