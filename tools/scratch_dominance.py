#!/usr/bin/env python3
"""scratch_dominance.py <file.s> [kernel-substring]: for every kernel of a gfx950 assembly listing (hipcc -S
--cuda-device-only) check that each scratch_load reads only bytes that a scratch_store has written on EVERY path from the
kernel entry (forward must-dataflow over the basic-block graph).  A reload of a never-written spill slot reads what an
earlier wave left in that scratch slot: right when the slot is fresh, wrong once waves share a SIMD.  Measurement helper
(round 4, the spilling streaming-forward builds), not product."""
import re
import sys

SIZE = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "short": 2, "byte": 1, "ubyte": 1, "ushort": 2}


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L") and name is None and m.group(1).startswith("_Z"):
            name, body = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                yield name, body
                name = None
            else:
                body.append(ln)


def analyse(name, body):
    # basic blocks
    blocks, cur, order = {}, "entry", ["entry"]
    blocks[cur] = []
    for ln in body:
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s)
    succ = {b: [] for b in order}
    for k, b in enumerate(order):
        fall = True
        for ins in blocks[b]:
            op = ins.split()[0]
            if op.startswith("s_cbranch"):
                succ[b].append(ins.split()[1].rstrip(","))
            elif op == "s_branch":
                succ[b].append(ins.split()[1])
                fall = False
            elif op == "s_endpgm":
                fall = False
        # a branch in the middle of a block does not occur in compiler output: labels start blocks, branches end them
        if fall and k + 1 < len(order):
            succ[b].append(order[k + 1])
    pred = {b: [] for b in order}
    for b in order:
        for s in succ[b]:
            if s in pred:
                pred[s].append(b)

    def acc(ins):
        m = re.match(r"scratch_(load|store)_(\w+)\s+(.*)", ins)
        if not m:
            return None
        kind, ty, rest = m.groups()
        rest = rest.split(";")[0]
        if not re.search(r"\boff, off\b|\boff,\s*v\[?\d+[:\d\]]*,\s*off\b|\boff\s*$", rest) and "off" not in rest:
            return (kind, None, None)
        if re.search(r"\bs\d+\b|\bv\d+\s*,\s*(s\d+|off)\s*(offset|$)", rest) and kind == "load" and not rest.strip().startswith("v"):
            pass
        mo = re.search(r"offset:(\d+)", rest)
        off = int(mo.group(1)) if mo else 0
        dyn = bool(re.search(r",\s*s\d+", rest)) or (kind == "load" and bool(re.search(r",\s*v\d+\s*,", rest))) or (
            kind == "store" and bool(re.match(r"\s*v\d+\s*,\s*v", rest)))
        return (kind, None if dyn else off, SIZE[ty])

    ALL = None  # top
    out = {b: ALL for b in order}
    out_entry_in = frozenset()
    changed = True
    inn = {}
    while changed:
        changed = False
        for b in order:
            if b == "entry":
                cur = set(out_entry_in)
            else:
                ps = [out[p] for p in pred[b] if out[p] is not ALL]
                if not ps:
                    if not pred[b]:
                        cur = set()
                    else:
                        continue
                else:
                    cur = set(ps[0])
                    for p in ps[1:]:
                        cur &= p
            inn[b] = frozenset(cur)
            for ins in blocks[b]:
                a = acc(ins)
                if a and a[0] == "store" and a[1] is not None:
                    cur.update(range(a[1], a[1] + a[2]))
            cur = frozenset(cur)
            if out[b] is ALL or cur != out[b]:
                out[b] = cur
                changed = True
    bad, n_ld, n_st, n_dyn = [], 0, 0, 0
    for b in order:
        if b not in inn:
            continue
        cur = set(inn[b])
        for ins in blocks[b]:
            a = acc(ins)
            if not a:
                continue
            if a[1] is None:
                n_dyn += 1
                continue
            if a[0] == "store":
                n_st += 1
                cur.update(range(a[1], a[1] + a[2]))
            else:
                n_ld += 1
                miss = [x for x in range(a[1], a[1] + a[2]) if x not in cur]
                if miss:
                    bad.append((b, ins, miss[0], miss[-1]))
    return n_ld, n_st, n_dyn, bad


def main():
    lines = open(sys.argv[1]).read().splitlines()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    rc = 0
    for name, body in kernels(lines):
        if want not in name:
            continue
        n_ld, n_st, n_dyn, bad = analyse(name, body)
        if n_ld + n_st + n_dyn == 0:
            continue
        print(f"{name}: {n_st} spill stores, {n_ld} reloads, {n_dyn} dynamically addressed; {len(bad)} reloads not dominated by a store")
        for b, ins, lo, hi in bad:
            rc = 1
            print(f"    {b}: {ins}   (bytes {lo}..{hi} unwritten on some path)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
