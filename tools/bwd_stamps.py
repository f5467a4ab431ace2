"""Per-phase cycle distribution of the row-resident attention backward (acattn_bwd_fast.hip) at the benchmark shape.
Needs a library whose acattn_bwd_fast.hip was compiled with -DACATTN_BWD_STAMPS (tools/gpu_bwd_stamps.sh builds it):
    ACATTN_LIB=tools/tmp_libs/libacattn_stamps.so python tools/bwd_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ac_tsr_amd as A
from ac_tsr_amd import _lib

lib = _lib.load()
lib.acattn_select_backward_kernel(2)  # row-resident
B, L, H, nh = 512, 50, 64, 2
g = torch.Generator().manual_seed(1)
mk = lambda *s: torch.randn(*s, generator=g).cuda()
t = {k: mk(B, L, H).requires_grad_(True) for k in ("q", "k", "v", "qa", "ka")}
t["gl"] = mk(B, L, L).requires_grad_(True)
dh = H // nh
w = {k: (0.3 * torch.randn(*s, generator=g)).cuda().requires_grad_(True)
     for k, s in (("w_order", (1, 2 * dh)), ("b_order", (1,)), ("w_dist", (1, 2 * dh)), ("b_dist", (1,)), ("scalar", (1,)))}
lens = torch.randint(1, L + 1, (B,), generator=g)
kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).cuda()
mask = A.StructuredMask(kv, causal=True)
cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
out = A.calibrated_attention(t["q"], t["k"], t["v"], t["qa"], t["ka"], t["gl"], mask, cfg, p_drop=0.5, seed=7, **w)
cot = [mk(B, L, H), mk(B, L, H), mk(B, nh, L, L)]
loss = sum((o * c).sum() for o, c in zip(out[:3], cot))
ins = list(t.values()) + list(w.values())
for _ in range(3):
    torch.autograd.grad(loss, ins, retain_graph=True)
torch.cuda.synchronize()
n_waves = B * nh * 4
buf = (C.c_ulonglong * (n_waves * 16))()
fn = lib.acattn_debug_bwd_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
rc = fn(buf, n_waves * 16)
assert rc == 0, rc
s = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 16).astype(np.int64)
names = ["prologue: requests, row statistics", "stage K/Ka/V, barrier", "phase 0 scores", "phase 1 perturbed (+ key side V)",
         "phase 2 calibrated (+ key side V)", "phase 3 soft-max backward", "phase 4 dq / dqa", "key sides K, Ka",
         "wait at the final barrier", "write-out"]
qb = s[:, 11] & 0xff
t0 = s[s[:, 0] > 0, 0].min()
print(f"launch span {s[:, 10].max() - t0} cycles; wave start spread: p50 {int(np.median(s[:, 0] - t0))}, max {int((s[:, 0] - t0).max())}")
for q in range(4):
    sel = qb == q
    d = np.diff(s[sel, :11], axis=1)
    tot = (s[sel, 10] - s[sel, 0])
    print(f"query block {q}: {sel.sum()} waves, lifetime mean {tot.mean():.0f}")
    for k, n in enumerate(names):
        print(f"    {n:36s} mean {d[:, k].mean():8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
