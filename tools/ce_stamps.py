"""Cycles per phase of the cross-entropy backward's row-block loop (B = 512, N = 100000, H = 64), per wave, from a library
whose acattn_ce.hip was compiled with -DACATTN_CE_STAMPS:
    ACATTN_LIB=tools/tmp_libs/libacattn_cestamps.so python tools/ce_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ac_tsr_amd import _lib, ce

lib = _lib.load()
B, N, H = 512, 100000, 64
g = torch.Generator().manual_seed(0)
out = torch.randn(B, H, generator=g).cuda().requires_grad_(True)
table = (0.05 * torch.randn(N, H, generator=g)).cuda().requires_grad_(True)
target = torch.randint(1, N, (B,), generator=g).cuda()
names = ["stage rows (2 barriers)", "logits product (112 MFMAs = 3584)", "soft-max arithmetic", "d_out product (3584)",
         "d_table product (3584)", "park tiles, barrier, fold, slab store"]
fn = lib.acattn_debug_ce_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
for label, table_grad in (("backward with table gradient (ce_bwd_kernel<.., true, false>)", True),):
    for _ in range(3):
        loss = ce.full_sort_cross_entropy(out, table, target, table_grad=True)
        loss.backward()
    torch.cuda.synchronize()
    n_waves = 896
    buf = (C.c_ulonglong * (n_waves * 8))()
    assert fn(buf, n_waves * 8) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 8).astype(np.int64)[:, :6]
    tot = s.sum(axis=1)
    print(label, "- cycles per wave over its 32 row blocks: mean total", int(tot.mean()))
    for k, n in enumerate(names):
        print(f"    {n:38s} mean {s[:, k].mean():9.0f}  ({100 * s[:, k].mean() / tot.mean():4.1f} %)   per row block {s[:, k].mean() / 32:7.0f}")
