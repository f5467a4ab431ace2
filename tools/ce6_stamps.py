"""Cycles per phase of the split-product cross-entropy sweeps (B = 512, N = 100000, H = 64), per wave, from a library whose
acattn_ce_bf16.hip was compiled with -DACATTN_CE_STAMPS:
    ACATTN_LIB=tools/tmp_libs/libacattn_ce6stamps.so python tools/ce6_stamps.py"""
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
names = ["P1 logits (+ operand reads)", "soft-max arithmetic", "split dl, image stores", "P2 d_out + park", "P3 d_table (tr reads)",
         "barrier, stage, fold, barrier", "prologue (table operands)", "leftover units + d_table store"]
fn = lib.acattn_debug_ce6_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int


def dump(label):
    torch.cuda.synchronize()
    n_waves = 1024
    buf = (C.c_ulonglong * (n_waves * 8))()
    assert fn(buf, n_waves * 8) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 8).astype(np.int64)
    tot = s.sum(axis=1)
    print(label, "- cycles per wave: mean total", int(tot.mean()), "max", int(tot.max()))
    for k, n in enumerate(names):
        print(f"    {n:34s} mean {s[:, k].mean():9.0f}  ({100 * s[:, k].mean() / tot.mean():4.1f} %)   per super-block {s[:, k].mean() / 16:7.0f}   max wave {s[:, k].max():8d}")


for _ in range(3):
    loss = ce.full_sort_cross_entropy(out, table, target, table_grad=True)
    loss.backward()
dump("backward with table gradient")
for _ in range(3):
    loss = ce.full_sort_cross_entropy(out, table.detach(), target, table_grad=False)
dump("forward with direction")
names = ["P1 (operand reads + MFMAs)", "soft-max arithmetic + LDS store", "wait + barrier", "DMA issue + fold of previous", "-", "-", "prologue (table operands, first DMA)", "leftover units"]
lse_loss = ce.full_sort_cross_entropy(out.detach(), table.detach(), target)
for _ in range(3):
    lse_loss = ce.full_sort_cross_entropy(out.detach(), table.detach(), target)


def dump_fwd(label):
    torch.cuda.synchronize()
    n_waves = 2048
    buf = (C.c_ulonglong * (n_waves * 8))()
    assert fn(buf, n_waves * 8) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 8).astype(np.int64)
    tot = s.sum(axis=1)
    print(label, "- cycles per wave: mean total", int(tot.mean()), "max", int(tot.max()))
    for k, n in enumerate(names):
        print(f"    {n:38s} mean {s[:, k].mean():9.0f}  ({100 * s[:, k].mean() / tot.mean():4.1f} %)   per super-block {s[:, k].mean() / 16:7.0f}   max wave {s[:, k].max():8d}")


dump_fwd("forward (eight waves of three tiles)")
