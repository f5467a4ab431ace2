"""Cycles per phase of the one-row attention backward inside a benchmark training step, from a library whose
acattn_bwd_stream.hip was compiled with -DACATTN_ONEROW_STAMPS (waits for all memory traffic at every stamp):
    ACATTN_LIB=tools/tmp_libs/libacattn_onerowstamps.so python tools/onerow_stamps.py"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench  # noqa: F401  (path set-up)
from ac_tsr_amd import _lib

lib = _lib.load()
fn = lib.acattn_debug_onerow_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
# a few eager steps of the benchmark model
import ac_tsr_amd as A
torch.manual_seed(0)
cfg = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.5, attn_dropout_prob=0.5, hidden_act='gelu',
           layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate', two_level=True, use_order=True,
           use_distance=True, mask_loss_weight=0.03, MAX_ITEM_LIST_LENGTH=50, gate_seq_length=50)
model = A.ACSASRec(A.DictConfig(cfg), A.ItemCount(100000)).cuda()
tr = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
model.train()
g = torch.Generator().manual_seed(1)
B, L = 512, 50
lens = torch.randint(1, L + 1, (B,), generator=g)
ids = torch.randint(1, 100000, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
batch = {"item_id_list": ids.cuda(), "item_length": lens.cuda(), "item_id": ids[torch.arange(B), lens - 1].cuda()}
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
n = 1024
buf = (C.c_ulonglong * (n * 8))()
assert fn(buf, n * 8) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
names = ["row vectors -> LDS, barrier", "key rows (K, KA, V), dot products", "gate segment, elementwise, row sums, col[]", "key-side stores (dk, dka, dv)",
         "gate gradient row, barrier", "query side: K / KA by column, sums, dq / dqa rows, partials"]
tot = s[:, :6].sum(axis=1)
print("one-row backward (last launch of the step), cycles per wave, every stamp waits for memory: mean total", int(tot.mean()), "max", int(tot.max()))
for k, nme in enumerate(names):
    print(f"    {nme:60s} mean {s[:, k].mean():8.0f}  ({100 * s[:, k].mean() / tot.mean():4.1f} %)")
