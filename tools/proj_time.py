"""Times acattn_projections_fwd / _bwd (HIP events, 50 launches each) against the same products through torch (hipBLASLt).
    python tools/proj_time.py [--hidden 128] [--rows 102400] [--gate 200]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ac_tsr_amd import _lib, linear  # noqa: E402
from ac_tsr_amd.ops import _ptr, _stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--rows", type=int, default=102400)
ap.add_argument("--gate", type=int, default=200)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
H, R, G = a.hidden, a.rows, a.gate
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(R, H, device=dev)
W = {}
for n in ("q", "k", "v", "aq", "ak"):
    W["w" + n], W["b" + n] = 0.1 * torch.randn(H, H, device=dev), 0.1 * torch.randn(H, device=dev)
W["wg"], W["bg"] = 0.1 * torch.randn(G, H, device=dev), 0.1 * torch.randn(G, device=dev)
names = ("wq", "bq", "wk", "bk", "wv", "bv", "waq", "baq", "wak", "bak", "wg", "bg")
p = linear._FusedProjections._problem(x, *(W[n] for n in names))
outs = [torch.empty_like(x) for _ in range(5)] + [torch.empty(R, G, device=dev)]
o = _lib.ProjOut()
o.mq, o.mk, o.mv, o.qa, o.ka, o.gate = (_ptr(t) for t in outs)
lib = _lib.load()


def timed(fn, label, flop):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"{label:28s} {us:9.1f} us   {flop / us / 1e6:7.1f} TFLOP/s")


flop = 2.0 * R * (5 * H * H + G * H)
timed(lambda: _lib.check(lib.acattn_projections_fwd(C.byref(p), C.byref(o), _stream()), "fwd"), "fused forward", flop)


def lib_fwd():
    mq = torch.addmm(W["bq"], x, W["wq"].t())
    mk = torch.addmm(W["bk"], x, W["wk"].t())
    torch.addmm(W["bv"], x, W["wv"].t())
    torch.addmm(W["baq"], mq, W["waq"].t())
    torch.addmm(W["bak"], mk, W["wak"].t())
    torch.addmm(W["bg"], mq, W["wg"].t())


timed(lib_fwd, "library forward (6 GEMMs)", flop)
cot = [torch.randn(R, H, device=dev) for _ in range(5)] + [torch.randn(R, G, device=dev)]
io = _lib.ProjBwdIO()
io.dmq, io.dmk, io.dmv, io.dqa, io.dka, io.dgate = (_ptr(t) for t in cot)
tq, tk, dx = (torch.empty_like(x) for _ in range(3))
io.dmq_total, io.dmk_total, io.dx = _ptr(tq), _ptr(tk), _ptr(dx)
nb = int(lib.acattn_projections_bwd_workspace_bytes(C.byref(p)))
ws = torch.empty(max(nb // 4, 1), device=dev)
io.workspace = _ptr(ws)
timed(lambda: _lib.check(lib.acattn_projections_bwd(C.byref(p), C.byref(io), _stream()), "bwd"), "fused backward (dgrad)", flop)


def lib_bwd():
    t = cot[0].clone().addmm_(cot[3], W["waq"]).addmm_(cot[5], W["wg"])
    u = cot[1].clone().addmm_(cot[4], W["wak"])
    (t @ W["wq"]).addmm_(u, W["wk"]).addmm_(cot[2], W["wv"])


timed(lib_bwd, "library backward (6 GEMMs)", flop)
