"""Times the fused layer tail (acattn_layer_tail_fwd / _bwd) against the unfused node (hipBLASLt GEMMs + fused LayerNorm
launches + ATen GELU): forward, and forward + backward, HIP events.  python tools/tail_time.py [--hidden 128 --inner 512 --rows 102400]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ac_tsr_amd import tail  # noqa: E402
from ac_tsr_amd.state import StepState  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--inner", type=int, default=512)
ap.add_argument("--rows", type=int, default=102400)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
H, I, R = a.hidden, a.inner, a.rows
dev = "cuda"
torch.manual_seed(0)
NAMES = ("c", "x", "wd", "bd", "g1", "b1", "w1", "bb1", "w2", "bb2", "g2", "b2")
shapes = dict(c=(R, H), x=(R, H), wd=(H, H), bd=(H,), g1=(H,), b1=(H,), w1=(I, H), bb1=(I,), w2=(H, I), bb2=(H,), g2=(H,), b2=(H,))
t = {k: (0.1 * torch.randn(*s, device=dev)).requires_grad_(True) for k, s in shapes.items()}
cot = torch.randn(R, H, device=dev)
flop_f = 2.0 * R * (H * H + 2 * H * I)


def timed(fn, label, flop):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"{label:44s} {us:9.1f} us   {flop / us / 1e6:7.1f} TFLOP/s")
    return us


for name, node in (("fused", tail._FusedLayerTail), ("unfused", tail._LayerTail)):
    def fwd():
        with torch.no_grad():
            return node.apply(*(t[k] for k in NAMES), 1e-12, 1e-12, 0.5, 0.5, None, None, 11, 12, None, StepState())

    def fwd_bwd(inputs_only=False):
        out = node.apply(*(t[k] for k in NAMES), 1e-12, 1e-12, 0.5, 0.5, None, None, 11, 12, None, StepState())
        leaves = [t["c"], t["x"]] if inputs_only else [t[k] for k in NAMES]
        torch.autograd.grad((out * cot).sum(), leaves)

    f = timed(fwd, f"{name} forward", flop_f)
    fb = timed(fwd_bwd, f"{name} forward + backward (all gradients)", 3 * flop_f)
    print(f"{'':44s} backward alone ~ {fb - f:9.1f} us")
