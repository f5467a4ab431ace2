#!/usr/bin/env python3
"""Run the layer tail (fused launch, and the unfused node) forward + backward a few times, for a kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python3 tools/tail_bench.py ROWS NB [unfused]
NB = rows per wave / 16 of the fused forward (0 = automatic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ac_tsr_amd import tail, _lib
from ac_tsr_amd.state import StepState

NAMES = ("c", "x", "wd", "bd", "g1", "b1", "w1", "bb1", "w2", "bb2", "g2", "b2")


def main():
    rows, nb = int(sys.argv[1]), int(sys.argv[2])
    node = tail._LayerTail if len(sys.argv) > 3 else tail._FusedLayerTail
    H, I, p = 64, 256, 0.5
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    t = dict(c=r(rows, H), x=r(rows, H), wd=0.1 * r(H, H), bd=r(H), g1=r(H), b1=r(H), w1=0.1 * r(I, H), bb1=r(I),
             w2=0.1 * r(H, I), bb2=r(H), g2=r(H), b2=r(H))
    dev = {k: v.requires_grad_(True) for k, v in t.items()}
    cot = r(rows, H)
    _lib.load().acattn_select_layer_tail_blocks(nb)
    st = StepState()
    leaves = [dev[k] for k in NAMES]
    for _ in range(20):
        out = node.apply(*leaves, 1e-12, 1e-12, p, p, None, None, 11, 12, None, st)
        torch.autograd.grad(out, leaves, cot, retain_graph=True)
        with st.attack_pass():
            torch.autograd.grad(out, leaves[:2], cot)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
