import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ac_tsr_amd as A
from ac_tsr_amd import _lib, ops
from tests.test_hip_onehop import _problem
DEV = "cuda"
def run(B, L, H, nh, pin):
    t, kv, lens, g = _problem(B, L, H, nh, seed=202)
    rows = (lens - 1).view(-1, 1)
    names = ["q", "k", "v", "qa", "ka", "gl", "w_order", "b_order", "w_dist", "b_dist", "scalar"]
    dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv.to(DEV), causal=True)
    g2 = torch.Generator().manual_seed(5)
    row_cot = torch.randn(B, 1, H, generator=g2)
    cot = torch.zeros(B, L, H)
    cot.scatter_(1, rows.unsqueeze(-1).expand(-1, -1, H), row_cot)
    lib = _lib.load()
    lib.acattn_select_backward_kernel(pin)
    try:
        ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask, cfg,
                                                    p_drop=0.5, seed=777, read_rows=rows.to(DEV), **{k: dev[k] for k in names[6:]})
        (dgl,) = torch.autograd.grad((ctx_c * cot.to(DEV)).sum(), [dev["gl"]])
    finally:
        lib.acattn_select_backward_kernel(0)
    return dgl.cpu(), rows
for shape in [(5, 50, 256, 2), (512, 50, 64, 2), (3, 200, 256, 2)]:
    a, rows = run(*shape, 0)
    b, _ = run(*shape, 1)   # pinned streaming pair: per-head partials
    d = (a - b).abs()
    print(shape, "max diff", d.max().item(), "scale", b.abs().max().item())
    if d.max() > 1e-4:
        bad = (d > 1e-4).nonzero()
        print("  bad entries", len(bad), "first", bad[:5].tolist(), "read rows", rows.view(-1)[:5].tolist())
        print("  a row", a[bad[0][0], bad[0][1], :8].tolist())
        print("  b row", b[bad[0][0], bad[0][1], :8].tolist())
