"""Per-shape timing of acattn_linear_wgrad (both stages together) against torch's dy.t() @ x + dy.sum(0)."""
import torch
from ac_tsr_amd import ops

def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M, K, N in [(25600, 64, 64), (25600, 64, 256), (25600, 256, 64), (25600, 64, 50)]:
    xs = [torch.randn(M, K, device="cuda") for _ in range(8)]
    gs = [torch.randn(M, N, device="cuda") for _ in range(8)]
    i = [0]
    def ours():
        i[0] = (i[0] + 1) % 8
        return ops.linear_wgrad(xs[i[0]], gs[i[0]], True)
    def lib():
        i[0] = (i[0] + 1) % 8
        return gs[i[0]].t() @ xs[i[0]], gs[i[0]].sum(0)
    print(f"M={M} K={K} N={N}: wgrad {t(ours):.1f} us   torch {t(lib):.1f} us   ideal {(M*(K+N)*4)/8e6:.1f} us @8TB/s", flush=True)
