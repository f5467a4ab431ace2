"""Errors of the fused cross-entropy against fp64, exact-fp32 products (mode 0) vs split bf16 products (mode 2)."""
import sys
import torch
from ac_tsr_amd import ce
from ac_tsr_amd._lib import load

lib = load()
DEV = "cuda"


def run(B, N, scale, mode, table_grad=True):
    lib.acattn_full_sort_ce_products(mode)
    g = torch.Generator().manual_seed(B + N)
    out = (scale * torch.randn(B, 64, generator=g)).requires_grad_(True)
    table = (scale * torch.randn(N, 64, generator=g)).requires_grad_(True)
    target = torch.randint(0, N, (B,), generator=g)
    target[: B // 4] = N - 1 - torch.arange(B // 4) % min(N, 1500)
    od, td = out.detach().double().to(DEV).requires_grad_(True), table.detach().double().to(DEV).requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(od @ td.t(), target.to(DEV))
    g_out, g_tab = torch.autograd.grad(ref, [od, td])
    o = out.detach().to(DEV).requires_grad_(True)
    t = table.detach().to(DEV).requires_grad_(True)
    if table_grad:
        loss = ce.full_sort_cross_entropy(o, t, target.to(DEV))
        d_o, d_t = torch.autograd.grad(loss, [o, t])
        e_t = ((d_t.double() - g_tab).abs().max() / g_tab.abs().max()).item()
    else:
        loss = ce.full_sort_cross_entropy(o, t.detach(), target.to(DEV), table_grad=False)
        (d_o,) = torch.autograd.grad(loss, [o])
        e_t = float("nan")
    e_l = abs(loss.item() - ref.item()) / abs(ref.item())
    e_o = ((d_o.double() - g_out).abs().max() / g_out.abs().max()).item()
    return e_l, e_o, e_t


for B, N in [(37, 1000), (512, 100000), (70, 99990), (33, 385), (64, 5000)]:
    for scale in (0.02, 1.0):
        for tg in (True, False):
            a = run(B, N, scale, 0, tg)
            b = run(B, N, scale, 2, tg)
            print(f"B={B} N={N} scale={scale} table_grad={tg}: fp32 loss {a[0]:.2e} d_out {a[1]:.2e} d_table {a[2]:.2e} | split loss {b[0]:.2e} d_out {b[1]:.2e} d_table {b[2]:.2e}", flush=True)
lib.acattn_full_sort_ce_products(1)
