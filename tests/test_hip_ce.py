"""GPU suite (-m gpu): fused full-catalogue cross-entropy (acattn_full_sort_ce_*) against torch's
CrossEntropyLoss on materialised logits (fp64 on the CPU as the reference of record, fp32 GPU as a cross-check).
Tolerance: loss 1e-5 relative, gradients 1e-4 of their scale (fp32 accumulation over up to 100k items)."""
import pytest
import torch

from ac_tsr_amd import ce

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,N,H", [(512, 100000, 64), (37, 1000, 64), (16, 257, 64), (130, 5003, 128), (1, 64, 64)])
@pytest.mark.parametrize("scale", [0.02, 1.0])
def test_fused_ce_matches_materialised_logits(B, N, H, scale):
    g = torch.Generator().manual_seed(B + N)
    out = (scale * torch.randn(B, H, generator=g)).requires_grad_(True)
    table = (scale * torch.randn(N, H, generator=g)).requires_grad_(True)
    target = torch.randint(0, N, (B,), generator=g)
    up = torch.randn((), generator=g).item()  # arbitrary upstream factor (the attacked loss uses -1)
    ref = torch.nn.functional.cross_entropy(out.double() @ table.double().t(), target)
    g_out, g_tab = torch.autograd.grad(ref * up, [out, table])
    o = out.detach().to(DEV).requires_grad_(True)
    t = table.detach().to(DEV).requires_grad_(True)
    loss = ce.full_sort_cross_entropy(o, t, target.to(DEV))
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    d_o, d_t = torch.autograd.grad(loss * up, [o, t])
    assert (d_o.cpu() - g_out.float()).abs().max() <= 1e-4 * g_out.abs().max() + 1e-9
    assert (d_t.cpu() - g_tab.float()).abs().max() <= 1e-4 * g_tab.abs().max() + 1e-9


def test_fused_ce_without_table_gradient():
    g = torch.Generator().manual_seed(0)
    out = torch.randn(64, 64, generator=g).to(DEV).requires_grad_(True)
    table = torch.randn(3000, 64, generator=g).to(DEV)  # no grad requested
    target = torch.randint(0, 3000, (64,), generator=g).to(DEV)
    loss = ce.full_sort_cross_entropy(out, table, target)
    (d_o,) = torch.autograd.grad(loss, [out])
    ref_out = out.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_out @ table.t(), target)
    (g_o,) = torch.autograd.grad(ref, [ref_out])
    assert (d_o - g_o).abs().max() <= 1e-5


@pytest.mark.parametrize("B,N,H", [(512, 100000, 64), (37, 1000, 64), (16, 257, 64), (130, 5003, 128), (1, 64, 64), (5, 449, 64)])
@pytest.mark.parametrize("scale", [0.02, 1.0, 4.0])
def test_forward_with_direction_matches_materialised_logits(B, N, H, scale):
    """table_grad=False (acattn_full_sort_ce_fwd_dir): loss and d_out from ONE sweep, for row weights of both signs;
    asking for the table gradient anyway still gives the right answer through the regular backward."""
    g = torch.Generator().manual_seed(B + N + 1)
    out = (scale * torch.randn(B, H, generator=g)).requires_grad_(True)
    table = (scale * torch.randn(N, H, generator=g)).requires_grad_(True)
    target = torch.randint(0, N, (B,), generator=g)
    wrow = torch.randn(B, generator=g)
    ref_rows = torch.nn.functional.cross_entropy(out.double() @ table.double().t(), target, reduction="none")
    g_out, g_tab = torch.autograd.grad((ref_rows * wrow.double()).sum(), [out, table])
    o = out.detach().to(DEV).requires_grad_(True)
    t = table.detach().to(DEV).requires_grad_(True)
    rows = ce.full_sort_cross_entropy_rows(o, t, target.to(DEV), table_grad=False)
    assert (rows.detach().cpu().double() - ref_rows).abs().max() <= 1e-5 * max(1.0, ref_rows.abs().max().item())
    (d_o,) = torch.autograd.grad((rows * wrow.to(DEV)).sum(), [o], retain_graph=True)
    assert (d_o.cpu() - g_out.float()).abs().max() <= 1e-4 * g_out.abs().max() + 1e-9
    d_o2, d_t = torch.autograd.grad((rows * wrow.to(DEV)).sum(), [o, t])
    assert (d_o2.cpu() - g_out.float()).abs().max() <= 1e-4 * g_out.abs().max() + 1e-9
    assert (d_t.cpu() - g_tab.float()).abs().max() <= 1e-4 * g_tab.abs().max() + 1e-9
    mean = ce.full_sort_cross_entropy(o, t, target.to(DEV), table_grad=False)
    assert abs(mean.item() - ref_rows.mean().item()) <= 1e-5 * max(1.0, abs(ref_rows.mean().item()))
