"""GPU suite (-m gpu): fused LayerNorm(dropout(z) + residual) against torch ops (fp64 reference on the CPU)."""
import pytest
import torch

from ac_tsr_amd import fused_ln

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("rows,H", [(25600, 64), (37, 64), (1000, 128), (513, 256)])
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_fused_ln_matches_torch(rows, H, p):
    g = torch.Generator().manual_seed(rows + H)
    z = torch.randn(rows, H, generator=g)
    res = torch.randn(rows, H, generator=g)
    norm = torch.nn.LayerNorm(H, eps=1e-12)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(H, generator=g))
        norm.bias.copy_(0.3 * torch.randn(H, generator=g))
    keep = torch.empty(rows, H).bernoulli_(1 - p, generator=g) if p > 0 else None
    cot = torch.randn(rows, H, generator=g)
    # fp64 reference
    zd, rd = z.double().requires_grad_(True), res.double().requires_grad_(True)
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    s = (zd * (keep.double() / (1 - p)) if keep is not None else zd) + rd
    ref = torch.nn.functional.layer_norm(s, (H,), wd, bd, 1e-12)
    gz, gr, gw, gb = torch.autograd.grad((ref * cot.double()).sum(), [zd, rd, wd, bd])
    # fused
    nd = torch.nn.LayerNorm(H, eps=1e-12).to(DEV)
    nd.load_state_dict(norm.state_dict())
    zc, rc = z.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    y = fused_ln.dropout_add_layer_norm(zc, rc, nd, p, training=False, keep=None if keep is None else keep.to(DEV))
    assert (y.detach().cpu() - ref.detach().float()).abs().max() <= 2e-5
    dz, dr, dw, db = torch.autograd.grad((y * cot.to(DEV)).sum(), [zc, rc, nd.weight, nd.bias])
    for got, want in ((dz, gz), (dr, gr), (dw, gw), (db, gb)):
        assert (got.cpu() - want.float()).abs().max() <= 1e-4 * want.abs().max() + 1e-6


def test_fused_ln_counter_dropout_statistics_and_replay():
    rows, H = 4096, 64
    norm = torch.nn.LayerNorm(H, eps=1e-12).to(DEV)
    z = torch.ones(rows, H, device=DEV, requires_grad=True)
    res = torch.zeros(rows, H, device=DEV)
    torch.manual_seed(5)
    y1 = fused_ln.dropout_add_layer_norm(z, res, norm, 0.5, training=True)
    torch.manual_seed(5)
    y2 = fused_ln.dropout_add_layer_norm(z, res, norm, 0.5, training=True)
    y3 = fused_ln.dropout_add_layer_norm(z, res, norm, 0.5, training=True)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    # z = 1, residual = 0: after LayerNorm the kept entries are positive, the dropped ones negative
    kept = (y1 > 0).float().mean().item()
    assert abs(kept - 0.5) < 0.01
    cot = torch.randn(rows, H, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    (dz,) = torch.autograd.grad((y1 * cot).sum(), [z])
    assert torch.isfinite(dz).all()
    assert (dz[y1 <= 0] == 0).all()  # nothing flows back through dropped entries ...
    assert (dz[y1 > 0] != 0).float().mean() > 0.999  # ... and (generically) something through every kept one


@pytest.mark.parametrize("shape,dim", [((25600, 64), 0), ((128, 64, 64), 0), ((512, 2, 2500), 1), ((1024, 4), 0),
                                       ((7, 3, 5), 1), ((512, 128), 0), ((33, 50), 0)])
def test_sum_rows_kernel(shape, dim):
    from ac_tsr_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(DEV)
    got = ops.sum_rows(x, dim)
    want = x.double().sum(dim)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max() <= 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("shape", [(512, 2, 50, 50), (3, 4, 37, 37), (1, 1, 1, 3), (2, 2, 50, 51)])
def test_mask_penalty_matches_torch_norm(shape):
    """ops.mask_penalty == torch.norm(1 - M, p=2) (acsasrec.py:135) and so does its gradient."""
    from ac_tsr_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    m = torch.rand(*shape, generator=g)
    md = m.double().requires_grad_(True)
    ref = torch.norm(1 - md, p=2)
    (gref,) = torch.autograd.grad(ref * 0.37, [md])
    mc = m.to(DEV).requires_grad_(True)
    out = ops.mask_penalty(mc)
    assert abs(out.item() - ref.item()) <= 2e-6 * ref.item() + 1e-6
    (gm,) = torch.autograd.grad(out * 0.37, [mc])
    assert (gm.cpu().double() - gref).abs().max() <= 1e-6 * gref.abs().max() + 1e-9
    ones = torch.ones(*shape, device=DEV, requires_grad=True)  # zero norm: zero gradient, like torch
    o = ops.mask_penalty(ones)
    (g1,) = torch.autograd.grad(o, [ones])
    assert o.item() == 0 and g1.abs().max() == 0
