"""GPU suite (-m gpu): weight / bias gradients of the tall-skinny linears (acattn_linear_wgrad) against an fp64
product of the same operands, and the layer-level autograd path (linear.skinny_linear) against torch's own."""
import pytest
import torch

from ac_tsr_amd import linear, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("M,K,N", [(25600, 64, 64), (25600, 64, 256), (25600, 256, 64), (25600, 64, 50),
                                   (37, 64, 50), (1000, 128, 128), (3, 64, 64), (4099, 256, 200), (777, 50, 64)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_wgrad_matches_fp64(M, K, N, bias):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    dy = torch.randn(M, N, generator=g)
    dw, db = ops.linear_wgrad(x.to(DEV), dy.to(DEV), bias)
    ref_w = (dy.double().t() @ x.double())
    ref_b = dy.double().sum(0)
    # fp32 accumulation of M products of unit-variance terms: error ~ sqrt(M) * eps * |term|
    tol = 4e-6 * (M ** 0.5) + 1e-5
    assert dw.shape == (N, K)
    assert (dw.cpu().double() - ref_w).abs().max() <= tol * max(1.0, ref_w.abs().max().item() / (M ** 0.5))
    if bias:
        assert (db.cpu().double() - ref_b).abs().max() <= tol * max(1.0, ref_b.abs().max().item() / (M ** 0.5))
    else:
        assert db is None


def test_linear_wgrad_is_deterministic():
    g = torch.Generator().manual_seed(5)
    x, dy = torch.randn(25600, 64, generator=g).to(DEV), torch.randn(25600, 64, generator=g).to(DEV)
    a = ops.linear_wgrad(x, dy, True)
    b = ops.linear_wgrad(x, dy, True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("shape,N", [((512, 50, 64), 64), ((16, 50, 64), 256), ((8, 50, 256), 64), ((4, 50, 64), 50)])
def test_skinny_linear_autograd_matches_torch(shape, N):
    g = torch.Generator().manual_seed(N)
    layer = torch.nn.Linear(shape[-1], N).to(DEV)
    x = torch.randn(*shape, generator=g).to(DEV)
    cot = torch.randn(*shape[:-1], N, generator=g).to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = linear.skinny_linear(xa, layer)
    ga = torch.autograd.grad((ya * cot).sum(), [xa, layer.weight, layer.bias])
    xb = x.clone().requires_grad_(True)
    yb = layer(xb)
    gb = torch.autograd.grad((yb * cot).sum(), [xb, layer.weight, layer.bias])
    assert torch.equal(ya, yb)
    for got, want in zip(ga, gb):
        assert (got - want).abs().max() <= 2e-5 * want.abs().max() + 1e-6


def test_grouped_wgrad_equals_individual_calls():
    """Eight layers' worth of weight/bias gradients in one launch pair: mixed K, mixed N (different numbers of
    64 x 64 blocks per item), mixed bias flags."""
    g = torch.Generator().manual_seed(9)
    M = 25600
    x, y = torch.randn(M, 64, generator=g).to(DEV), torch.randn(M, 64, generator=g).to(DEV)
    items = [(x, torch.randn(M, 64, generator=g).to(DEV), True), (x, torch.randn(M, 64, generator=g).to(DEV), False),
             (y, torch.randn(M, 50, generator=g).to(DEV), True), (x, torch.randn(M, 256, generator=g).to(DEV), True),
             (y, torch.randn(M, 64, generator=g).to(DEV), True), (y, torch.randn(M, 200, generator=g).to(DEV), False),
             (torch.randn(M, 256, generator=g).to(DEV), torch.randn(M, 64, generator=g).to(DEV), True),
             (torch.randn(M, 50, generator=g).to(DEV), torch.randn(M, 64, generator=g).to(DEV), True)]
    got = ops.linear_wgrad_grouped(items)
    assert len(got) == len(items)
    for (xi, gi, wb), (dw, db) in zip(items, got):
        rw, rb = ops.linear_wgrad(xi, gi, wb)
        # (the number of partials per block follows the largest item of a group: same sums, different association)
        assert (dw - rw).abs().max() <= 2e-5 * rw.abs().max()
        assert (db is None) == (not wb) and (db is None or (db - rb).abs().max() <= 2e-5 * rb.abs().max())


def test_deferred_stage_two_equals_the_one_call_form_bit_for_bit():
    """[r4] acattn_linear_wgrad_grouped_partial + ONE acattn_linear_wgrad_reduce_many for items of several groups (different
    M, sizes, with / without bias) and two plain row sums in the same launch, against acattn_linear_wgrad_grouped /
    acattn_sum_rows: the same kernels' sums in the same order, so every element is identical."""
    from types import SimpleNamespace
    from ac_tsr_amd.state import StepState
    g = torch.Generator().manual_seed(11)
    groups = [[(25600, 64, 64, True), (25600, 64, 256, True), (25600, 256, 64, False)], [(512, 64, 64, True)], [(4099, 50, 200, True)]]
    tensors = [[(torch.randn(M, K, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV), wb) for M, K, N, wb in grp] for grp in groups]
    sums_in = [torch.randn(1600, 256, generator=g).to(DEV), torch.randn(1024, 2, 66, generator=g).to(DEV)]
    ref = [ops.linear_wgrad_grouped(grp) for grp in tensors]
    ref_sums = [ops.sum_rows(t, 0) for t in sums_in]
    st = StepState()
    st.defer_reductions = True
    outs, leaves = [], []
    with st.attack_pass([]):  # (opened with leaves = deferral armed; the flush at its end is replaced below)
        assert st.deferring()
        for grp in tensors:
            outs.append(ops.linear_wgrad_grouped(grp, st))
        sums = [ops.sum_rows0(t, st) for t in sums_in]
        jobs = list(st._deferred)
        assert len(jobs) == 5 + 2
        # stand in for autograd: the returned tensors ARE the leaves' gradients
        for grp in outs:
            for gw, gb in grp:
                for t in (gw, gb):
                    if t is not None:
                        leaves.append(SimpleNamespace(grad=t))
        for s in sums:
            st._deferred[[j.get("sum_out") for j in st._deferred].index(s.data_ptr())]["watch"].append(s.data_ptr())
            leaves.append(SimpleNamespace(grad=s))
        st._flush_leaves = leaves
    torch.cuda.synchronize()
    for grp_ref, grp_out in zip(ref, outs):
        for (rw, rb), (gw, gb) in zip(grp_ref, grp_out):
            assert torch.equal(rw, gw)
            assert (rb is None and gb is None) or torch.equal(rb, gb)
    for r, s in zip(ref_sums, sums):
        assert torch.equal(r, s)
    # a destination nobody adopted is an error, not silent garbage
    with pytest.raises(RuntimeError, match="not adopted"):
        with st.attack_pass([]):
            ops.linear_wgrad_grouped(tensors[1], st)
