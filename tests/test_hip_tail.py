"""GPU suite (-m gpu): the fused layer tail (acattn_layer_tail_fwd / _bwd, csrc/acattn_tail.hip)

    a   = LayerNorm(dropout(dense(ctx)) + x)                  recbole/model/layers.py:681-683
    out = LayerNorm(dropout(dense_2(gelu(dense_1(a)))) + a)   recbole/model/layers.py:790-798

against the same chain of torch ops in fp64 on the CPU (output and every gradient), and against the unfused node
(hipBLASLt GEMMs + acattn_dropout_add_layernorm_*) with the in-kernel dropout: same seeds, same decisions.

Tolerances: output 3e-5 absolute (fp32 products, LayerNorm outputs are O(1)); gradients 2e-4 of the tensor's largest
magnitude."""
import pytest
import torch
import torch.nn.functional as F

from ac_tsr_amd import tail
from ac_tsr_amd.state import StepState

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("c", "x", "wd", "bd", "g1", "b1", "w1", "bb1", "w2", "bb2", "g2", "b2")


def _inputs(rows, H, I, seed, scale=0.3):
    scale = scale * (64 / H) ** 0.5  # keeps the products O(1) at every width
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(c=r(rows, H), x=r(rows, H), wd=scale * r(H, H), bd=0.1 * r(H), g1=1 + 0.3 * r(H), b1=0.3 * r(H),
                w1=scale * r(I, H), bb1=0.1 * r(I), w2=scale * r(H, I), bb2=0.1 * r(H), g2=1 + 0.3 * r(H), b2=0.3 * r(H)), g


def _reference(t, keep1, keep2, p, eps):
    d = {k: v.double().requires_grad_(True) for k, v in t.items()}
    drop = lambda z, k: z if k is None else z * (k.double() / (1 - p))
    H = d["c"].shape[-1]
    a = F.layer_norm(drop(F.linear(d["c"], d["wd"], d["bd"]), keep1) + d["x"], (H,), d["g1"], d["b1"], eps)
    h3 = F.linear(F.gelu(F.linear(a, d["w1"], d["bb1"])), d["w2"], d["bb2"])
    out = F.layer_norm(drop(h3, keep2) + a, (H,), d["g2"], d["b2"], eps)
    return d, out


def _apply(node, dev, eps, p, keep1, keep2, seed1, seed2, state):
    return node.apply(*(dev[k] for k in NAMES), eps, eps, p, p, keep1, keep2, seed1, seed2, None, state)


# hidden 128 [round 3]: the streamed-weight kernels (BASELINE configs[3]: H = 128, inner 512); four waves per row block
# up to 8192 rows, one above
@pytest.mark.parametrize("rows,I,H", [(512, 256, 64), (37, 256, 64), (16384 + 21, 256, 64), (100, 128, 64), (32768 + 21, 256, 64), (40000, 128, 64),
                                      (512, 512, 128), (37, 512, 128), (8192 + 21, 512, 128), (100, 256, 128), (20000, 256, 128),
                                      (512, 1024, 256), (37, 1024, 256), (4117, 1024, 256)])
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_fused_tail_matches_fp64_chain(rows, I, p, H):
    eps = 1e-12
    t, g = _inputs(rows, H, I, seed=rows + I)
    keep1 = torch.empty(rows, H).bernoulli_(1 - p, generator=g) if p > 0 else None
    keep2 = torch.empty(rows, H).bernoulli_(1 - p, generator=g) if p > 0 else None
    cot = torch.randn(rows, H, generator=g)
    d, ref = _reference(t, keep1, keep2, p, eps)
    want = torch.autograd.grad((ref * cot.double()).sum(), [d[k] for k in NAMES])
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    to = lambda k: None if k is None else k.to(DEV)
    out = _apply(tail._FusedLayerTail, dev, eps, p, to(keep1), to(keep2), 0, 0, StepState())
    assert (out.detach().cpu() - ref.detach().float()).abs().max() <= 3e-5
    got = torch.autograd.grad((out * cot.to(DEV)).sum(), [dev[k] for k in NAMES], retain_graph=True)
    for k, gv, wv in zip(NAMES, got, want):
        err = (gv.cpu() - wv.float()).abs().max().item()
        assert err <= 2e-4 * wv.abs().max().item() + 1e-6, (k, err)
    # the attack pass skips the parameter gradients (none of these is an attack transform) and returns the same inputs' ones
    st = StepState()
    out2 = _apply(tail._FusedLayerTail, dev, eps, p, to(keep1), to(keep2), 0, 0, st)
    with st.attack_pass():
        gc, gx = torch.autograd.grad((out2 * cot.to(DEV)).sum(), [dev["c"], dev["x"]])
    assert torch.equal(gc, got[0]) and torch.equal(gx, got[1])


@pytest.mark.parametrize("rows,H,I", [(512, 64, 256), (25600, 64, 256), (51200, 64, 256), (512, 128, 512), (25600, 128, 512), (6000, 256, 1024)])
def test_fused_tail_counter_dropout_equals_unfused_node(rows, H, I):
    """In-kernel dropout: the fused launch and the unfused node draw the same keep decisions from the same seeds, so
    outputs and gradients agree to rounding; a different seed changes the output."""
    eps, p = 1e-12, 0.5
    t, g = _inputs(rows, H, I, seed=7)
    cot = torch.randn(rows, H, generator=g).to(DEV)
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    outs, grads = [], []
    for node in (tail._FusedLayerTail, tail._LayerTail):
        out = _apply(node, dev, eps, p, None, None, 1234, 99, StepState())
        outs.append(out)
        grads.append(torch.autograd.grad((out * cot).sum(), [dev[k] for k in NAMES]))
    assert (outs[0] - outs[1]).abs().max() <= 3e-5
    for k, a, b in zip(NAMES, *grads):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 1e-6, k
    other = _apply(tail._FusedLayerTail, dev, eps, p, None, None, 1235, 99, StepState())
    assert (other - outs[0]).abs().max() > 0.1
    again = _apply(tail._FusedLayerTail, dev, eps, p, None, None, 1234, 99, StepState())
    assert torch.equal(again, outs[0])


@pytest.mark.parametrize("H,I", [(64, 256), (128, 512), (256, 1024)])
def test_fused_tail_row_selection_equals_gather_then_tail(H, I):
    """`pick`: the tail on selected positions of [B, L, H] inputs == gather, then the tail, then autograd's scatter
    (the tail is position-wise, abstract_recommender.py:130-134 reads one position per sequence)."""
    B, L, R, eps, p = 37, 50, 3, 1e-12, 0.5
    t, g = _inputs(B * L, H, I, seed=11)
    pick = torch.randint(0, L, (B, R), generator=g)
    cot = torch.randn(B, R, H, generator=g).to(DEV)
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    c3, x3 = dev["c"].view(B, L, H), dev["x"].view(B, L, H)
    idx = pick.to(DEV)
    args = [dev[k] for k in NAMES[2:]]
    out_a = tail._FusedLayerTail.apply(c3, x3, *args, eps, eps, p, p, None, None, 5, 6, None, StepState(), idx)
    index = idx.unsqueeze(-1).expand(-1, -1, H)
    out_b = tail._FusedLayerTail.apply(c3.gather(1, index), x3.gather(1, index), *args, eps, eps, p, p, None, None, 5, 6,
                                       None, StepState())
    assert out_a.shape == (B, R, H) and torch.equal(out_a, out_b)
    leaves = [dev[k] for k in NAMES]
    ga = torch.autograd.grad((out_a * cot).sum(), leaves)
    gb = torch.autograd.grad((out_b * cot).sum(), leaves)
    for k, a, b in zip(NAMES, ga, gb):
        assert (a - b).abs().max() <= 1e-6 * b.abs().max() + 1e-9, k
