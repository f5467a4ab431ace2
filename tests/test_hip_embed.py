"""GPU suite (-m gpu): the fused embedding front end dropout(LayerNorm(E[idx] + P)) against torch ops in fp64."""
import pytest
import torch

from ac_tsr_amd import fused_embed

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _modules(N, L, H, with_pos, g):
    emb = torch.nn.Embedding(N, H, padding_idx=0)
    pos = torch.nn.Embedding(L + 3, H) if with_pos else None  # more rows than positions: only the first L are touched
    norm = torch.nn.LayerNorm(H, eps=1e-12)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(N, H, generator=g))
        if pos is not None:
            pos.weight.copy_(0.5 * torch.randn(L + 3, H, generator=g))
        norm.weight.copy_(1 + 0.3 * torch.randn(H, generator=g))
        norm.bias.copy_(0.3 * torch.randn(H, generator=g))
    return emb, pos, norm


@pytest.mark.parametrize("B,L,H,N", [(512, 50, 64, 100000), (7, 37, 128, 300), (3, 50, 256, 50), (1, 1, 64, 5)])
@pytest.mark.parametrize("with_pos", [True, False])
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_embed_layernorm_matches_torch(B, L, H, N, with_pos, p):
    g = torch.Generator().manual_seed(B + L + H)
    emb, pos, norm = _modules(N, L, H, with_pos, g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    idx = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    keep = torch.empty(B, L, H).bernoulli_(1 - p, generator=g) if p > 0 else None
    cot = torch.randn(B, L, H, generator=g)
    # fp64 reference
    Ed = emb.weight.detach().double().requires_grad_(True)
    Pd = pos.weight.detach().double().requires_grad_(True) if with_pos else None
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    x = Ed[idx] + (Pd[:L].unsqueeze(0) if with_pos else 0)
    ref = torch.nn.functional.layer_norm(x, (H,), wd, bd, 1e-12)
    if keep is not None:
        ref = ref * keep.double() / (1 - p)
    leaves = [Ed, wd, bd] + ([Pd] if with_pos else [])
    grads = torch.autograd.grad((ref * cot.double()).sum(), leaves)
    grads[0][0] = 0  # nn.Embedding(padding_idx=0): the padding row receives no gradient
    # fused
    emb_c, norm_c = emb.to(DEV), norm.to(DEV)
    pos_c = pos.to(DEV) if with_pos else None
    y, nonzero = fused_embed.embed_layer_norm(idx.to(DEV), emb_c, pos_c, norm_c, p, training=False,
                                              keep=None if keep is None else keep.to(DEV), return_nonzero=True)
    assert (y.detach().cpu() - ref.detach().float()).abs().max() <= 3e-5
    # the same launch writes the structured mask's key-validity bytes (abstract_recommender.py:137)
    assert nonzero.dtype == torch.uint8 and torch.equal(nonzero.cpu(), (idx != 0).to(torch.uint8))
    got = torch.autograd.grad((y * cot.to(DEV)).sum(), [emb_c.weight, norm_c.weight, norm_c.bias] +
                              ([pos_c.weight] if with_pos else []))
    for a, b in zip(got, grads):
        assert (a.cpu() - b.float()).abs().max() <= 1e-4 * b.abs().max() + 1e-6
    if with_pos:
        assert got[3][L:].abs().max() == 0


def test_counter_dropout_statistics_eval_identity_and_bad_ids():
    g = torch.Generator().manual_seed(0)
    B, L, H, N = 256, 50, 64, 1000
    emb, pos, norm = (m.to(DEV) if m is not None else None for m in _modules(N, L, H, True, g))
    idx = torch.randint(1, N, (B, L), generator=g).to(DEV)
    torch.manual_seed(1)
    a = fused_embed.embed_layer_norm(idx, emb, pos, norm, 0.5, training=True)
    b = fused_embed.embed_layer_norm(idx, emb, pos, norm, 0.5, training=True)
    e = fused_embed.embed_layer_norm(idx, emb, pos, norm, 0.5, training=False)
    zero_a, zero_b = (a == 0).float().mean().item(), (b == 0).float().mean().item()
    assert abs(zero_a - 0.5) < 0.01 and abs(zero_b - 0.5) < 0.01 and not torch.equal(a == 0, b == 0)
    kept = a != 0
    assert (a[kept] - 2 * e[kept]).abs().max() <= 1e-5  # survivors are scaled by 1 / (1 - p)
    ref = torch.nn.functional.layer_norm(emb(idx) + pos.weight[:L], (H,), norm.weight, norm.bias, 1e-12)
    assert (e - ref).abs().max() <= 3e-5
    # ids outside the table are clamped, not dereferenced
    bad = idx.clone()
    bad[0, 0], bad[0, 1] = N + 12345, -7
    out = fused_embed.embed_layer_norm(bad, emb, pos, norm, 0.0, training=False)
    assert torch.isfinite(out).all()
    with pytest.raises(IndexError):
        fused_embed.embed_layer_norm(torch.ones(2, L + 4, dtype=torch.long, device=DEV), emb, pos, norm, 0.0, False)


def test_table_gradient_hand_over_equals_autograd_sum():
    """The item table is both the embedding table and the CE classifier (acsasrec.py:87, 117-120).  Inside the
    trainer's calibrated pass (ONE loss walked: recbole/trainer/trainer.py:672-677) the cross-entropy node publishes its
    dense table gradient and the embedding backward scatters into it (StepState.table_grad); anywhere else, and as
    soon as TWO nodes produce a table gradient in one walk, autograd adds the tensors.  Same gradients every way, also
    over two walks; nothing is handed over to a lookup of another table or across walks."""
    import contextlib
    from ac_tsr_amd import ce
    from ac_tsr_amd.state import StepState
    B, L, H, N = 16, 12, 64, 300
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, N, (B, L), generator=g).to(DEV)
    target = torch.randint(1, N, (B,), generator=g).to(DEV)
    target2 = torch.randint(1, N, (B,), generator=g).to(DEV)
    cot = torch.randn(B, L, H, generator=g).to(DEV)

    def grads(state, in_pass=False, two_losses=False):
        torch.manual_seed(0)
        emb = torch.nn.Embedding(N, H, padding_idx=0).to(DEV)
        pos = torch.nn.Embedding(L, H).to(DEV)
        norm = torch.nn.LayerNorm(H, eps=1e-12).to(DEV)
        if state is not None:
            for m in (emb, pos, norm):
                state.attach(m)
        kw = {} if state is None else {"state": state}
        out = []
        for _ in range(2):  # two forward/backward walks: nothing stale survives the first
            emb.weight.grad = None
            y = fused_embed.embed_layer_norm(idx, emb, pos, norm, 0.0, training=True)
            loss = ce.full_sort_cross_entropy(y[:, -1, :], emb.weight, target, **kw)
            if two_losses:  # a second producer of a table gradient in the same walk (sum(losses).backward())
                loss = loss + 0.5 * ce.full_sort_cross_entropy(y[:, 0, :], emb.weight, target2, **kw)
            with (state.calibrated_pass() if in_pass else contextlib.nullcontext()):
                (loss + (y * cot).sum() * 1e-2).backward()
            out.append(emb.weight.grad.clone())
        return out

    close = lambda a, b: (a - b).abs().max() <= 1e-6 * a.abs().max() + 1e-9
    st = StepState()
    ref, got = grads(None), grads(st, in_pass=True)
    assert st.table_grad is None  # published twice, taken twice
    assert all(close(a, b) for a, b in zip(ref, got))
    assert all(close(a, b) for a, b in zip(ref, grads(StepState())))  # no pass: the plain path
    # two producers: the hand-over is withdrawn, inside the pass and outside it
    ref2 = grads(None, two_losses=True)
    for in_pass in (True, False):
        st2 = StepState()
        assert all(close(a, b) for a, b in zip(ref2, grads(st2, in_pass=in_pass, two_losses=True))), in_pass
        assert st2.table_grad is None
    # a gradient published for ANOTHER table is left alone
    other = torch.nn.Embedding(N, H, padding_idx=0).to(DEV)
    with st.calibrated_pass():
        st.publish_table_grad(10 ** 9, other.weight, torch.zeros_like(other.weight))
    assert close(ref[0], grads(st, in_pass=True)[0])


def test_sum_of_both_losses_backward_equals_the_two_pass_gradients():
    """ACSASRec, `(attacked + calibrated).backward()` in ONE walk -- what RecBole's base Trainer does with a tuple of
    losses (recbole/trainer/trainer.py:184-185) -- against the sum of the two separate walks with the hand-over switched
    off: every parameter, the item table's lookup rows in particular."""
    import ac_tsr_amd as A
    from ac_tsr_amd import state as state_mod
    torch.manual_seed(1)
    cfg = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.0, attn_dropout_prob=0.0,
               hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, loss_type="CE", combine_option="gate",
               two_level=True, use_order=True, use_distance=True, rich_calibrated_combine="none", mask_loss_weight=0.03)
    model = A.ACSASRec(A.DictConfig(cfg), A.ItemCount(500)).to(DEV).eval()  # eval: no dropout draws; the noise seed is pinned
    B, L = 32, 50
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, 500, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": torch.randint(1, 500, (B,), generator=g).to(DEV)}

    def run(one_walk):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(7)
        att, cal = model.calculate_loss(batch)
        if one_walk:
            (att + cal).backward()
        else:
            cal.backward(retain_graph=True)
            att.backward()
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    old = state_mod._NO_HANDOVER
    state_mod._NO_HANDOVER = True
    try:
        want = run(False)
    finally:
        state_mod._NO_HANDOVER = old
    got = run(True)
    assert set(got) == set(want)
    for n in want:
        # (float atomics land in a different order from walk to walk: ~2e-4 of a gradient's scale; rows lost to a broken
        # hand-over would be off by the scale itself)
        assert (got[n] - want[n]).abs().max().item() <= 2e-3 * want[n].abs().max().item() + 1e-8, n
