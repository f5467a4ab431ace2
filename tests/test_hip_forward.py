"""GPU suite (-m gpu): the HIP forward, called through the C ABI, against the oracle and the goldens.

Tolerances (fp32, stated per BASELINE.json north_star "logits within 1e-4"):
  probabilities  |diff| <= 5e-6 at init-scale weights; 1e-4 on the stress cases whose spatial calibrator
                 runs into the ill-conditioned log(1 - sigmoid(o)) regime (SURVEY.md section 7 "numerics");
  contexts / layer outputs |diff| <= 1e-4; model logits |diff| <= 1e-4.
"""
import pytest
import torch

import ac_tsr_amd as A
from ac_tsr_amd import _lib
from oracle import ac_tsr_ref as O
from tests._golden import ENCODER_CASES, Case

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _project(c: Case, layer=0, x=None):
    cfg = c.encoder_cfg()
    P = O.layer_params(c.params(), f"layer.{layer}.")
    x = c.t("in.x") if x is None else x
    with torch.no_grad():
        mq = O._lin(x, P, "attack_attention.query")
        mk = O._lin(x, P, "attack_attention.key")
        mv = O._lin(x, P, "attack_attention.value")
        qa = O._lin(mq, P, "attack_attention.attack_query_transform")
        ka = O._lin(mk, P, "attack_attention.attack_key_transform")
        gl = O._lin(mq, P, "gate") if cfg.combine_option == "gate" else None
    return cfg, P, mq, mk, mv, qa, ka, gl


def _core_kwargs(P, cfg):
    kw = {}
    if cfg.use_order:
        kw.update(w_order=P["attack_attention.order_affine.weight"].to(DEV), b_order=P["attack_attention.order_affine.bias"].to(DEV))
    if cfg.use_distance:
        kw.update(w_dist=P["attack_attention.distance_affine.weight"].to(DEV), b_dist=P["attack_attention.distance_affine.bias"].to(DEV),
                  scalar=P["attack_attention.scalar"].to(DEV))
    if "rich_calibrated_combine_ratio" in P:  # layers.py:874-875
        kw.update(rich_ratio=P["rich_calibrated_combine_ratio"].to(DEV))
    return kw


def _tols(c: Case):
    stress = float(c.raw["meta.sigma"]) > 0.1
    return (1e-4 if stress else 5e-6), 1e-4


def _oracle_core(c, cfg, P, mq, mk, mv, qa, ka, gl, mask, noise, **kw):
    zero = torch.zeros(1, 2 * cfg.hidden_size // cfg.n_heads)
    with torch.no_grad():
        return O.core_from_projected(
            mq, mk, mv, qa, ka, gl, mask,
            P.get("attack_attention.order_affine.weight", zero), P.get("attack_attention.order_affine.bias"),
            P.get("attack_attention.distance_affine.weight", zero), P.get("attack_attention.distance_affine.bias"),
            P.get("attack_attention.scalar"), cfg, noise, rich_ratio=P.get("rich_calibrated_combine_ratio"), **kw)


@pytest.mark.parametrize("name", ENCODER_CASES)
@pytest.mark.parametrize("mask_kind", ["dense", "structured"])
def test_core_matches_oracle_and_golden(name, mask_kind):
    c = Case(name)
    cfg, P, mq, mk, mv, qa, ka, gl = _project(c)
    bidir = bool(int(c.raw["meta.bidirectional"]))
    mask = c.t("in.mask")
    noise = c.t("in.noise.0")
    ref = _oracle_core(c, cfg, P, mq, mk, mv, qa, ka, gl, mask, noise)
    if mask_kind == "dense":
        dmask = mask.to(DEV).contiguous()
    else:
        dmask = A.StructuredMask((c.t("in.item_seq") != 0).to(torch.uint8).to(DEV), causal=not bidir)
    acfg = A.AttentionConfig(n_heads=cfg.n_heads, combine_option=cfg.combine_option, two_level=cfg.two_level,
                             rich_calibrated_combine=cfg.rich_calibrated_combine)
    d = lambda t: None if t is None else t.to(DEV).contiguous()
    ctx_a, ctx_c, M, probs = A.calibrated_attention(
        d(mq), d(mk), d(mv), d(qa), d(ka), d(gl), dmask, acfg, rnd=A.ExplicitRandomness(noise=d(noise)),
        want_probs=True, **_core_kwargs(P, cfg))
    tp, to = _tols(c)
    assert (M.cpu() - ref["M"]).abs().max() <= tp
    assert (M.cpu() - c.t("out.0.M")).abs().max() <= tp  # the reference's own tensor
    assert (probs["after_spatial"].cpu() - ref["after"]).abs().max() <= tp
    assert (probs["before_spatial"].cpu() - ref["before"]).abs().max() <= tp
    assert (probs["perturbed_attention"].cpu() - ref["attacked"]).abs().max() <= tp
    assert (probs["calibrated_attention"].cpu() - ref["combined"]).abs().max() <= tp
    assert (probs["calibrated_attention"].cpu() - c.t("out.0.calibrated_attention")).abs().max() <= tp
    assert (ctx_a.cpu() - c.t("out.0.ctx_attacked")).abs().max() <= to
    assert (ctx_c.cpu() - c.t("out.0.ctx_calibrated")).abs().max() <= to
    # fast path (no probability dumps) must give the same contexts and M
    ctx_a2, ctx_c2, M2, _ = A.calibrated_attention(
        d(mq), d(mk), d(mv), d(qa), d(ka), d(gl), dmask, acfg, rnd=A.ExplicitRandomness(noise=d(noise)),
        **_core_kwargs(P, cfg))
    assert (M2 - M).abs().max() <= 1e-6
    assert (ctx_a2.cpu() - c.t("out.0.ctx_attacked")).abs().max() <= to
    assert (ctx_c2.cpu() - c.t("out.0.ctx_calibrated")).abs().max() <= to


def _build_encoder(c: Case):
    cfg = c.encoder_cfg()
    enc = A.AttackRTransformerEncoder(cfg.n_layers, cfg.n_heads, cfg.hidden_size, cfg.inner_size, 0.5, 0.5, "gelu", 1e-12,
                                      cfg.combine_option, cfg.use_order, cfg.use_distance, cfg.two_level,
                                      cfg.rich_calibrated_combine, cfg.seq_length)
    enc.load_state_dict(c.params())
    return cfg, enc.to(DEV).eval()


@pytest.mark.parametrize("name", ENCODER_CASES)
def test_encoder_module_matches_reference_outputs(name):
    """Drop-in check: load the reference's state dict, call forward like the reference does."""
    c = Case(name)
    cfg, enc = _build_encoder(c)
    rnds = [A.ExplicitRandomness(noise=c.t(f"in.noise.{i}").to(DEV)) for i in range(cfg.n_layers)]
    if cfg.combine_option == "annealing":
        for l in enc.layer:
            l.anneal_step = 0
    with torch.no_grad():
        outs, masks, probs = enc(c.t("in.x").to(DEV), c.t("in.mask").to(DEV), output_all_encoded_layers=True,
                                 return_all_attention_prob=True, _rnds=rnds)
    tp, to = _tols(c)
    for i in range(cfg.n_layers):
        assert (outs[i][0].cpu() - c.t(f"out.{i}.attacked")).abs().max() <= 5 * to
        assert (outs[i][1].cpu() - c.t(f"out.{i}.calibrated")).abs().max() <= 5 * to
        assert (masks[i].cpu() - c.t(f"out.{i}.M")).abs().max() <= 5 * tp
        assert (probs[i]["calibrated_attention"].cpu() - c.t(f"out.{i}.calibrated_attention")).abs().max() <= 5 * tp
        for k in ("before_spatial", "after_spatial", "perturbed_attention"):
            if c.has(f"out.{i}.{k}"):
                assert (probs[i][k].cpu() - c.t(f"out.{i}.{k}")).abs().max() <= 5 * tp
    # return arities of layers.py:1127-1131
    with torch.no_grad():
        r2 = enc(c.t("in.x").to(DEV), c.t("in.mask").to(DEV), _rnds=rnds)
        r3 = enc(c.t("in.x").to(DEV), c.t("in.mask").to(DEV), return_attention_prob=True, _rnds=rnds)
    assert len(r2) == 2 and len(r3) == 3 and len(r3[2]) == cfg.n_layers


def _build_model(c: Case):
    cfg = c.model_cfg()
    m = A.ACSASRec(A.DictConfig(
        n_layers=cfg.enc.n_layers, n_heads=cfg.enc.n_heads, hidden_size=cfg.enc.hidden_size, inner_size=cfg.enc.inner_size,
        hidden_dropout_prob=0.5, attn_dropout_prob=0.5, hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02,
        loss_type='CE', combine_option='gate', two_level=True, use_order=True, use_distance=True,
        rich_calibrated_combine='none', mask_loss_weight=cfg.mask_loss_weight, MAX_ITEM_LIST_LENGTH=cfg.max_seq_length),
        A.ItemCount(cfg.n_items))
    m.load_state_dict(c.params())
    return cfg, m.to(DEV)


@pytest.mark.parametrize("name", ["model_eval", "model_eval_stress", "model_beauty"])
def test_model_logits_within_1e4(name):
    c = Case(name)
    cfg, m = _build_model(c)
    m.eval()
    batch = {k: v.to(DEV) for k, v in c.batch().items()}
    with torch.no_grad():
        none, scores = m.full_sort_predict(batch)
    assert none is None
    assert (scores.cpu() - c.t("out.logits")).abs().max() <= 1e-4


def test_spatial_only_operator():
    """BASELINE config 2: spatial calibrator only -> ctx = after_spatial . V."""
    c = Case("enc_gate_stress")
    cfg, P, mq, mk, mv, qa, ka, gl = _project(c)
    ref = _oracle_core(c, cfg, P, mq, mk, mv, qa, ka, gl, c.t("in.mask"), c.t("in.noise.0"))
    v = O._heads(mv, cfg.n_heads).permute(0, 2, 1, 3)
    expect = O.context_only(ref["after"], v)
    acfg = A.AttentionConfig(n_heads=cfg.n_heads, adversarial=False)
    smask = A.StructuredMask((c.t("in.item_seq") != 0).to(torch.uint8).to(DEV), causal=True)
    _, ctx, M, _ = A.calibrated_attention(mq.to(DEV), mk.to(DEV), mv.to(DEV), None, None, None, smask, acfg,
                                          **_core_kwargs(P, cfg))
    assert M is None
    assert (ctx.cpu() - expect).abs().max() <= 1e-4


def test_explicit_dropout_masks_match_oracle():
    """Training-mode semantics with the dropout draws handed over as tensors."""
    c = Case("enc_gate_init")
    cfg, P, mq, mk, mv, qa, ka, gl = _project(c)
    g = torch.Generator().manual_seed(5)
    shape = c.t("in.noise.0").shape
    ka_, km_ = (torch.empty(shape).bernoulli_(0.5, generator=g) for _ in range(2))
    ref = _oracle_core(c, cfg, P, mq, mk, mv, qa, ka, gl, c.t("in.mask"), c.t("in.noise.0"), keep_after=ka_, keep_mask=km_)
    acfg = A.AttentionConfig(n_heads=cfg.n_heads, combine_option="gate")
    rnd = A.ExplicitRandomness(noise=c.t("in.noise.0").to(DEV), keep_after=ka_.to(torch.uint8).to(DEV),
                               keep_mask=km_.to(torch.uint8).to(DEV))
    ctx_a, ctx_c, M, _ = A.calibrated_attention(mq.to(DEV), mk.to(DEV), mv.to(DEV), qa.to(DEV), ka.to(DEV), gl.to(DEV),
                                                c.t("in.mask").to(DEV), acfg, p_drop=0.5, rnd=rnd, **_core_kwargs(P, cfg))
    assert (M.cpu() - ref["M"]).abs().max() <= 5e-6
    assert (ctx_a.cpu() - ref["ctx_attacked"]).abs().max() <= 1e-4
    assert (ctx_c.cpu() - ref["ctx_calibrated"]).abs().max() <= 1e-4


def test_counter_rng_replays_exactly_and_has_the_right_statistics():
    B, L, H, nh = 64, 50, 64, 2
    g = torch.Generator().manual_seed(3)
    q, k, v, qa, ka = (torch.randn(B, L, H, generator=g).to(DEV) for _ in range(5))
    gl = torch.randn(B, L, L, generator=g).to(DEV)
    kv = torch.ones(B, L, dtype=torch.uint8, device=DEV)
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv, causal=True)
    seed = 0x1234_5678_9ABC
    out1 = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=0.5, seed=seed)
    out1b = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=0.5, seed=seed)
    rnd = A.materialize_randomness(B, nh, L, seed, 0.5, DEV)
    out2 = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=0.5, rnd=rnd)
    for a, b_, c_ in zip(out1[:3], out1b[:3], out2[:3]):
        assert torch.equal(a, b_)
        assert torch.equal(a, c_)
    out3 = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=0.5, seed=seed + 1)
    assert not torch.equal(out1[2], out3[2])
    n = rnd.noise.flatten().double()
    assert abs(n.mean().item()) < 5e-3 and abs(n.var().item() - 1) < 1e-2
    assert abs((n ** 4).mean().item() - 3) < 0.1  # kurtosis of a normal
    for keep in (rnd.keep_after, rnd.keep_mask, rnd.keep_before):
        assert abs(keep.float().mean().item() - 0.5) < 5e-3
    assert abs((rnd.keep_after.float() * rnd.keep_mask.float()).mean().item() - 0.25) < 5e-3  # independent masks


def test_full_size_properties():
    """BASELINE size (B=512, L=50, H=64, 2 heads): properties that need no oracle."""
    B, L, H, nh = 512, 50, 64, 2
    g = torch.Generator().manual_seed(42)
    q, k, v, qa, ka = (torch.randn(B, L, H, generator=g).to(DEV) for _ in range(5))
    gl = torch.randn(B, L, L, generator=g).to(DEV)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    noise = torch.randn(B, nh, L, L, generator=g).to(DEV)
    w = lambda *s: (0.3 * torch.randn(*s, generator=g)).to(DEV)
    kw = dict(w_order=w(1, 64), b_order=w(1), w_dist=w(1, 64), b_dist=w(1), scalar=w(1))
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    rnd = A.ExplicitRandomness(noise=noise)
    smask = A.StructuredMask(kv, causal=True)
    ctx_a, ctx_c, M, probs = A.calibrated_attention(q, k, v, qa, ka, gl, smask, cfg, rnd=rnd, want_probs=True, **kw)
    for t in (ctx_a, ctx_c, M):
        assert torch.isfinite(t).all()
    for name in ("after_spatial", "before_spatial", "perturbed_attention", "calibrated_attention"):
        assert (probs[name].sum(-1) - 1).abs().max() <= 1e-5, name
    assert (M.sum(-1) - 1).abs().max() <= 1e-5
    # causal + padding structure: no mass on future or padded keys
    future = torch.triu(torch.ones(L, L, device=DEV), diagonal=1).bool()
    assert M[:, :, future].abs().max() == 0
    pad = (kv == 0)[:, None, None, :].expand(B, nh, L, L)
    assert probs["calibrated_attention"][pad].abs().max() == 0
    # structured mask == the reference's dense mask
    dense = smask.dense().contiguous()
    ctx_a2, ctx_c2, M2, _ = A.calibrated_attention(q, k, v, qa, ka, gl, dense, cfg, rnd=rnd, **kw)
    assert (M2 - M).abs().max() <= 1e-6 and (ctx_c2 - ctx_c).abs().max() <= 1e-5 and (ctx_a2 - ctx_a).abs().max() <= 1e-5
    # contexts are linear in V
    v2 = torch.randn(B, L, H, generator=g).to(DEV)
    _, c1, _, _ = A.calibrated_attention(q, k, v2, qa, ka, gl, smask, cfg, rnd=rnd, **kw)
    _, c12, _, _ = A.calibrated_attention(q, k, v + v2, qa, ka, gl, smask, cfg, rnd=rnd, **kw)
    assert (c12 - (ctx_c + c1)).abs().max() <= 1e-4
    # first row of every sequence attends to key 0 only
    assert (ctx_c[:, 0, :] - v[:, 0, :]).abs().max() <= 1e-5


def test_error_behaviour_matches_reference():
    x = torch.zeros(2, 50, 64, device=DEV)
    gl = torch.zeros(2, 50, 50, device=DEV)
    m = A.StructuredMask(torch.ones(2, 50, dtype=torch.uint8, device=DEV))
    with pytest.raises(ValueError):  # layers.py:618-622
        A.AttackRTransformerEncoder(n_heads=3, hidden_size=64)
    with pytest.raises(KeyError):  # layers.py:894-895
        A.calibrated_attention(x, x, x, x, x, gl, m, A.AttentionConfig(n_heads=2, combine_option="nope"),
                               rnd=A.ExplicitRandomness(noise=torch.zeros(2, 2, 50, 50, device=DEV)))
    with pytest.raises(RuntimeError):  # gate built for seq_length 50 fed L = 40 (SURVEY 8c quirk i)
        enc = A.AttackRTransformerEncoder(combine_option="gate", seq_length=50).to(DEV).eval()
        enc(torch.zeros(2, 40, 64, device=DEV), torch.zeros(2, 1, 40, 40, device=DEV))
    x256 = torch.zeros(2, 50, 256, device=DEV)
    with pytest.raises(_lib.AcattnError):  # head size 256 is outside the supported set {16, 32, 64, 128}
        A.calibrated_attention(x256, x256, x256, x256, x256, gl, m, A.AttentionConfig(n_heads=1), seed=1)


@pytest.fixture
def forward_kernel():
    """Pins the forward kernel (acattn_select_forward_kernel) for one test and restores the automatic choice."""
    lib = _lib.load()
    yield lambda which: lib.acattn_select_forward_kernel(which)
    lib.acattn_select_forward_kernel(_lib.FWD_AUTO)


# (B, L, H, heads, kernel): the LDS-staged kernels (L <= 64: acattn_fwd_dma.hip for 48 < L, acattn_fwd_fast.hip below)
# and the streaming kernel (acattn_fwd_stream.hip, L <= 208).  (512, 50, 64, 2) is the shape bench.py times.
_TUNED_CASES = [(96, 50, 64, 2, _lib.FWD_STAGED), (24, 50, 64, 4, _lib.FWD_STAGED), (16, 64, 128, 2, _lib.FWD_STAGED),
                (8, 37, 64, 2, _lib.FWD_STAGED), (512, 50, 64, 2, _lib.FWD_STAGED),
                (96, 50, 64, 2, _lib.FWD_STREAM), (24, 50, 64, 4, _lib.FWD_STREAM), (16, 64, 128, 2, _lib.FWD_STREAM),
                (8, 37, 64, 2, _lib.FWD_STREAM), (512, 50, 64, 2, _lib.FWD_STREAM),
                (6, 200, 128, 4, _lib.FWD_STREAM), (4, 200, 64, 2, _lib.FWD_STREAM), (3, 130, 256, 4, _lib.FWD_STREAM),
                (5, 77, 64, 4, _lib.FWD_STREAM)]


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("p_drop", [0.0, 0.5])
@pytest.mark.parametrize("case", _TUNED_CASES, ids=lambda c: "B%d_L%d_H%d_h%d_k%d" % c)
def test_fast_training_kernel_equals_general_kernel(causal, p_drop, case, forward_kernel):
    """The tuned training kernels (counter RNG, gate, structured mask) against the general kernel fed the same draws
    as explicit tensors (itself pinned to the oracle above): contexts and M.  Includes the exact shape, mask,
    dropout rate and length distribution of the benchmark (B=512, L=50, H=64, 2 heads)."""
    B, L, H, nh, which = case
    forward_kernel(which)
    g = torch.Generator().manual_seed(7)
    q, k, v, qa, ka = (torch.randn(B, L, H, generator=g).to(DEV) for _ in range(5))
    gl = torch.randn(B, L, L, generator=g).to(DEV)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8)
    kv[0] = 1 - kv[0]  # one left-padded sequence: fully masked leading rows
    kv = kv.to(DEV)
    w = lambda *s: (0.3 * torch.randn(*s, generator=g)).to(DEV)
    dh = H // nh
    kw = dict(w_order=w(1, 2 * dh), b_order=w(1), w_dist=w(1, 2 * dh), b_dist=w(1), scalar=w(1))
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv, causal=causal)
    seed = 991
    fast = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=p_drop, seed=seed, **kw)
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    if p_drop == 0.0:
        rnd = A.ExplicitRandomness(noise=rnd.noise)
    ref = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=p_drop, rnd=rnd, **kw)
    # Sequence 0 is left-padded: under the causal mask its leading rows have NO unmasked key, every score
    # then carries the additive -10000 and the softmax sees fp32-quantised inputs (ulp(1e4) ~ 1e-3).  The two
    # kernels quantise in different domains (natural vs exp2), so those rows agree only to ~1e-3; they are rows
    # of padding queries whose outputs no consumer reads.  Everything else must agree tightly.
    sl = slice(1, None)
    assert (fast[2][sl] - ref[2][sl]).abs().max() <= 5e-6  # M
    assert (fast[0][sl] - ref[0][sl]).abs().max() <= 5e-5  # ctx_attacked
    assert (fast[1][sl] - ref[1][sl]).abs().max() <= 5e-5  # ctx_calibrated
    for a, b_ in zip(fast[:3], ref[:3]):
        assert (a[0] - b_[0]).abs().max() <= 2e-3
    # spatial-only flavour
    scfg = A.AttentionConfig(n_heads=nh, adversarial=False)
    f2 = A.calibrated_attention(q, k, v, None, None, None, mask, scfg, p_drop=p_drop, seed=seed, **kw)
    r2 = A.calibrated_attention(q, k, v, None, None, None, mask, scfg, p_drop=p_drop,
                                rnd=A.ExplicitRandomness(keep_after=rnd.keep_after) if p_drop > 0 else A.ExplicitRandomness(), **kw)
    assert (f2[1][sl] - r2[1][sl]).abs().max() <= 5e-5
    assert (f2[1][0] - r2[1][0]).abs().max() <= 2e-3
