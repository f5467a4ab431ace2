"""GPU suite (-m gpu): the kernels bench.py TIMES, at the shape it times them, against the ORACLE in one hop.

Round 2 pinned the tuned kernels to the general HIP kernels at B = 512 and the general kernels to the oracle on the
B <= 4 goldens.  Here the streaming forward (recbole/model/layers.py:686-742, 657-674, 898-951 fused) and the tuned
backward kernels (row-resident, one-row, mask-only split; the trainer's cotangent patterns of
recbole/trainer/trainer.py:672-686) run at B = 512, L = 50, H = 64, 2 heads, causal, p_drop 0.5, item_length ~
U{1..L} -- and at BASELINE configs[3]'s shape through the streaming backward -- and are compared with
oracle/ac_tsr_ref.py::core_from_projected (and its autograd) fed the kernels' own counter-RNG draws
(acattn_rng_materialize).  Also: the producer's affine planes / gate probabilities (ABI 26) in the forward kernel, in
the projections launch and in acattn_spatial_affines.

Tolerances: M 5e-6, contexts 1e-4 (north_star: logits within 1e-4); gradients 2e-3 of each tensor's largest magnitude
(the oracle differentiates the materialised formulation, the kernels the rank-1 / log-normaliser formulation)."""
import ctypes as C
import math

import pytest
import torch

import ac_tsr_amd as A
from ac_tsr_amd import _lib, linear, ops
from ac_tsr_amd.state import StepState
from oracle import ac_tsr_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _problem(B, L, H, nh, seed, causal=True, scale=0.3, left_pad=False):
    g = torch.Generator().manual_seed(seed)
    t = {k: torch.randn(B, L, H, generator=g) for k in ("q", "k", "v", "qa", "ka")}
    t["gl"] = torch.randn(B, L, L, generator=g)
    dh = H // nh
    for k, shp in (("w_order", (1, 2 * dh)), ("b_order", (1,)), ("w_dist", (1, 2 * dh)), ("b_dist", (1,)), ("scalar", (1,))):
        t[k] = scale * torch.randn(*shp, generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8)
    if left_pad:
        kv[1] = 1 - kv[1]
    return t, kv, lens, g


def _oracle_mask(kv, causal):
    B, L = kv.shape
    item_seq = kv.to(torch.int64)  # any non-zero id is a real item (abstract_recommender.py:137)
    return O.attention_mask(item_seq, bidirectional=not causal)


def _affine_planes(t, nh):
    """The producer's affine planes, in fp64 on the host (include/acattn.h: acattn_problem.affine)."""
    B, L, H = t["q"].shape
    dh = H // nh
    LP = 16 * ((L + 15) // 16)
    q = t["q"].double().view(B, L, nh, dh).permute(0, 2, 1, 3)
    k = t["k"].double().view(B, L, nh, dh).permute(0, 2, 1, 3)
    wo, wd = t["w_order"].double().reshape(-1), t["w_dist"].double().reshape(-1)
    l2e = 1.0 / math.log(2.0)
    planes = torch.zeros(B, nh, 4, LP, dtype=torch.float64)
    planes[:, :, 0, :L] = -l2e * (q @ wo[:dh] + t["b_order"].double())
    planes[:, :, 1, :L] = q @ wd[:dh] + t["b_dist"].double()
    planes[:, :, 2, :L] = -l2e * (k @ wo[dh:])
    planes[:, :, 3, :L] = k @ wd[dh:]
    return planes.float()


@pytest.mark.parametrize("B,L,H,nh,causal", [(512, 50, 64, 2, True), (512, 50, 64, 2, False), (8, 200, 128, 4, True),
                                             (4, 200, 256, 2, False), (6, 50, 256, 2, True), (128, 200, 128, 2, True)],
                         ids=["bench_shape", "bench_shape_bidirectional", "cfg4_shape", "dh128_L200_bidirectional", "dh128_L50",
                              "dh64_L200_spilling"])
@pytest.mark.parametrize("extras", [False, True], ids=["in_kernel", "producer_extras"])
def test_streaming_forward_matches_oracle_in_one_hop(B, L, H, nh, causal, extras):
    """id dh64_L200_spilling [r4]: head size 64, L > 64, in-kernel affines is the one instantiation of the streaming forward
    that keeps registers in scratch (33; tests/test_abi_cpu.py pins that set); 128 x 2 x 13 = 3,328 waves are more than two
    per SIMD, the occupancy at which round 3's work-in-progress builds went wrong."""
    t, kv, lens, g = _problem(B, L, H, nh, seed=101, causal=causal)
    dev = {k: v.to(DEV) for k, v in t.items()}
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv.to(DEV), causal=causal)
    kw = {k: dev[k] for k in ("w_order", "b_order", "w_dist", "b_dist", "scalar")}
    seed, p_drop = 20251, 0.5
    gate = dev["gl"]
    if extras:
        kw.update(affine=_affine_planes(t, nh).to(DEV), gate_is_prob=True)
        gate = torch.sigmoid(dev["gl"])
    lib = _lib.load()
    lib.acattn_select_forward_kernel(_lib.FWD_STREAM)
    try:
        ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], gate, mask, cfg,
                                                    p_drop=p_drop, seed=seed, **kw)
    finally:
        lib.acattn_select_forward_kernel(_lib.FWD_AUTO)
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    ocfg = O.EncoderCfg(n_layers=1, n_heads=nh, hidden_size=H, inner_size=4 * H, combine_option="gate", seq_length=L,
                        attn_dropout_prob=p_drop)
    with torch.no_grad():
        ref = O.core_from_projected(t["q"], t["k"], t["v"], t["qa"], t["ka"], t["gl"], _oracle_mask(kv, causal),
                                    t["w_order"], t["b_order"], t["w_dist"], t["b_dist"], t["scalar"], ocfg,
                                    rnd.noise.cpu(), keep_after=rnd.keep_after.cpu().float(),
                                    keep_mask=rnd.keep_mask.cpu().float(), materialize=False)
    assert (M.cpu() - ref["M"]).abs().max().item() <= 5e-6
    assert (ctx_a.cpu() - ref["ctx_attacked"]).abs().max().item() <= 1e-4
    assert (ctx_c.cpu() - ref["ctx_calibrated"]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("pattern", ["all", "calibrated_read_row", "attacked_read_row_plus_mask", "calibrated_dense_only",
                                     "calibrated_dense_plus_mask"])
@pytest.mark.parametrize("B,L,H,nh", [(512, 50, 64, 2), (8, 200, 128, 4), (3, 200, 256, 2), (5, 50, 256, 2)],
                         ids=["bench_shape", "cfg4_shape", "dh128_L200", "dh128_L50"])
def test_tuned_backward_matches_oracle_autograd_in_one_hop(B, L, H, nh, pattern):
    """`all`: every cotangent dense (the first layer's calibrated pass; row-resident kernel at L = 50, the streaming
    pair at L = 200).  `calibrated_read_row`: one context row per sequence (the last layer, pass 1: one-row kernel).
    `attacked_read_row_plus_mask`: the attacked context at the read row + a dense mask cotangent (the last layer, pass 2:
    mask-only blocks + the one-row chain; at L = 200 [r4] the streaming pair with mask-only query blocks).
    `calibrated_dense_only` / `calibrated_dense_plus_mask` [r4]: every row of the calibrated context (+ the mask) carries a
    cotangent and the attacked context NONE -- a layer that is not the last, in pass 1 / pass 2: at L > 64 the streaming
    pair's form without perturbed attention and without the Gaussian draws (acattn_bwd_row_kernel<.., false>)."""
    t, kv, lens, g = _problem(B, L, H, nh, seed=202)
    seed, p_drop = 777, 0.5
    rows = (lens - 1).view(-1, 1)
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    cot = {"ctx_attacked": torch.zeros(B, L, H), "ctx_calibrated": torch.zeros(B, L, H), "M": torch.zeros(B, nh, L, L)}
    row_cot = torch.randn(B, 1, H, generator=g)
    idx = rows.unsqueeze(-1).expand(-1, -1, H)
    if pattern == "all":
        cot = {k: torch.randn(v.shape, generator=g) for k, v in cot.items()}
    elif pattern.startswith("calibrated_dense"):
        cot["ctx_calibrated"] = torch.randn(B, L, H, generator=g)
        if pattern.endswith("plus_mask"):
            cot["M"] = torch.randn(B, nh, L, L, generator=g) * 1e-2
    elif pattern == "calibrated_read_row":
        cot["ctx_calibrated"].scatter_(1, idx, row_cot)
    else:
        cot["ctx_attacked"].scatter_(1, idx, row_cot)
        cot["M"] = torch.randn(B, nh, L, L, generator=g) * 1e-3  # the penalty's cotangent is small and dense
    names = ["q", "k", "v", "qa", "ka", "gl", "w_order", "b_order", "w_dist", "b_dist", "scalar"]
    cpu = {k: t[k].clone().requires_grad_(True) for k in names}
    ocfg = O.EncoderCfg(n_layers=1, n_heads=nh, hidden_size=H, inner_size=4 * H, combine_option="gate", seq_length=L,
                        attn_dropout_prob=p_drop)
    ref = O.core_from_projected(cpu["q"], cpu["k"], cpu["v"], cpu["qa"], cpu["ka"], cpu["gl"], _oracle_mask(kv, True),
                                cpu["w_order"], cpu["b_order"], cpu["w_dist"], cpu["b_dist"], cpu["scalar"], ocfg,
                                rnd.noise.cpu(), keep_after=rnd.keep_after.cpu().float(),
                                keep_mask=rnd.keep_mask.cpu().float(), materialize=False)
    want = dict(zip(names, torch.autograd.grad(sum((ref[k] * cot[k]).sum() for k in cot), [cpu[k] for k in names])))

    dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv.to(DEV), causal=True)
    read_rows = None if pattern == "all" or pattern.startswith("calibrated_dense") else rows.to(DEV)
    ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask, cfg,
                                                p_drop=p_drop, seed=seed, read_rows=read_rows,
                                                **{k: dev[k] for k in names[6:]})
    # (an output that is not part of the loss has NO cotangent -- None, not zeros -- which is what the trainer's walks hand
    # the attention node: the kernels specialise on it)
    used = [k for k in ("ctx_attacked", "ctx_calibrated", "M") if pattern == "all" or bool((cot[k] != 0).any())]
    outs = {"ctx_attacked": ctx_a, "ctx_calibrated": ctx_c, "M": M}
    loss = sum((outs[k] * cot[k].to(DEV)).sum() for k in used)
    got = dict(zip(names, torch.autograd.grad(loss, [dev[k] for k in names])))
    for k in names:
        scale = want[k].abs().max().item()
        err = (got[k].cpu() - want[k]).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (k, err, scale)


@pytest.mark.parametrize("B,L,H,nh", [(6, 50, 64, 2), (3, 37, 64, 4), (2, 200, 128, 4), (2, 130, 256, 4)])
def test_spatial_affines_entry_point(B, L, H, nh):
    t, kv, lens, g = _problem(B, L, H, nh, seed=5)
    dev = {k: v.to(DEV).contiguous() for k, v in t.items()}
    p = _lib.Problem()
    p.B, p.L, p.H, p.n_heads = B, L, H, nh
    p.q, p.k = dev["q"].data_ptr(), dev["k"].data_ptr()
    wo, wd = dev["w_order"].reshape(-1).contiguous(), dev["w_dist"].reshape(-1).contiguous()
    p.w_order, p.b_order, p.w_dist, p.b_dist = wo.data_ptr(), dev["b_order"].data_ptr(), wd.data_ptr(), dev["b_dist"].data_ptr()
    LP = 16 * ((L + 15) // 16)
    out = torch.full((B, nh, 4, LP), float("nan"), device=DEV)
    _lib.check(_lib.load().acattn_spatial_affines(C.byref(p), out.data_ptr(), ops._stream()), "spatial_affines")
    want = _affine_planes(t, nh)
    assert torch.isfinite(out).all()  # the padding entries are written (zeros)
    assert (out.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,L,nh,H", [(32, 50, 2, 64), (5, 37, 4, 64), (512, 50, 2, 64), (3, 200, 1, 64),
                                      (32, 100, 2, 128), (5, 37, 4, 128), (7, 50, 8, 128), (9, 100, 4, 256), (3, 37, 2, 256)])
def test_projections_launch_hands_over_affine_planes_and_gate_probabilities(B, L, nh, H):
    """acattn_projections_fwd with acattn_proj_out.affine / .gate_prob: the planes from the rows still in its
    accumulators, sigmoid(gate) in place of the logits; everything else and every gradient as without them (the gate
    cotangent that comes back is the gradient of the LOGITS: include/acattn.h)."""
    g = torch.Generator().manual_seed(B + L)
    r = lambda *s: torch.randn(*s, generator=g)
    t = dict(x=r(B, L, H))
    ws = 0.2 * (64 / H) ** 0.5
    for n in ("q", "k", "v", "aq", "ak"):
        t["w" + n], t["b" + n] = ws * r(H, H), 0.1 * r(H)
    t["wg"], t["bg"] = ws * r(L, H), 0.1 * r(L)
    dh = H // nh
    sp = dict(w_order=0.3 * r(1, 2 * dh), b_order=0.3 * r(1), w_dist=0.3 * r(1, 2 * dh), b_dist=0.3 * r(1))
    W = ("wq", "bq", "wk", "bk", "wv", "bv", "waq", "baq", "wak", "bak", "wg", "bg")
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    spd = tuple(v.to(DEV) for v in sp.values()) + (nh,)
    plain = linear._FusedProjections.apply(dev["x"], *(dev[n] for n in W), True, StepState())
    extra = linear._FusedProjections.apply(dev["x"], *(dev[n] for n in W), True, StepState(), spd)
    assert plain[7] is None and extra[7] is not None
    for a, b in zip(plain[:5], extra[:5]):
        assert torch.equal(a, b)
    assert (extra[5] - torch.sigmoid(plain[5])).abs().max().item() <= 2e-6
    mq, mk = plain[0].detach().cpu(), plain[1].detach().cpu()
    want = _affine_planes(dict(q=mq, k=mk, **sp), nh)
    got = extra[7].cpu()
    assert (got - want).abs().max().item() <= 3e-5 * max(1.0, want.abs().max().item())
    assert got[..., L:].abs().max().item() == 0.0 if got.shape[-1] > L else True
    cot = [r(B, L, H).to(DEV) for _ in range(5)] + [r(B, L, L).to(DEV)]
    names = ["x"] + list(W)
    ga = torch.autograd.grad(sum((o * c).sum() for o, c in zip(plain[:6], cot)), [dev[n] for n in names])
    gb = torch.autograd.grad(sum((o * c).sum() for o, c in zip(extra[:6], cot)), [dev[n] for n in names])
    for n, a, b in zip(names, ga, gb):
        assert torch.equal(a, b), n


def test_encoder_layer_equals_itself_without_producer_extras():
    """One AttackRTransformerLayer, training mode, counter RNG: with the projections launch handing the planes and the
    gate probabilities to the core (the default) and with the core deriving both itself -- outputs and gradients."""
    torch.manual_seed(3)
    B, L, H = 64, 50, 64
    layer = A.AttackRTransformerLayer(n_heads=2, hidden_size=H, intermediate_size=256, hidden_dropout_prob=0.0,
                                      attn_dropout_prob=0.5, hidden_act="gelu", layer_norm_eps=1e-12, combine_option="gate",
                                      use_order=True, use_distance=True, two_level=True, rich_calibrated_combine="none",
                                      seq_length=L).to(DEV).train()
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "affine" in n or "scalar" in n:
                p.copy_(0.3 * torch.randn_like(p))
    x = torch.randn(B, L, H, device=DEV, requires_grad=True)
    lens = torch.randint(1, L + 1, (B,))
    mask = A.StructuredMask((torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV), causal=True)
    res = []
    for on in (True, False):
        linear.PRODUCER_EXTRAS = on
        try:
            torch.manual_seed(11)  # the kernels' seeds are drawn from torch's CPU generator
            att, cal, M, _ = layer(x, mask)
            # random cotangents (sums of squares of LayerNorm outputs are nearly constant: their gradients are noise)
            gc = torch.Generator().manual_seed(5)
            c1, c2, c3 = (torch.randn(t_.shape, generator=gc).to(DEV) for t_ in (att, cal, M))
            loss = (att * c1).sum() + (cal * c2).sum() + (M * c3).sum()
            res.append((att, cal, M, torch.autograd.grad(loss, [x] + list(layer.parameters()))))
        finally:
            linear.PRODUCER_EXTRAS = True
    for a, b in zip(res[0][:3], res[1][:3]):
        assert (a - b).abs().max().item() <= 2e-5
    for (n, _), a, b in zip([("x", None)] + list(layer.named_parameters()), res[0][3], res[1][3]):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-5, n


def test_dispatcher_operators_equal_the_direct_c_abi_path():
    """torch.ops.acattn.calibrated_attention_fwd / _bwd (ac_tsr_amd/dispatch.py): the raw operator with its own autograd
    formula, the autograd node routed through it (the default) and the node on the direct C-ABI path give the same
    outputs and gradients; the operator passes torch.library.opcheck's schema / fake-tensor checks."""
    from ac_tsr_amd import dispatch  # noqa: F401
    B, L, H, nh = 24, 50, 64, 2
    t, kv, lens, g = _problem(B, L, H, nh, seed=31)
    names = ["q", "k", "v", "qa", "ka", "gl", "w_order", "b_order", "w_dist", "b_dist", "scalar"]
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv.to(DEV), causal=True)
    cot = [torch.randn(B, L, H, generator=g).to(DEV), torch.randn(B, L, H, generator=g).to(DEV),
           torch.randn(B, nh, L, L, generator=g).to(DEV)]
    seed, p_drop = 4711, 0.5

    def via_node(use_dispatcher):
        ops.USE_DISPATCHER = use_dispatcher
        try:
            dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
            out = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask, cfg, p_drop=p_drop,
                                         seed=seed, **{k: dev[k] for k in names[6:]})
            loss = sum((o * c).sum() for o, c in zip(out[:3], cot))
            return out[:3], torch.autograd.grad(loss, [dev[k] for k in names])
        finally:
            ops.USE_DISPATCHER = True

    def via_raw_op():
        dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
        out = torch.ops.acattn.calibrated_attention_fwd(
            dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask.key_valid, True, dev["w_order"].reshape(-1),
            dev["b_order"], dev["w_dist"].reshape(-1), dev["b_dist"], dev["scalar"], nh, p_drop, seed, None, False, None, True)
        loss = sum((o * c).sum() for o, c in zip(out[:3], cot))
        return out[:3], torch.autograd.grad(loss, [dev[k] for k in names])

    ref_out, ref_g = via_node(False)
    for got_out, got_g in (via_node(True), via_raw_op()):
        for a, b in zip(got_out, ref_out):
            assert torch.equal(a, b)
        for n, a, b in zip(names, got_g, ref_g):
            assert (a.reshape(b.shape) - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-9, n
    dev = {k: t[k].to(DEV) for k in names}
    torch.library.opcheck(torch.ops.acattn.calibrated_attention_fwd.default,
                          (dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask.key_valid, True,
                           dev["w_order"].reshape(-1), dev["b_order"], dev["w_dist"].reshape(-1), dev["b_dist"], dev["scalar"],
                           nh, p_drop, seed, None, False, None, True),
                          test_utils=("test_schema", "test_faketensor"))


def test_dispatcher_operators_reject_what_the_c_abi_would_misread():
    """The operators hand raw pointers to the C ABI (fp32 / uint8, exact shapes): a half-precision activation (autocast), an
    int64 validity mask, a gate or affine of another length, a tensor on the host must raise, not be reinterpreted
    (ADVICE r3).  The evaluation form (want_penalty=False) returns an empty penalty tensor."""
    from ac_tsr_amd import dispatch  # noqa: F401
    B, L, H, nh = 4, 50, 64, 2
    t, kv, lens, g = _problem(B, L, H, nh, seed=9)
    dev = {k: v.to(DEV) for k, v in t.items()}
    kvd = kv.to(DEV)

    def call(**over):
        a = dict(q=dev["q"], k=dev["k"], v=dev["v"], qa=dev["qa"], ka=dev["ka"], gate=dev["gl"], key_valid=kvd, affine=None,
                 w_order=dev["w_order"].reshape(-1), want_penalty=True)
        a.update(over)
        return torch.ops.acattn.calibrated_attention_fwd(
            a["q"], a["k"], a["v"], a["qa"], a["ka"], a["gate"], a["key_valid"], True, a["w_order"], dev["b_order"],
            dev["w_dist"].reshape(-1), dev["b_dist"], dev["scalar"], nh, 0.5, 11, None, False, a["affine"], True, a["want_penalty"])

    out = call()
    assert out[4].shape == (B, nh, 4)
    assert call(want_penalty=False)[4].numel() == 0
    with pytest.raises(TypeError):
        call(q=dev["q"].bfloat16())
    with pytest.raises(TypeError):
        call(qa=dev["qa"].half())
    with pytest.raises(TypeError):
        call(key_valid=kvd.long())
    with pytest.raises(ValueError):
        call(gate=dev["gl"][:, :, :48].contiguous())
    with pytest.raises(ValueError):
        call(key_valid=kvd[:, :48].contiguous())
    with pytest.raises(ValueError):
        call(affine=torch.zeros(B, nh, 4, 48, device=DEV))
    with pytest.raises(ValueError):
        call(w_order=dev["w_order"].reshape(-1)[:10].contiguous())
    with pytest.raises(ValueError):
        call(k=dev["k"].transpose(0, 1).contiguous().transpose(0, 1))
    with pytest.raises(_lib.AcattnError):
        call(v=t["v"])
    M, stats = out[2], out[3]
    with pytest.raises(TypeError):
        torch.ops.acattn.calibrated_attention_bwd(
            dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], kvd, True, dev["w_order"].reshape(-1), dev["b_order"],
            dev["w_dist"].reshape(-1), dev["b_dist"], dev["scalar"], nh, 0.5, 11, None, False, M, stats, dev["q"].double(), None, None,
            None, None, False, None)
    with pytest.raises(ValueError):
        torch.ops.acattn.calibrated_attention_bwd(
            dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], kvd, True, dev["w_order"].reshape(-1), dev["b_order"],
            dev["w_dist"].reshape(-1), dev["b_dist"], dev["scalar"], nh, 0.5, 11, None, False, M[:, :1].contiguous(), stats, dev["q"], None,
            None, None, None, False, None)


@pytest.mark.parametrize("rich", ["fixed", "trainable"])
@pytest.mark.parametrize("B,L,H,nh,causal", [(3, 100, 64, 2, True), (2, 200, 128, 4, True), (2, 200, 64, 2, False), (3, 130, 64, 4, True)],
                         ids=["L100", "L200_cfg4_heads", "L200_bidirectional_atomics", "L130_dh16"])
def test_one_level_backward_beyond_64_matches_oracle_autograd(B, L, H, nh, causal, rich):
    """two_level = False (recbole/model/layers.py:911-914, 929-936: the origin attention is before_spatial, after_spatial
    enters through the final mix with ratio 0.5 or the trainable parameter) at L > 64 [round 3: rounds 1-2 built the
    backward of this variant for L <= 64 only], counter RNG with dropout, against the oracle's autograd."""
    t, kv, lens, g = _problem(B, L, H, nh, seed=L + nh, causal=causal)
    seed, p_drop = 99, 0.5
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    names = ["q", "k", "v", "qa", "ka", "gl", "w_order", "b_order", "w_dist", "b_dist", "scalar"]
    t["rich_ratio"] = torch.tensor([0.37])
    if rich == "trainable":
        names.append("rich_ratio")
    cpu = {k: t[k].clone().requires_grad_(True) for k in names}
    ocfg = O.EncoderCfg(n_layers=1, n_heads=nh, hidden_size=H, inner_size=4 * H, combine_option="gate", seq_length=L,
                        attn_dropout_prob=p_drop, two_level=False, rich_calibrated_combine=rich)
    ref = O.core_from_projected(cpu["q"], cpu["k"], cpu["v"], cpu["qa"], cpu["ka"], cpu["gl"], _oracle_mask(kv, causal),
                                cpu["w_order"], cpu["b_order"], cpu["w_dist"], cpu["b_dist"], cpu["scalar"], ocfg,
                                rnd.noise.cpu(), keep_after=rnd.keep_after.cpu().float(), keep_mask=rnd.keep_mask.cpu().float(),
                                keep_before=rnd.keep_before.cpu().float(), rich_ratio=cpu.get("rich_ratio"))
    cot = {k: torch.randn(ref[k].shape, generator=g) for k in ("ctx_attacked", "ctx_calibrated", "M")}
    want = dict(zip(names, torch.autograd.grad(sum((ref[k] * cot[k]).sum() for k in cot), [cpu[k] for k in names])))
    dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate", two_level=False, rich_calibrated_combine=rich)
    mask = A.StructuredMask(kv.to(DEV), causal=causal)
    ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask, cfg,
                                                p_drop=p_drop, seed=seed, **{k: dev[k] for k in names[6:]})
    assert (M.cpu() - ref["M"]).abs().max().item() <= 5e-6
    assert (ctx_a.cpu() - ref["ctx_attacked"]).abs().max().item() <= 1e-4
    assert (ctx_c.cpu() - ref["ctx_calibrated"]).abs().max().item() <= 1e-4
    loss = sum((o * cot[k].to(DEV)).sum() for o, k in ((ctx_a, "ctx_attacked"), (ctx_c, "ctx_calibrated"), (M, "M")))
    got = dict(zip(names, torch.autograd.grad(loss, [dev[k] for k in names])))
    bad = []
    for k in names:
        scale = want[k].abs().max().item()
        err = (got[k].cpu() - want[k]).abs().max().item()
        if err > 2e-3 * scale + 1e-7:
            bad.append((k, err, scale))
    assert not bad, bad


@pytest.mark.parametrize("B,L,H,nh,causal,read_row", [(512, 50, 64, 2, True, False), (512, 50, 64, 2, True, True),
                                                      (8, 200, 128, 4, True, False), (6, 200, 64, 2, False, True),
                                                      (4, 130, 256, 2, True, False)],
                         ids=["bench_shape", "bench_shape_read_row", "cfg4_shape", "L200_bidirectional_read_row", "dh128"])
def test_penalty_row_sums_carry_the_mask_penalty_gradient(B, L, H, nh, causal, read_row):
    """ops.PENALTY_ROWS: the attention node's last output pen = acattn_mask_penalty_rows(M) (sum (1 - M)^2 per sequence,
    head and query block).  || 1 - M ||_2 (acsasrec.py:131-137) taken from pen gives the loss value of torch.norm(1 - M)
    and -- through acattn_bwd_io.d_penalty_part, d M = 2 d_pen (M - 1) formed inside the backward kernels -- the gradients
    the dense mask cotangent gives; with a context cotangent at one read row per sequence as well (the trainer's attacked
    pass through the last layer: mask-only blocks + the one-row chain at L <= 64)."""
    t, kv, lens, g = _problem(B, L, H, nh, seed=L + H, causal=causal)
    names = ["q", "k", "v", "qa", "ka", "gl", "w_order", "b_order", "w_dist", "b_dist", "scalar"]
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv.to(DEV), causal=causal)
    rows = (lens - 1).view(-1, 1)
    row_cot = torch.randn(B, 1, H, generator=g).to(DEV)
    idx = rows.unsqueeze(-1).expand(-1, -1, H).to(DEV)
    res = []
    for via_rows in (False, True):
        dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
        ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask, cfg,
                                                    p_drop=0.5, seed=4242, read_rows=rows.to(DEV) if read_row else None,
                                                    **{k: dev[k] for k in names[6:]})
        pen = M._acattn_pen
        assert pen.shape == (B, nh, (L + 15) // 16)
        norm = torch.sqrt(pen.sum()) if via_rows else torch.norm(1 - M, p=2)
        loss = 0.03 * norm
        if read_row:
            loss = loss + (ctx_a.gather(1, idx) * row_cot).sum()
        res.append((norm.detach(), torch.autograd.grad(loss, [dev[k] for k in names])))
    assert abs(res[0][0].item() - res[1][0].item()) <= 2e-5 * res[0][0].item()
    for k, a, b in zip(names, res[0][1], res[1][1]):
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-4 * scale + 1e-9, (k, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("causal", [True, False], ids=["causal", "bidirectional"])
@pytest.mark.parametrize("p_drop", [0.5, 0.0])
def test_spatial_only_forward_with_producer_planes_matches_oracle_at_the_bench_shape(causal, p_drop):
    """BASELINE configs[1] as bench.py times it (roofline_spatial_only): acattn_fwd_stream_kernel<32,4,false,*,true> --
    spatial calibrator only (layers.py:705-740, contract A'), affine planes from the producer, B = 512, lengths ~ U{1..L}
    incl. a left-padded sequence: ctx = dropout(after_spatial) . V against the oracle, fed the kernel's own keep draws."""
    B, L, H, nh = 512, 50, 64, 2
    t, kv, lens, g = _problem(B, L, H, nh, seed=404, causal=causal, left_pad=True)
    dev = {k: v.to(DEV) for k, v in t.items()}
    cfg = A.AttentionConfig(n_heads=nh, adversarial=False)
    mask = A.StructuredMask(kv.to(DEV), causal=causal)
    kw = {k: dev[k] for k in ("w_order", "b_order", "w_dist", "b_dist", "scalar")}
    seed = 31337
    lib = _lib.load()
    lib.acattn_select_forward_kernel(_lib.FWD_STREAM)
    try:
        none, ctx, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], None, None, None, mask, cfg, p_drop=p_drop, seed=seed,
                                                 affine=_affine_planes(t, nh).to(DEV), **kw)
        _, ctx_in, _, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], None, None, None, mask, cfg, p_drop=p_drop, seed=seed, **kw)
    finally:
        lib.acattn_select_forward_kernel(_lib.FWD_AUTO)
    assert M is None and none is None
    ocfg = O.EncoderCfg(n_layers=1, n_heads=nh, hidden_size=H, inner_size=4 * H, combine_option="gate", seq_length=L,
                        attn_dropout_prob=p_drop)
    keep_after = None
    if p_drop > 0:
        keep_after = A.materialize_randomness(B, nh, L, seed, p_drop, DEV).keep_after.cpu().float()
    with torch.no_grad():
        zeros = torch.zeros(B, nh, L, L)
        after = O.core_from_projected(t["q"], t["k"], t["v"], t["q"], t["k"], t["gl"], _oracle_mask(kv, causal), t["w_order"],
                                      t["b_order"], t["w_dist"], t["b_dist"], t["scalar"], ocfg, zeros, keep_after=keep_after,
                                      materialize=False)["after"]
        v = O._heads(t["v"], nh).permute(0, 2, 1, 3)
        expect = O.context_only(after, v)
    assert (ctx.cpu() - expect).abs().max().item() <= 1e-4
    assert (ctx_in.cpu() - expect).abs().max().item() <= 1e-4


@pytest.mark.parametrize("where", ["q", "k", "v", "qa", "gate"])
@pytest.mark.parametrize("L,extras", [(50, True), (50, False), (200, True)], ids=["L50_extras", "L50_in_kernel", "L200_extras"])
def test_a_nan_in_the_inputs_reaches_the_outputs_of_the_streaming_forward(where, L, extras):
    """The reference's only failure detector is _check_nan on the losses (recbole/trainer/trainer.py:763-765).  The
    streaming forward's translation units are built with -fno-honor-nans (csrc/Makefile: it removes canonicalising
    v_max x, x in front of the row maxima); that must not let a non-finite activation turn into plausible numbers: a NaN
    in a VALID row of any input has to come out as a NaN in the context rows that row feeds."""
    B, H, nh = 64, 64, 2
    t, kv, lens, g = _problem(B, L, H, nh, seed=77)
    kv[:] = 1  # every position valid: the poisoned row is visible to itself and to every later query
    b, i = 5, 3
    if where == "gate":
        t["gl"][b, i, :] = float("nan")
    else:
        t[where][b, i, 7] = float("nan")
    dev = {k: v.to(DEV) for k, v in t.items()}
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    mask = A.StructuredMask(kv.to(DEV), causal=True)
    kw = {k: dev[k] for k in ("w_order", "b_order", "w_dist", "b_dist", "scalar")}
    gate = dev["gl"]
    if extras:
        kw.update(affine=torch.nan_to_num(_affine_planes(t, nh)).to(DEV) if where not in ("q", "k") else _affine_planes(t, nh).to(DEV),
                  gate_is_prob=True)
        gate = torch.sigmoid(dev["gl"])
    lib = _lib.load()
    lib.acattn_select_forward_kernel(_lib.FWD_STREAM)
    try:
        ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], gate, mask, cfg,
                                                    p_drop=0.0, seed=5, **kw)
    finally:
        lib.acattn_select_forward_kernel(_lib.FWD_AUTO)
    torch.cuda.synchronize()
    # head 0 holds column 7; query row i sees key i under the causal mask, so row i of the poisoned head is the witness
    if where in ("q", "k", "v", "gate"):
        assert torch.isnan(ctx_c[b, i, :H // nh]).any(), "NaN swallowed on the calibrated branch"
    if where in ("q", "k", "v"):
        assert torch.isnan(ctx_a[b, i, :H // nh]).any(), "NaN swallowed on the attacked branch"
    if where == "qa":
        assert torch.isnan(M[b, 0, i]).any() and torch.isnan(ctx_a[b, i, :H // nh]).any()
    # and nothing leaks into other sequences
    other = torch.ones(B, dtype=torch.bool)
    other[b] = False
    assert torch.isfinite(ctx_c[other.to(DEV)]).all() and torch.isfinite(ctx_a[other.to(DEV)]).all()
