"""GPU suite (-m gpu): the HIP backward through the C ABI against autograd of the oracle, and the
model-level two-pass gradients against the reference's golden gradients.

Tolerance: every gradient tensor within 2e-3 of its own max magnitude (fp32; the oracle differentiates
the materialised formulation, the kernel the rank-1 / log-normaliser formulation)."""
import types

import pytest
import torch

import ac_tsr_amd as A
from oracle import ac_tsr_ref as O
from tests._golden import Case
from tests.test_hip_forward import _build_model, _core_kwargs, _project

pytestmark = pytest.mark.gpu
DEV = "cuda"
BWD_CASES = ["enc_gate_init", "enc_gate_stress", "enc_gate_h4", "enc_fixed_dist", "enc_fixed_order_bidir",
             "enc_gate_bidir", "enc_plain", "enc_anneal", "enc_leftpad", "enc_L37_ragged", "enc_L200_h4",
             "enc_L200_d64_bidir", "enc_onelevel", "enc_onelevel_trainable"]


def _rel(a, b):
    return ((a - b).abs().max() / max(b.abs().max().item(), 1e-12)).item()


@pytest.mark.parametrize("name", BWD_CASES)
@pytest.mark.parametrize("drop", [False, True])
def test_core_backward_matches_oracle_autograd(name, drop):
    c = Case(name)
    cfg, P, mq, mk, mv, qa, ka, gl = _project(c)
    if drop and name.startswith("enc_L200"):
        pytest.skip("dropout variant covered at L <= 50")
    mask = c.t("in.mask")
    noise = c.t("in.noise.0")
    g = torch.Generator().manual_seed(11)
    shape = noise.shape
    keep_a = torch.empty(shape).bernoulli_(0.5, generator=g) if drop else None
    keep_m = torch.empty(shape).bernoulli_(0.5, generator=g) if drop else None
    keep_b = torch.empty(shape).bernoulli_(0.5, generator=g) if (drop and not cfg.two_level) else None
    leaves = {"q": mq, "k": mk, "v": mv, "qa": qa, "ka": ka}
    if gl is not None:
        leaves["gl"] = gl
    dh2 = 2 * cfg.hidden_size // cfg.n_heads
    zero = torch.zeros(1, dh2)
    small = {"w_order": P.get("attack_attention.order_affine.weight", zero), "b_order": P.get("attack_attention.order_affine.bias"),
             "w_dist": P.get("attack_attention.distance_affine.weight", zero), "b_dist": P.get("attack_attention.distance_affine.bias"),
             "scalar": P.get("attack_attention.scalar"), "rich_ratio": P.get("rich_calibrated_combine_ratio")}
    leaves.update({k: v for k, v in small.items() if v is not None and (k.startswith("w_") is False or True)})
    cpu = {k: v.detach().clone().requires_grad_(True) for k, v in leaves.items()}
    ref = O.core_from_projected(cpu["q"], cpu["k"], cpu["v"], cpu["qa"], cpu["ka"], cpu.get("gl"), mask, cpu["w_order"],
                                cpu.get("b_order"), cpu["w_dist"], cpu.get("b_dist"), cpu.get("scalar"), cfg, noise,
                                keep_after=keep_a, keep_mask=keep_m, keep_before=keep_b, rich_ratio=cpu.get("rich_ratio"))
    cot = {k: torch.randn(ref[k].shape, generator=g) for k in ("ctx_attacked", "ctx_calibrated", "M")}
    loss = sum((ref[k] * cot[k]).sum() for k in cot)
    names = [k for k in cpu]
    grads = dict(zip(names, torch.autograd.grad(loss, [cpu[k] for k in names], allow_unused=True)))

    dev = {k: v.detach().to(DEV).contiguous().requires_grad_(True) for k, v in leaves.items()}
    acfg = A.AttentionConfig(n_heads=cfg.n_heads, combine_option=cfg.combine_option, two_level=cfg.two_level,
                             rich_calibrated_combine=cfg.rich_calibrated_combine)
    rnd = A.ExplicitRandomness(noise=noise.to(DEV), keep_after=None if keep_a is None else keep_a.to(torch.uint8).to(DEV),
                               keep_mask=None if keep_m is None else keep_m.to(torch.uint8).to(DEV),
                               keep_before=None if keep_b is None else keep_b.to(torch.uint8).to(DEV))
    kw = {}
    if "rich_ratio" in dev:
        kw.update(rich_ratio=dev["rich_ratio"])
    if cfg.use_order:
        kw.update(w_order=dev["w_order"], b_order=dev["b_order"])
    if cfg.use_distance:
        kw.update(w_dist=dev["w_dist"], b_dist=dev["b_dist"], scalar=dev["scalar"])
    ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev.get("gl"),
                                                mask.to(DEV).contiguous(), acfg, p_drop=0.5 if drop else 0.0, rnd=rnd, **kw)
    dloss = (ctx_a * cot["ctx_attacked"].to(DEV)).sum() + (ctx_c * cot["ctx_calibrated"].to(DEV)).sum() + \
        (M * cot["M"].to(DEV)).sum()
    used = [k for k in names if (k in ("q", "k", "v", "qa", "ka", "gl")) or (k in kw)]
    got = dict(zip(used, torch.autograd.grad(dloss, [dev[k] for k in used], retain_graph=True)))
    for k in used:
        if grads[k] is None:
            continue
        assert _rel(got[k].cpu(), grads[k]) <= 2e-3, (k, _rel(got[k].cpu(), grads[k]))
    # second walk over the same graph (recbole/trainer/trainer.py:677,684): identical result
    again = dict(zip(used, torch.autograd.grad(dloss, [dev[k] for k in used])))
    for k in used:
        assert torch.equal(again[k], got[k]) or _rel(again[k], got[k]) <= 1e-5


def _rnds_for(c: Case, n_layers, train):
    out = []
    for r in c.layer_randomness(n_layers, train):
        ns = types.SimpleNamespace()
        for f in ("noise", "keep_after", "keep_before", "keep_mask", "keep_out_att", "keep_out_cal", "keep_ffn_att",
                  "keep_ffn_cal"):
            t = getattr(r, f)
            if t is None:
                setattr(ns, f, None)
            elif f == "noise":
                setattr(ns, f, t.to(DEV))
            else:
                setattr(ns, f, t.to(torch.uint8).to(DEV))
        out.append(ns)
    return out


@pytest.mark.parametrize("name", ["model_eval", "model_eval_stress", "model_train", "model_beauty"])
def test_two_pass_trainer_gradients_match_reference(name, prune_dead_work=True):
    """calculate_loss + the trainer's two backward passes: losses and every parameter gradient against
    the tensors the genuine reference produced (tests/golden/model_*.npz)."""
    c = Case(name)
    cfg, model = _build_model(c)
    model.step_state.prune_dead_work = prune_dead_work
    train = bool(int(c.raw["meta.train"]))
    model.train(train)
    batch = {k: v.to(DEV) for k, v in c.batch().items()}
    rnds = _rnds_for(c, cfg.enc.n_layers, train)
    keep_emb = c.t("in.keep_emb").to(DEV) if train else None
    model.zero_grad()
    att, cal = model.calculate_loss(batch, _rnds=rnds, _keep_emb=keep_emb)
    assert abs(att.item() - float(c.raw["out.att_loss"])) <= 1e-4
    assert abs(cal.item() - float(c.raw["out.cal_loss"])) <= 1e-4
    for n, p in model.named_parameters():
        p.requires_grad = not A.is_attack_param(n)
    cal.backward(retain_graph=True)
    for n, p in model.named_parameters():
        p.requires_grad = A.is_attack_param(n)
    att.backward()
    for n, p in model.named_parameters():
        p.requires_grad = True
    ref = c.grads()
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        # attack_key_transform.bias has a mathematically zero gradient (softmax is invariant to a per-query
        # shift); both sides hold only ~1e-10 cancellation noise there, hence the absolute floor
        err = (g.cpu() - ref[n]).abs().max().item()
        assert err <= 2e-3 * ref[n].abs().max().item() + 2e-8, (n, err, ref[n].abs().max().item())


def test_train_step_full_size_is_finite_and_learns():
    torch.manual_seed(0)
    cfgd = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.5, attn_dropout_prob=0.5,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03)
    model = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(5000)).to(DEV)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
    model.train()
    g = torch.Generator().manual_seed(1)
    B, L = 512, 50
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, 5000, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    # learnable toy target: the last item of the sequence
    target = ids[torch.arange(B), lens - 1]
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": target.to(DEV)}
    first = last = None
    for step in range(30):
        att, cal = trainer.train_step(batch)
        assert torch.isfinite(att) and torch.isfinite(cal)
        first = cal.item() if first is None else first
        last = cal.item()
    assert last < first - 0.5, (first, last)
    for n, p in model.named_parameters():
        assert torch.isfinite(p).all(), n


def test_trainer_step_leaves_reference_gradients_on_non_attack_parameters():
    """AttackSASRecTrainer.train_step (pruned two-pass backward) against the reference's golden gradients.  The
    calibrated loss does not depend on the noise draw, so every non-attack parameter must match even though the
    trainer lets the kernel draw its own noise; attack parameters must be non-zero."""
    c = Case("model_eval")
    cfg, model = _build_model(c)
    model.eval()  # golden taken in eval mode (no dropout)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="sgd", learning_rate=0.0), model)
    batch = {k: v.to(DEV) for k, v in c.batch().items()}
    trainer.optimizer.zero_grad(set_to_none=False)
    att, cal = model.calculate_loss(batch)
    cal.backward(retain_graph=True, inputs=trainer._others)
    att.backward(inputs=trainer._attack)
    ref = c.grads()
    for n, p in model.named_parameters():
        if A.is_attack_param(n):
            if n.endswith("weight"):
                assert p.grad is not None and p.grad.abs().max() > 0, n
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        err = (g.cpu() - ref[n]).abs().max().item()
        assert err <= 2e-3 * ref[n].abs().max().item() + 2e-8, (n, err)
    assert abs(cal.item() - float(c.raw["out.cal_loss"])) <= 1e-4


def test_graph_mode_matches_eager_and_redraws_randomness():
    """trainer.enable_graph: a captured step equals the eager step on the same batch/seed state, successive
    replays draw different noise, and training still reduces the loss."""
    torch.manual_seed(0)
    cfgd = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.0, attn_dropout_prob=0.5,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03)
    g = torch.Generator().manual_seed(1)
    B, L, N = 128, 50, 3000
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}
    model = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(N)).to(DEV)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
    trainer.enable_graph(batch)
    att1, cal1 = (t.clone() for t in trainer.train_step(batch))
    att2, cal2 = (t.clone() for t in trainer.train_step(batch))
    assert torch.isfinite(att1) and torch.isfinite(cal1)
    assert att1.item() != att2.item()  # new noise / dropout draw and updated weights
    first = cal1.item()
    for _ in range(40):
        att, cal = trainer.train_step(batch)
    assert cal.item() < first - 0.3, (first, cal.item())


@pytest.fixture
def backward_kernel():
    """Pins the backward kernel (acattn_select_backward_kernel) for one test and restores the automatic choice."""
    from ac_tsr_amd import _lib
    lib = _lib.load()
    yield lambda which: lib.acattn_select_backward_kernel(which)
    lib.acattn_select_backward_kernel(_lib.BWD_AUTO)


# (B, L, H, heads, kernel): 2 = row-resident tuned kernel (acattn_bwd_fast.hip, L <= 64), 1 = streaming two-kernel
# backward (acattn_bwd_stream.hip, L <= 208).  (512, 50, 64, 2) is the shape bench.py trains on.
_BWD_CASES = [(64, 50, 64, 2, 2), (12, 50, 64, 4, 2), (8, 64, 128, 2, 2), (6, 37, 64, 2, 2), (512, 50, 64, 2, 2),
              (64, 50, 64, 2, 1), (12, 50, 64, 4, 1), (8, 64, 128, 2, 1), (6, 37, 64, 2, 1),
              (4, 200, 128, 4, 1), (3, 200, 64, 2, 1), (2, 130, 256, 4, 1), (5, 77, 64, 4, 1)]


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("p_drop", [0.0, 0.5])
@pytest.mark.parametrize("case", _BWD_CASES, ids=lambda c: "B%d_L%d_H%d_h%d_k%d" % c)
def test_fast_training_backward_equals_general_backward(causal, p_drop, case, backward_kernel):
    """The tuned backward kernels (counter RNG, gate, structured mask) against the general backward fed the same draws
    as explicit tensors (itself pinned to the oracle's autograd above): every gradient of the fused operator."""
    B, L, H, nh, which = case
    backward_kernel(which)
    g = torch.Generator().manual_seed(21)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    base = {k: mk(B, L, H) for k in ("q", "k", "v", "qa", "ka")}
    base["gl"] = mk(B, L, L)
    dh = H // nh
    for k, shp in (("w_order", (1, 2 * dh)), ("b_order", (1,)), ("w_dist", (1, 2 * dh)), ("b_dist", (1,)), ("scalar", (1,))):
        base[k] = (0.3 * torch.randn(*shp, generator=g)).to(DEV)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    mask = A.StructuredMask(kv, causal=causal)
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    cot = [mk(B, L, H), mk(B, L, H), mk(B, nh, L, L)]
    seed = 4242

    def grads(rnd):
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        out = A.calibrated_attention(t["q"], t["k"], t["v"], t["qa"], t["ka"], t["gl"], mask, cfg, p_drop=p_drop,
                                     seed=None if rnd is not None else seed, rnd=rnd, w_order=t["w_order"],
                                     b_order=t["b_order"], w_dist=t["w_dist"], b_dist=t["b_dist"], scalar=t["scalar"])
        loss = sum((o * c_).sum() for o, c_ in zip(out[:3], cot))
        names = list(t)
        return dict(zip(names, torch.autograd.grad(loss, [t[n] for n in names])))

    fast = grads(None)
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    if p_drop == 0.0:
        rnd = A.ExplicitRandomness(noise=rnd.noise)
    ref = grads(rnd)
    for n in ref:
        scale = ref[n].abs().max().item()
        assert (fast[n] - ref[n]).abs().max().item() <= 2e-4 * scale + 1e-7, (n, (fast[n] - ref[n]).abs().max().item(), scale)


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("p_drop", [0.0, 0.5])
@pytest.mark.parametrize("case", [(64, 50, 64, 2), (512, 50, 64, 2), (9, 37, 64, 4), (5, 200, 128, 4), (7, 64, 128, 2),
                                  (6, 130, 64, 4)], ids=lambda c: "B%d_L%d_H%d_h%d" % c)
def test_one_row_backward_equals_general_backward(causal, p_drop, case):
    """The calibrated-loss pass through the last layer: the context cotangent lives in ONE row per sequence (the read
    position, abstract_recommender.py:130-134).  `read_rows` with one position takes acattn_bwd_onerow_kernel; the
    reference here is the general backward fed the same draws explicitly and the same (mostly zero) cotangent."""
    B, L, H, nh = case
    g = torch.Generator().manual_seed(33)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    base = {k: mk(B, L, H) for k in ("q", "k", "v", "qa", "ka")}
    base["gl"] = mk(B, L, L)
    dh = H // nh
    for k, shp in (("w_order", (1, 2 * dh)), ("b_order", (1,)), ("w_dist", (1, 2 * dh)), ("b_dist", (1,)), ("scalar", (1,))):
        base[k] = (0.3 * torch.randn(*shp, generator=g)).to(DEV)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8)
    if B > 4:
        kv[1] = 0
        kv[1, L - 3:] = 1  # a left-padded sequence: rows before the first item are "dead" (uniform over every key)
    kv = kv.to(DEV)
    rows = (lens - 1).view(-1, 1).to(DEV)
    if B > 4:
        rows[1, 0] = 2  # ... and its read position is one of the dead rows
    mask = A.StructuredMask(kv, causal=causal)
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    cot = torch.zeros(B, L, H, device=DEV)
    cot.scatter_(1, rows.unsqueeze(-1).expand(-1, -1, H), mk(B, 1, H))
    seed = 777

    def grads(rnd, read_rows):
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        out = A.calibrated_attention(t["q"], t["k"], t["v"], t["qa"], t["ka"], t["gl"], mask, cfg, p_drop=p_drop,
                                     seed=None if rnd is not None else seed, rnd=rnd, read_rows=read_rows,
                                     w_order=t["w_order"], b_order=t["b_order"], w_dist=t["w_dist"], b_dist=t["b_dist"],
                                     scalar=t["scalar"])
        names = list(t)
        return dict(zip(names, torch.autograd.grad((out[1] * cot).sum(), [t[n] for n in names])))

    fast = grads(None, rows)
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    if p_drop == 0.0:
        rnd = A.ExplicitRandomness(noise=rnd.noise)
    ref = grads(rnd, None)
    # 5e-4 of the tensor's scale (a single row's gradients are small numbers, both kernels use the fast exp2) + 1e-6
    # absolute: the bias / scalar gradients are sums with cancellation whose own magnitude can be ~1e-4
    for n in ref:
        scale = ref[n].abs().max().item()
        assert (fast[n] - ref[n]).abs().max().item() <= 5e-4 * scale + 1e-6, (n, (fast[n] - ref[n]).abs().max().item(), scale)


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("p_drop", [0.0, 0.5])
@pytest.mark.parametrize("case", [(64, 50, 64, 2), (512, 50, 64, 2), (9, 37, 64, 4), (7, 64, 128, 2)],
                         ids=lambda c: "B%d_L%d_H%d_h%d" % c)
@pytest.mark.parametrize("which", [2, 0], ids=["row_resident", "auto_split"])
def test_mask_only_blocks_equal_general_backward(causal, p_drop, case, which, backward_kernel):
    """The attacked-loss pass through the last layer: the attacked context is read at one position per sequence, the
    mask penalty's cotangent reaches every row.  Query blocks without the read position take the mask-only path of the
    row-resident kernel; reference: the general backward with the same draws and the same cotangents, no hint.
    `auto_split`: the automatic choice runs the mask cotangent through the mask-only path for EVERY block and adds the
    read row's chain with the one-row kernel (the backward is linear in its cotangents)."""
    B, L, H, nh = case
    backward_kernel(which)
    g = torch.Generator().manual_seed(44)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    base = {k: mk(B, L, H) for k in ("q", "k", "v", "qa", "ka")}
    base["gl"] = mk(B, L, L)
    dh = H // nh
    for k, shp in (("w_order", (1, 2 * dh)), ("b_order", (1,)), ("w_dist", (1, 2 * dh)), ("b_dist", (1,)), ("scalar", (1,))):
        base[k] = (0.3 * torch.randn(*shp, generator=g)).to(DEV)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    rows = (lens - 1).view(-1, 1).to(DEV)
    mask = A.StructuredMask(kv, causal=causal)
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    cot_a = torch.zeros(B, L, H, device=DEV)
    cot_a.scatter_(1, rows.unsqueeze(-1).expand(-1, -1, H), mk(B, 1, H))
    cot_m = 0.01 * mk(B, nh, L, L)
    seed = 909

    def grads(rnd, read_rows):
        t = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        out = A.calibrated_attention(t["q"], t["k"], t["v"], t["qa"], t["ka"], t["gl"], mask, cfg, p_drop=p_drop,
                                     seed=None if rnd is not None else seed, rnd=rnd, read_rows=read_rows,
                                     w_order=t["w_order"], b_order=t["b_order"], w_dist=t["w_dist"], b_dist=t["b_dist"],
                                     scalar=t["scalar"])
        names = list(t)
        return dict(zip(names, torch.autograd.grad((out[0] * cot_a).sum() + (out[2] * cot_m).sum(), [t[n] for n in names])))

    fast = grads(None, rows)
    rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
    if p_drop == 0.0:
        rnd = A.ExplicitRandomness(noise=rnd.noise)
    ref = grads(rnd, None)
    for n in ref:
        scale = ref[n].abs().max().item()
        assert (fast[n] - ref[n]).abs().max().item() <= 5e-4 * scale + 1e-6, (n, (fast[n] - ref[n]).abs().max().item(), scale)


@pytest.mark.parametrize("name", ["model_eval", "model_train"])
def test_reference_schedule_switch_gives_the_same_gradients(name):
    """model.step_state.prune_dead_work = False (every tail on all positions, every layer's attacked branch, every
    input gradient: the reference's schedule) against the same golden vectors the default schedule is checked with."""
    test_two_pass_trainer_gradients_match_reference(name, prune_dead_work=False)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("graph", [False, True])
def test_data_parallel_code_path_matches_plain_trainer(graph, split):
    """The N > 1 code path on one GPU (GradSynchronizer: store-then-pack gradients, optimizer fed from the flat buffer,
    graph without the optimizer) updates the parameters exactly like the plain trainer given the same random state."""
    from ac_tsr_amd import parallel
    cfgd = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.1, attn_dropout_prob=0.2,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03)
    g = torch.Generator().manual_seed(1)
    B, L, N = 64, 50, 2000
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}
    states = []
    for use_sync in (False, True):
        torch.manual_seed(3)
        model = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(N)).to(DEV)
        sync = None
        if use_sync:  # with and without the early / late split (one graph or two)
            sync = parallel.GradSynchronizer.for_two_pass_model(model) if split else parallel.GradSynchronizer(model.parameters())
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model, grad_sync=sync)
        torch.manual_seed(5)  # the in-kernel RNG seeds are drawn from torch's CPU generator
        if graph:
            trainer.enable_graph(batch, warmup=1)  # (>= 1: Adam's state must exist before the capture)
        for _ in range(3):
            trainer.train_step(batch)
        torch.cuda.synchronize()
        states.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for k in states[0]:
        assert (states[0][k] - states[1][k]).abs().max() <= 1e-6 * max(1.0, states[0][k].abs().max().item()), k


def test_two_trainers_in_one_process_are_independent():
    """The pass identity and the replay seed counter are per-model state (state.StepState): a second model being
    trained in the same process, interleaved step by step and even from inside the other model's pass context, does
    not change what the first one learns."""
    cfgd = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.1, attn_dropout_prob=0.2,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03)
    g = torch.Generator().manual_seed(1)
    B, L, N = 32, 50, 500
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}

    def run(with_intruder):
        torch.manual_seed(3)
        model = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(N)).to(DEV)
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
        other = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(N)).to(DEV)
        other_trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), other)
        assert model.step_state is not other.step_state and trainer.state is model.step_state
        for step in range(3):
            torch.manual_seed(100 + step)
            if with_intruder:
                with other.step_state.attack_pass():  # somebody else's pass context must not leak into `model`
                    trainer.train_step(batch)
                cpu_rng = torch.get_rng_state()
                other_trainer.train_step(batch)
                torch.set_rng_state(cpu_rng)
            else:
                trainer.train_step(batch)
        torch.cuda.synchronize()
        assert model.step_state.pass_mode is None
        return {k: v.detach().clone() for k, v in model.state_dict().items()}

    alone, together = run(False), run(True)
    for k in alone:  # (float atomics in the embedding scatter: equal to rounding, not bit for bit)
        assert (alone[k] - together[k]).abs().max() <= 1e-6 * max(1.0, alone[k].abs().max().item()), k
    from ac_tsr_amd import state
    with pytest.raises(AttributeError):  # the default state of stand-alone modules is read-only
        state.DEFAULT.pass_mode = "attack"



@pytest.mark.parametrize("B,L", [(64, 50), (24, 130)])
def test_model_gradients_do_not_depend_on_the_backward_kernels(B, L, backward_kernel):
    """The two-pass gradients of the whole model (training mode, in-kernel randomness, pruned schedule) with the
    automatic kernel choice -- one-row backward for the last layer's calibrated pass, mask-only blocks and the
    mask + one-row split in the attacked pass, row-resident kernel elsewhere (L <= 64) -- against the same step with the
    streaming row / key kernels pinned, which share none of that code.  Same torch seed, hence the same kernel seeds."""
    cfgd = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.5, attn_dropout_prob=0.5,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03, MAX_ITEM_LIST_LENGTH=L,
                gate_seq_length=L)
    g = torch.Generator().manual_seed(5)
    N = 700
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}

    def grads(which):
        backward_kernel(which)
        torch.manual_seed(11)
        model = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(N)).to(DEV).train()
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
        torch.manual_seed(12)
        att = trainer._pass_one(batch)[0]
        trainer._pass_two(att)
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    auto, stream = grads(0), grads(1)
    assert set(auto) == set(stream) and any("attack_query_transform" in n for n in auto)
    for n in auto:
        scale = stream[n].abs().max().item()
        assert (auto[n] - stream[n]).abs().max().item() <= 1e-3 * scale + 1e-7, (n, (auto[n] - stream[n]).abs().max().item(), scale)


@pytest.mark.parametrize("L,H,nh", [(200, 64, 2), (70, 128, 4)])
def test_replayed_graph_starts_every_accumulation_from_zero(L, H, nh):
    """The streaming backward adds the calibrator's parameter partials into zeroed rows.  Captured into a hipGraph and
    replayed, every replay must start them from zero: with hipMemsetAsync nodes the sums of earlier replays survived
    (parameters went non-finite after ~70 steps of the L = 200, d = 128 configuration); the zero fill is a kernel now."""
    B = 8
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    t = {k: mk(B, L, H).requires_grad_(True) for k in ("q", "k", "v", "qa", "ka")}
    t["gl"] = mk(B, L, L).requires_grad_(True)
    dh = H // nh
    w = {k: (0.3 * torch.randn(*s, generator=g)).to(DEV).requires_grad_(True)
         for k, s in (("w_order", (1, 2 * dh)), ("b_order", (1,)), ("w_dist", (1, 2 * dh)), ("b_dist", (1,)), ("scalar", (1,)))}
    lens = torch.randint(1, L + 1, (B,), generator=g)
    kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    mask = A.StructuredMask(kv, causal=True)
    cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
    cot = [mk(B, L, H), mk(B, L, H), mk(B, nh, L, L)]
    ins = list(t.values()) + list(w.values())

    def walk():  # forward and backward in one go (a backward alone cannot be captured: it runs on the forward's stream)
        out = A.calibrated_attention(t["q"], t["k"], t["v"], t["qa"], t["ka"], t["gl"], mask, cfg, p_drop=0.5, seed=7, **w)
        loss = sum((o * c).sum() for o, c in zip(out[:3], cot))
        return torch.autograd.grad(loss, ins)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        want = [x.clone() for x in walk()]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = walk()
    for _ in range(25):
        graph.replay()
    torch.cuda.synchronize()
    for name, a, b in zip(list(t) + list(w), got, want):
        assert torch.isfinite(a).all(), name
        assert (a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-6, name


def test_long_graph_training_of_the_long_configuration_stays_finite():
    """BASELINE configs[3] (L = 200, d = 128, 4 heads, 100k items, B = 512) replayed from its hipGraph for 130 steps: the
    streaming backward's parameter partials and the cross-entropy's d_out (atomics beyond the slab limit) both start
    from a zero fill inside the graph.  With hipMemsetAsync nodes this run went non-finite after 12-116 replays
    (tools/nan_probe_graph.py; no small reproduction was found: the unit test above passes either way)."""
    cfgd = dict(n_layers=2, n_heads=4, hidden_size=128, inner_size=512, hidden_dropout_prob=0.5, attn_dropout_prob=0.5,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, rich_calibrated_combine='none',
                use_position_embedding=False, trainable_mask_loss_weight=False, mask_loss_weight=0.03,
                MAX_ITEM_LIST_LENGTH=200, gate_seq_length=200)
    B, L, N = 512, 200, 100000
    torch.manual_seed(42)
    model = A.ACSASRec(A.DictConfig(cfgd), A.ItemCount(N)).to(DEV)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model)
    model.train()
    g = torch.Generator().manual_seed(1000)
    pool = []
    for _ in range(8):
        lens = torch.randint(1, L + 1, (B,), generator=g)
        ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
        pool.append({"item_id_list": ids.to(DEV), "item_length": lens.to(DEV),
                     "item_id": torch.randint(1, N, (B,), generator=g).to(DEV)})
    trainer.enable_graph(pool[0])
    for i in range(130):
        att, cal = trainer.train_step(pool[i % 8])
        # eager work and a synchronisation between the replays, as a training loop that logs its losses has them (the
        # failure needed this: back-to-back replays alone stayed finite)
        bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        assert not bad and float(att) == float(att) and float(cal) == float(cal), (i, bad[:6])
    assert float(cal) < 11.6  # the calibrated loss has moved down from log(100000) = 11.51 + noise, not blown up
