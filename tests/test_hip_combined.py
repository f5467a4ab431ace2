"""GPU suite (-m gpu): the single-pass combined backward (ac_tsr_amd/combined.py) -- ONE walk of the autograd
graph carrying the cotangents of both losses -- against the same golden gradients of the genuine reference that pin the
two-walk protocol (recbole/trainer/trainer.py:672-686), and against the two-walk trainer step by step.

Tolerances as for the two-walk tests: every gradient within 2e-3 of its tensor's largest magnitude of the reference's."""
import pytest
import torch

import ac_tsr_amd as A
from tests._golden import Case
from tests.test_hip_backward import _build_model, _rnds_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("prune", [True, False], ids=["pruned_schedule", "reference_schedule"])
@pytest.mark.parametrize("name", ["model_eval", "model_eval_stress", "model_train", "model_beauty"])
def test_one_walk_gives_the_reference_gradients(name, prune):
    c = Case(name)
    cfg, model = _build_model(c)
    model.step_state.prune_dead_work = prune
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model, combined_backward=True)
    train = bool(int(c.raw["meta.train"]))
    model.train(train)
    batch = {k: v.to(DEV) for k, v in c.batch().items()}
    rnds = _rnds_for(c, cfg.enc.n_layers, train)
    keep_emb = c.t("in.keep_emb").to(DEV) if train else None
    model.zero_grad()
    att, cal = model.calculate_loss(batch, _rnds=rnds, _keep_emb=keep_emb)
    assert abs(att.item() - float(c.raw["out.att_loss"])) <= 1e-4
    assert abs(cal.item() - float(c.raw["out.cal_loss"])) <= 1e-4
    trainer.combined_backward_walk(att, cal)
    stats = trainer.last_walk_stats
    assert stats["dual_nodes"] + stats["pair_nodes"] >= 2 * cfg.enc.n_layers and stats["prefix_nodes"] >= 1, stats
    ref = c.grads()
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        err = (g.cpu() - ref[n]).abs().max().item()
        assert err <= 2e-3 * ref[n].abs().max().item() + 2e-8, (n, err, ref[n].abs().max().item())


@pytest.mark.parametrize("name", ["bert_gate", "bert_fixed_scores"])
def test_one_walk_gives_the_reference_gradients_of_acbert4rec(name):
    from tests.test_hip_bert4rec import _model, _noise
    c = Case(name)
    cfg, model = _model(c)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model, combined_backward=True)
    model.eval()
    cloze = tuple(c.t(k).to(DEV) for k in ("in.masked_seq", "in.pos_items", "in.neg_items", "in.masked_index"))
    model.zero_grad()
    att, cal = model.calculate_loss({"item_id_list": c.t("in.item_id_list").to(DEV)}, _cloze=cloze,
                                    _rnds=_noise(c, cfg.enc.n_layers))
    trainer.combined_backward_walk(att, cal)
    ref = c.grads()
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        err = (g.cpu() - ref[n]).abs().max().item()
        assert err <= 2e-3 * ref[n].abs().max().item() + 2e-8, (n, err, ref[n].abs().max().item())


def _cfgd(**over):
    d = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.1, attn_dropout_prob=0.2,
             hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
             two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03)
    d.update(over)
    return d


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("L,H,nh", [(50, 64, 2), (200, 128, 4)], ids=["headline_shape", "cfg4_shape"])
def test_combined_trainer_takes_the_steps_of_the_two_walk_trainer(graph, L, H, nh):
    """Three optimizer steps from the same initial state and the same random draws: the parameters after the one-walk steps
    equal those after the reference protocol's (the kernels' sums run in a different order: 1e-5 relative)."""
    g = torch.Generator().manual_seed(1)
    B, N = 48, 3000
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}
    states, losses = [], []
    for combined in (False, True):
        torch.manual_seed(3)
        model = A.ACSASRec(A.DictConfig(_cfgd(hidden_size=H, n_heads=nh, inner_size=4 * H, MAX_ITEM_LIST_LENGTH=L,
                                              gate_seq_length=L)), A.ItemCount(N)).to(DEV)
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model, combined_backward=combined)
        model.train()
        torch.manual_seed(5)
        if graph:
            trainer.enable_graph(batch, warmup=1)
        for _ in range(3):
            out = trainer.train_step(batch)
        torch.cuda.synchronize()
        losses.append([float(x.detach()) for x in out])
        states.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        if combined:
            # L > 64: the first layer's attention node evaluates BOTH cotangent sets in one launch pair (acattn_bwd_io.dqa2:
            # the streaming kernels share the rebuilt tiles); L <= 64: the row-resident kernel, one launch per set
            assert trainer.last_walk_stats["pair_nodes"] == (1 if L > 64 else 0), trainer.last_walk_stats
    assert losses[0] == pytest.approx(losses[1], rel=1e-5, abs=1e-6)
    for k in states[0]:
        # (attack_key_transform.bias has a mathematically zero gradient -- soft-max is invariant to a per-query shift -- so
        # Adam normalises pure cancellation noise there: absolute floor)
        scale = max(1e-3, states[0][k].abs().max().item())
        assert (states[0][k] - states[1][k]).abs().max().item() <= 2e-5 * scale + 2e-7, k


def test_combined_walk_with_the_data_parallel_gradient_path():
    """combined_backward with a GradSynchronizer on one GPU (flat buffer, optimizer fed from it): same parameters as the
    plain two-walk trainer."""
    from ac_tsr_amd import parallel
    g = torch.Generator().manual_seed(2)
    B, L, N = 32, 50, 2000
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}
    states = []
    for combined in (False, True):
        torch.manual_seed(3)
        model = A.ACSASRec(A.DictConfig(_cfgd()), A.ItemCount(N)).to(DEV)
        sync = parallel.GradSynchronizer.for_two_pass_model(model) if combined else None
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model, grad_sync=sync,
                                        combined_backward=combined)
        model.train()
        torch.manual_seed(5)
        for _ in range(2):
            trainer.train_step(batch)
        torch.cuda.synchronize()
        states.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for k in states[0]:
        scale = max(1e-3, states[0][k].abs().max().item())
        assert (states[0][k] - states[1][k]).abs().max().item() <= 2e-5 * scale, k


@pytest.mark.parametrize("over", [dict(hidden_size=96, inner_size=192, n_heads=6), dict(use_position_embedding=True),
                                  dict(trainable_mask_loss_weight=True, mask_loss_weight=None), dict(combine_option="fixed"),
                                  dict(n_layers=3, n_heads=4, inner_size=128), dict(two_level=False, rich_calibrated_combine="fixed")],
                         ids=["hidden_96_library_paths", "position_embedding", "trainable_penalty_weight", "fixed_combine",
                              "three_layers", "one_level"])
def test_one_walk_equals_two_walks_on_the_less_common_paths(over):
    """Widths the hand-written chains do not cover (library GEMMs, torch LayerNorm: plain torch nodes between the custom ones),
    position embeddings, a trainable penalty weight (never receives a gradient: recbole/trainer/trainer.py:679-684), the
    'fixed' combine and one-level forms (general kernels), a middle layer with attack transforms upstream: the gradients of
    one eager step in the one-walk mode equal the two-walk protocol's."""
    g = torch.Generator().manual_seed(7)
    B, L, N = 24, 50, 900
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}
    grads = []
    for combined in (False, True):
        torch.manual_seed(11)
        model = A.ACSASRec(A.DictConfig(_cfgd(**over)), A.ItemCount(N)).to(DEV)
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model, combined_backward=combined)
        model.train()
        torch.manual_seed(13)
        model.zero_grad()
        if combined:
            att, cal = model.calculate_loss(batch)
            trainer.combined_backward_walk(att, cal)
        else:
            att, cal = trainer._pass_one(batch)
            trainer._pass_two(att)
        grads.append({n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()})
    for n in grads[0]:
        scale = grads[0][n].abs().max().item()
        assert (grads[0][n] - grads[1][n]).abs().max().item() <= 1e-4 * scale + 1e-8, n


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("L,H,nh", [(50, 64, 2), (200, 128, 4)], ids=["headline_shape", "cfg4_shape"])
def test_deferred_weight_gradient_reductions_change_nothing(graph, L, H, nh):
    """[r4] StepState.defer_reductions (the trainer's default): stage 2 of every weight gradient and the LayerNorm parameter
    sums of a backward walk in ONE launch at its end, written into the tensors autograd adopted as the leaves' .grad.  The
    same kernels sum in the same order; three optimizer steps must leave the same parameters up to the run-to-run noise of
    the step's float atomics (embedding scatter, the cross-entropy's one-hot rows) -- and the deferred path must have been
    the one that ran."""
    from ac_tsr_amd.state import StepState
    g = torch.Generator().manual_seed(2)
    B, N = 40, 2500
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    batch = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": ids[torch.arange(B), lens - 1].to(DEV)}
    states, flushed = [], []
    for defer in (False, True):
        torch.manual_seed(3)
        model = A.ACSASRec(A.DictConfig(_cfgd(hidden_size=H, n_heads=nh, inner_size=4 * H, MAX_ITEM_LIST_LENGTH=L,
                                              gate_seq_length=L)), A.ItemCount(N)).to(DEV)
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
        assert trainer.state.defer_reductions  # the default
        trainer.state.defer_reductions = defer
        n_jobs = []
        orig = StepState.flush_deferred

        def counting(self, params, _orig=orig, _n=n_jobs):
            _n.append(len(self._deferred))
            return _orig(self, params)  # (called by the pass context the trainer opens with the walk's leaves)

        StepState.flush_deferred = counting
        try:
            model.train()
            torch.manual_seed(5)
            if graph:
                trainer.enable_graph(batch, warmup=1)
            for _ in range(3):
                trainer.train_step(batch)
            torch.cuda.synchronize()
        finally:
            StepState.flush_deferred = orig
        flushed.append(sum(n_jobs))
        states.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    assert flushed[0] == 0 and flushed[1] >= 10, flushed  # (projections: 5-6 weights per layer, tails: 3 + a row sum)
    # a pass opened WITHOUT the walk's leaves never defers: nobody would flush
    st = trainer.state
    with st.calibrated_pass():
        assert not st.deferring()
    with st.calibrated_pass([p for p in model.parameters()]):
        assert st.deferring()
    for k in states[0]:
        scale = max(1e-3, states[0][k].abs().max().item())
        assert (states[0][k] - states[1][k]).abs().max().item() <= 2e-5 * scale + 2e-7, k
