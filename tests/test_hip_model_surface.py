"""GPU suite: the parts of the model / trainer surface that the golden-vector tests do not reach --
ACSASRec.predict, the BPR branch of _cal_loss (acsasrec.py:107-116, 146-155), trainable_mask_loss_weight
(:59-60), AttackSASRecTrainer.evaluate_scores (trainer.py:926-945), _train_epoch with and without a captured
graph, and the error behaviour of the fused cross-entropy on labels outside the catalogue."""
import pytest
import torch

import ac_tsr_amd as A

pytestmark = pytest.mark.gpu
DEV = "cuda"

CFG = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.5, attn_dropout_prob=0.5,
           hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
           two_level=True, use_order=True, use_distance=True, mask_loss_weight=0.03)


def _batch(B=48, L=50, N=700, seed=1, neg=False):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    ids = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    b = {"item_id_list": ids.to(DEV), "item_length": lens.to(DEV), "item_id": torch.randint(1, N, (B,), generator=g).to(DEV)}
    if neg:
        b["neg_item_id"] = torch.randint(1, N, (B,), generator=g).to(DEV)
    return b


def _model(N=700, **over):
    torch.manual_seed(0)
    return A.ACSASRec(A.DictConfig(dict(CFG, **over)), A.ItemCount(N)).to(DEV)


def test_predict_equals_full_sort_scores_at_the_test_item():
    model = _model().eval()
    batch = _batch()
    with torch.no_grad():
        attacked, scores = model.predict(batch)
        _, full = model.full_sort_predict(batch)
    assert attacked.shape == scores.shape == (48,)
    picked = full.gather(1, batch["item_id"].view(-1, 1)).squeeze(1)
    assert (scores - picked).abs().max() <= 1e-5
    assert torch.isfinite(attacked).all()
    # the attacked branch is the same function of the attacked output (noise drawn from the seeded CPU generator)
    with torch.no_grad():
        torch.manual_seed(11)
        a1, _ = model.predict(batch)
        torch.manual_seed(11)
        att_out, _, _ = model.forward(batch["item_id_list"], batch["item_length"])
    assert (a1 - (att_out * model.item_embedding(batch["item_id"])).sum(1)).abs().max() <= 1e-6


def test_bpr_loss_branch_matches_its_definition_and_trains():
    model = _model(loss_type='BPR').eval()
    batch = _batch(neg=True)
    torch.manual_seed(5)
    att, cal = model.calculate_loss(batch)
    with torch.no_grad():
        _, cal_out, _ = model.forward(batch["item_id_list"], batch["item_length"])
        pos = (cal_out * model.item_embedding(batch["item_id"])).sum(-1)
        neg = (cal_out * model.item_embedding(batch["neg_item_id"])).sum(-1)
        expect = -torch.log(1e-10 + torch.sigmoid(pos - neg)).mean()  # recbole/model/loss.py:21-47
    assert abs(cal.item() - expect.item()) <= 1e-5
    assert torch.isfinite(att)
    model.train()
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
    first = None
    for _ in range(25):
        att, cal = trainer.train_step(batch)
        first = cal.item() if first is None else first
    assert cal.item() < first


def test_trainable_mask_loss_weight_is_a_parameter_that_never_receives_a_gradient():
    """acsasrec.py:59-60 makes the weight a Parameter initialised to 0.3; it only enters the attacked loss, whose
    backward runs with everything but the attack transforms frozen (trainer.py:678-684), so it never moves."""
    model = _model(trainable_mask_loss_weight=True).train()
    assert "mask_loss_weight" in model.state_dict() and model.mask_loss_weight.shape == (1,)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-2), model)
    batch = _batch()
    for _ in range(3):
        att, cal = trainer.train_step(batch)
    assert torch.isfinite(att) and torch.isfinite(cal)
    assert model.mask_loss_weight.grad is None or float(model.mask_loss_weight.grad.abs().sum()) == 0.0
    assert abs(float(model.mask_loss_weight) - 0.3) <= 1e-7


def test_evaluate_scores_masks_the_padding_item():
    model = _model()
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
    batch = _batch()
    scores = trainer.evaluate_scores(batch)
    with torch.no_grad():
        _, full = model.full_sort_predict(batch)
    assert scores.shape == (48, 700) and torch.isinf(scores[:, 0]).all() and (scores[:, 0] < 0).all()
    assert (scores[:, 1:] - full[:, 1:]).abs().max() <= 1e-6
    assert not model.training


def test_train_epoch_sums_are_the_same_with_and_without_a_graph():
    """_train_epoch accumulates the per-batch losses (trainer.py:664-669); in graph mode train_step hands back static
    buffers, and a shorter last batch falls back to the eager step."""
    batches = [_batch(seed=s) for s in (1, 2, 3)] + [_batch(B=20, seed=4)]
    sums = []
    for graph in (False, True):
        torch.manual_seed(0)
        model = _model(hidden_dropout_prob=0.0, attn_dropout_prob=0.0).train()
        # (a learning rate of exactly 0 reads as "unset" in the reference's `config[...] or default` idiom)
        trainer = A.AttackSASRecTrainer(A.DictConfig(learner='sgd', learning_rate=1e-30), model)
        if graph:
            trainer.enable_graph(batches[0], warmup=1)
        # frozen weights and no dropout: the only randomness left is the attack noise, which enters the attacked loss only
        sums.append(trainer._train_epoch(batches))
    assert abs(sums[0][1] - sums[1][1]) <= 1e-4 * abs(sums[0][1])  # calibrated losses: deterministic
    assert abs(sums[0][0] - sums[1][0]) <= 0.05 * abs(sums[0][0])  # attacked losses differ by their noise draws only
    per_batch = sums[0][1] / len(batches)
    assert 5.0 < per_batch < 8.0  # ~log(700): four batches were counted, none twice, none dropped


def test_annealing_refuses_graph_capture():
    model = _model(combine_option='annealing').train()
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-3), model)
    with pytest.raises(ValueError, match="annealing"):
        trainer.enable_graph(_batch())
    att, cal = trainer.train_step(_batch())  # eager training keeps working, and the rate keeps decaying
    assert torch.isfinite(cal) and model.trm_encoder.layer[0].anneal_step >= 1


def test_fused_cross_entropy_flags_labels_outside_the_catalogue():
    from ac_tsr_amd import ce
    g = torch.Generator().manual_seed(0)
    out = torch.randn(32, 64, generator=g).to(DEV)
    table = (0.1 * torch.randn(500, 64, generator=g)).to(DEV)
    tgt = torch.randint(0, 500, (32,), generator=g).to(DEV)
    good = ce.full_sort_cross_entropy_rows(out, table, tgt)
    assert torch.isfinite(good).all()
    bad = tgt.clone()
    bad[3], bad[7] = -100, 500
    for table_grad in (True, False):
        rows = ce.full_sort_cross_entropy_rows(out, table, bad, table_grad=table_grad)
        assert torch.isnan(rows[3]) and torch.isnan(rows[7])
        ok = torch.ones(32, dtype=torch.bool, device=DEV)
        ok[3] = ok[7] = False
        assert (rows[ok] - good[ok]).abs().max() <= 1e-5


def test_step_inputs_launch_copies_counts_and_forms_the_read_positions():
    """[r4] acattn_step_inputs: the start of a replayed step in one launch (batch copies of any alignment, replay counter + 1,
    item_length - 1 read from the SOURCE of a tensor that is being copied)."""
    import ctypes as C
    from ac_tsr_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    src = [torch.randint(0, 1000, (512, 50), generator=g).to(DEV), torch.randint(1, 51, (512,), generator=g).to(DEV),
           torch.randint(0, 255, (37,), generator=g).to(torch.uint8).to(DEV)[1:]]  # (the last one: odd size, odd address)
    dst = [torch.zeros_like(t) for t in src]
    counter = torch.tensor([41], dtype=torch.int64, device=DEV)
    last = torch.empty(512, dtype=torch.int64, device=DEV)
    n = len(src)
    s = (C.c_void_p * n)(*(t.data_ptr() for t in src))
    d = (C.c_void_p * n)(*(t.data_ptr() for t in dst))
    nb = (C.c_int64 * n)(*(t.numel() * t.element_size() for t in src))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.acattn_step_inputs(s, d, nb, n, C.c_void_p(counter.data_ptr()), C.c_void_p(src[1].data_ptr()),
                                      C.c_void_p(last.data_ptr()), 512, stream), "step_inputs")
    torch.cuda.synchronize()
    for a, b in zip(src, dst):
        assert torch.equal(a, b)
    assert counter.item() == 42 and torch.equal(last, src[1] - 1)
    assert lib.acattn_step_inputs(s, d, nb, 7, None, None, None, 0, stream) < 0 and b"ACATTN_MAX_COPIES" in lib.acattn_last_error()
    assert lib.acattn_step_inputs(s, d, nb, n, None, C.c_void_p(src[1].data_ptr()), None, 512, stream) < 0
