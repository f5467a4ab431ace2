"""CPU suite: AcBERT4Rec -- the oracle's restatement against the reference's golden vectors (tests/golden/bert_*.npz)
and the host logic of the product model (cloze reconstruction, state-dict surface).  No GPU compute."""
import random

import pytest
import torch

import ac_tsr_amd as A
from oracle import ac_tsr_ref as O
from tests._golden import BERT_CASES, Case


def _product_model(c: Case, **extra):
    cfg = c.bert_cfg()
    conf = dict(n_layers=cfg.enc.n_layers, n_heads=cfg.enc.n_heads, hidden_size=cfg.enc.hidden_size,
                inner_size=cfg.enc.inner_size, hidden_dropout_prob=0.5, attn_dropout_prob=0.5, hidden_act='gelu',
                layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option=str(c.raw["meta.combine"]),
                two_level=True, use_order=True, use_distance=True, rich_calibrated_combine='none',
                use_position_embedding=cfg.use_position_embedding, mask_loss_weight=cfg.mask_loss_weight,
                mask_ratio=float(c.raw["meta.mask_ratio"]), MAX_ITEM_LIST_LENGTH=cfg.max_seq_length, device='cpu')
    conf.update(extra)
    return cfg, A.AcBERT4Rec(A.DictConfig(conf), A.ItemCount(cfg.n_items))


@pytest.mark.parametrize("name", BERT_CASES)
def test_oracle_losses_and_two_pass_grads(name):
    c = Case(name)
    cfg = c.bert_cfg()
    rnds = c.layer_randomness(cfg.enc.n_layers)
    att, cal, grads = O.bert_two_pass_grads(c.t("in.masked_seq"), c.t("in.pos_items"), c.t("in.masked_index"), c.params(),
                                            cfg, False, rnds)
    assert abs(att.item() - float(c.raw["out.att_loss"])) <= 2e-5
    assert abs(cal.item() - float(c.raw["out.cal_loss"])) <= 2e-5
    for n, g in c.grads().items():
        scale = max(g.abs().max().item(), 1e-6)
        assert ((grads[n] - g).abs().max() / scale) <= 2e-3, n
    # the mask-token row is used as an input embedding, never as a candidate: it does get a gradient
    assert c.grads()["item_embedding.weight"][cfg.n_items].abs().max() > 0


def test_oracle_full_sort_scores():
    c = Case("bert_fixed_scores")
    cfg = c.bert_cfg()
    rn = [O.LayerRandomness(noise=c.t(f"in.noise_eval.{i}")) for i in range(cfg.enc.n_layers)]
    with torch.no_grad():
        att_s, s = O.bert_full_sort_predict(c.t("in.item_id_list"), c.t("in.item_length"), c.params(), cfg, rn)
    assert s.shape == (c.t("in.item_id_list").shape[0], cfg.n_items)
    assert (s - c.t("out.scores")).abs().max() <= 2e-5
    assert (att_s - c.t("out.att_scores")).abs().max() <= 2e-5


@pytest.mark.parametrize("name", BERT_CASES)
def test_cloze_reconstruction_reproduces_the_reference_batch(name):
    """Seeded `random` -> the masked batch the genuine reference built (oracle restatement and product model)."""
    c = Case(name)
    cfg, model = _product_model(c)
    want = [c.t(k) for k in ("in.masked_seq", "in.pos_items", "in.neg_items", "in.masked_index")]
    seed = int(c.raw["in.random_seed"])
    ratio = float(c.raw["meta.mask_ratio"])
    random.seed(seed)
    got_o = O.cloze_mask_host(c.t("in.item_id_list"), ratio, cfg.n_items, cfg.n_items, int(ratio * cfg.max_seq_length))
    random.seed(seed)
    got_m = model.reconstruct_train_data(c.t("in.item_id_list"))
    for w, a, b in zip(want, got_o, got_m):
        assert torch.equal(w, a) and torch.equal(w, b)


def test_state_dict_surface_and_reference_checkpoint_loads():
    c = Case("bert_gate")
    cfg, model = _product_model(c)
    assert set(model.state_dict()) == set(c.params())
    model.load_state_dict(c.params())
    assert model.item_embedding.weight.shape[0] == cfg.n_items + 1 and model.mask_token == cfg.n_items
    assert model.mask_item_length == int(0.2 * 50)
    with pytest.raises(AssertionError):
        _product_model(c, loss_type='XE')


@pytest.mark.parametrize("ratio", [0.2, 0.9])
def test_device_cloze_has_the_reference_semantics(ratio):
    """The tensor-op flavour (here on CPU tensors): masks only real items before the first padding, keeps order,
    right-aligns the short lists and keeps the LAST mask_item_length entries, negatives avoid the sequence."""
    c = Case("bert_gate")
    cfg, model = _product_model(c, cloze_on_device=True, mask_ratio=ratio)
    g = torch.Generator().manual_seed(3)
    B, L, N = 64, 50, cfg.n_items
    lens = torch.randint(1, L + 1, (B,), generator=g)
    seq = torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    seq[3, 4] = 0  # a hole: everything after it is "ended" for the reference's loop
    masked, pos, neg, idx = model._reconstruct_train_data_device(seq, generator=g)
    ml = model.mask_item_length
    assert masked.shape == (B, L) and pos.shape == neg.shape == idx.shape == (B, ml)
    real = (seq != 0).long().cumprod(1).bool()
    is_masked = masked == model.mask_token
    assert not (is_masked & ~real).any() and torch.equal(masked[~is_masked], seq[~is_masked])
    frac = is_masked[real].float().mean().item()
    assert abs(frac - ratio) < 0.08
    for b in range(B):
        where = is_masked[b].nonzero().flatten().tolist()[-ml:] if ml else []
        want_idx = [0] * (ml - len(where)) + where
        assert idx[b].tolist() == want_idx
        assert pos[b].tolist() == [0] * (ml - len(where)) + [int(seq[b, j]) for j in where]
        for slot, j in enumerate(want_idx):
            if slot >= ml - len(where):
                assert 1 <= int(neg[b, slot]) < N and int(neg[b, slot]) not in seq[b].tolist()
            else:
                assert int(neg[b, slot]) == 0


def test_reconstruct_test_data_appends_the_mask_token():
    c = Case("bert_gate")
    cfg, model = _product_model(c)
    seq, ln = c.t("in.item_id_list"), c.t("in.item_length")
    out = model.reconstruct_test_data(seq, ln)
    assert torch.equal(out, O.bert_append_mask_token(seq, ln, cfg.n_items))
    assert out.shape[1] == seq.shape[1] + 1 and (out[torch.arange(len(ln)), ln] == cfg.n_items).all()
