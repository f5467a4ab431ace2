"""GPU suite (-m gpu): AcBERT4Rec on the HIP path against the genuine reference's vectors (tests/golden/bert_*.npz)."""
import types

import pytest
import torch

import ac_tsr_amd as A
from tests._golden import BERT_CASES, Case
from tests.test_bert4rec_cpu import _product_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _noise(c, n_layers, key="in.noise"):
    out = []
    for i in range(n_layers):
        ns = types.SimpleNamespace(noise=c.t(f"{key}.{i}").to(DEV))
        for f in ("keep_after", "keep_before", "keep_mask", "keep_out_att", "keep_out_cal", "keep_ffn_att", "keep_ffn_cal"):
            setattr(ns, f, None)
        out.append(ns)
    return out


def _model(c, **extra):
    cfg, m = _product_model(c, device=DEV, **extra)
    m.load_state_dict(c.params(), strict=not extra)
    return cfg, m.to(DEV)


@pytest.mark.parametrize("name", BERT_CASES)
def test_losses_and_two_pass_gradients_match_reference(name):
    c = Case(name)
    cfg, model = _model(c)
    model.eval()
    cloze = tuple(c.t(k).to(DEV) for k in ("in.masked_seq", "in.pos_items", "in.neg_items", "in.masked_index"))
    model.zero_grad()
    att, cal = model.calculate_loss({"item_id_list": c.t("in.item_id_list").to(DEV)}, _cloze=cloze,
                                    _rnds=_noise(c, cfg.enc.n_layers))
    assert abs(att.item() - float(c.raw["out.att_loss"])) <= 1e-4
    assert abs(cal.item() - float(c.raw["out.cal_loss"])) <= 1e-4
    for n, p in model.named_parameters():
        p.requires_grad = not A.is_attack_param(n)
    cal.backward(retain_graph=True)
    for n, p in model.named_parameters():
        p.requires_grad = A.is_attack_param(n)
    att.backward()
    ref = c.grads()
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        err = (g.cpu() - ref[n]).abs().max().item()
        assert err <= 2e-3 * ref[n].abs().max().item() + 2e-8, (n, err, ref[n].abs().max().item())


def test_full_sort_scores_within_1e4():
    c = Case("bert_fixed_scores")
    cfg, model = _model(c)
    model.eval()
    batch = {"item_id_list": c.t("in.item_id_list").to(DEV), "item_length": c.t("in.item_length").to(DEV)}
    with torch.no_grad():
        att_s, s = model.full_sort_predict(batch, _rnds=_noise(c, cfg.enc.n_layers, "in.noise_eval"))
    assert (s.cpu() - c.t("out.scores")).abs().max() <= 1e-4
    assert (att_s.cpu() - c.t("out.att_scores")).abs().max() <= 1e-4


def test_evaluation_keeps_the_reference_failure_modes():
    """L+1 columns at evaluation: the gate is L wide (RuntimeError) and the position table has L rows (IndexError),
    exactly where the reference fails (acbert4rec.py:47,152-160; layers.py:888)."""
    c = Case("bert_gate")  # gate + position embedding
    _, model = _model(c)
    model.eval()
    batch = {"item_id_list": c.t("in.item_id_list").to(DEV), "item_length": c.t("in.item_length").to(DEV)}
    with pytest.raises(IndexError), torch.no_grad():
        model.full_sort_predict(batch)
    _, model = _model(c, use_position_embedding=False)
    model.eval()
    with pytest.raises(RuntimeError), torch.no_grad():
        model.full_sort_predict(batch)


def test_device_cloze_training_in_a_captured_graph():
    """cloze_on_device + trainer.enable_graph: the whole AcBERT4Rec step (masking included) replays as one hipGraph,
    losses stay finite and the calibrated loss goes down."""
    torch.manual_seed(0)
    conf = dict(n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.1, attn_dropout_prob=0.1,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, use_position_embedding=True, mask_loss_weight=0.03,
                mask_ratio=0.2, cloze_on_device=True, device=DEV)
    B, L, N = 128, 50, 2000
    model = A.AcBERT4Rec(A.DictConfig(conf), A.ItemCount(N)).to(DEV)
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(5, L + 1, (B,), generator=g)
    seq = (torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])).to(DEV)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="adam", learning_rate=1e-3), model)
    batch = {"item_id_list": seq}
    trainer.enable_graph(batch)
    cals = []
    for _ in range(30):
        att, cal = trainer.train_step(batch)
        cals.append(cal.item())
        assert torch.isfinite(att).item() and torch.isfinite(cal).item()
    assert sum(cals[-5:]) < sum(cals[:5])


@pytest.mark.parametrize("kind,L,H,nh", [("sasrec", 200, 128, 4), ("bert", 200, 256, 4)])
def test_baseline_configs_4_and_5_train_end_to_end(kind, L, H, nh):
    """BASELINE configs[3] (L=200, d=128, 4 heads, causal) and configs[4] (AcBERT4Rec, bidirectional, L=200, d=256):
    a few trainer steps through the general kernels (the gate needs gate_seq_length = L) stay finite and
    reduce the calibrated loss."""
    torch.manual_seed(0)
    conf = dict(n_layers=2, n_heads=nh, hidden_size=H, inner_size=2 * H, hidden_dropout_prob=0.1, attn_dropout_prob=0.1,
                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE', combine_option='gate',
                two_level=True, use_order=True, use_distance=True, use_position_embedding=(kind == "bert"),
                mask_loss_weight=0.03, mask_ratio=0.2, cloze_on_device=True, gate_seq_length=L, MAX_ITEM_LIST_LENGTH=L,
                device=DEV)
    B, N = 16, 3000
    cls = A.AcBERT4Rec if kind == "bert" else A.ACSASRec
    model = cls(A.DictConfig(conf), A.ItemCount(N)).to(DEV)
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(20, L + 1, (B,), generator=g)
    seq = (torch.randint(1, N, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])).to(DEV)
    batch = {"item_id_list": seq, "item_length": lens.to(DEV), "item_id": torch.randint(1, N, (B,), generator=g).to(DEV)}
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="adam", learning_rate=2e-3), model)
    cals = []
    for _ in range(12):
        att, cal = trainer.train_step(batch)
        assert torch.isfinite(att).item() and torch.isfinite(cal).item()
        cals.append(cal.item())
    assert sum(cals[-3:]) < sum(cals[:3])
