"""CPU suite: the oracle (oracle/ac_tsr_ref.py) against every committed golden vector.

The vectors were produced by running the genuine reference (oracle/gen_golden.py); these tests pin
the restatement to them on any machine, without the reference being present.
"""
import pytest
import torch

from oracle import ac_tsr_ref as O
from tests._golden import ENCODER_CASES, MODEL_CASES, Case

torch.set_num_threads(4)


@pytest.mark.parametrize("name", ENCODER_CASES)
@pytest.mark.parametrize("materialize", [True, False])
def test_encoder_matches_reference(name, materialize):
    c = Case(name)
    cfg = c.encoder_cfg()
    rnds = c.layer_randomness(cfg.n_layers)
    with torch.no_grad():
        outs, masks, dbgs = O.encoder_forward(c.t("in.x"), c.t("in.mask"), c.params(), cfg, rnds,
                                              materialize=materialize)
    # literal restatement is bit-near; the rank-1 form of the spatial calibrator reassociates sums
    tol_p = 2e-6 if materialize else 2e-5
    tol_o = 2e-5 if materialize else 5e-4
    for i in range(cfg.n_layers):
        assert (outs[i][0] - c.t(f"out.{i}.attacked")).abs().max() <= tol_o
        assert (outs[i][1] - c.t(f"out.{i}.calibrated")).abs().max() <= tol_o
        assert (masks[i] - c.t(f"out.{i}.M")).abs().max() <= tol_p
        assert (dbgs[i]["calibrated_attention"] - c.t(f"out.{i}.calibrated_attention")).abs().max() <= tol_p
        for k in ("before_spatial", "after_spatial", "perturbed_attention"):
            if c.has(f"out.{i}.{k}"):
                assert (dbgs[i][k] - c.t(f"out.{i}.{k}")).abs().max() <= tol_p
        assert (dbgs[i]["ctx_calibrated"] - c.t(f"out.{i}.ctx_calibrated")).abs().max() <= tol_o
        assert (dbgs[i]["ctx_attacked"] - c.t(f"out.{i}.ctx_attacked")).abs().max() <= tol_o


@pytest.mark.parametrize("name", ENCODER_CASES)
def test_attention_mask(name):
    c = Case(name)
    m = O.attention_mask(c.t("in.item_seq"), bool(int(c.raw["meta.bidirectional"])))
    assert torch.equal(m, c.t("in.mask"))


@pytest.mark.parametrize("name", ENCODER_CASES[:4])
def test_core_from_projected_equals_layer(name):
    """The projected-tensor boundary used by the HIP core reproduces the layer's probabilities."""
    c = Case(name)
    cfg = c.encoder_cfg()
    P = O.layer_params(c.params(), "layer.0.")
    x, mask = c.t("in.x"), c.t("in.mask")
    with torch.no_grad():
        mq = O._lin(x, P, "attack_attention.query")
        mk = O._lin(x, P, "attack_attention.key")
        mv = O._lin(x, P, "attack_attention.value")
        qa = O._lin(mq, P, "attack_attention.attack_query_transform")
        ka = O._lin(mk, P, "attack_attention.attack_key_transform")
        gl = O._lin(mq, P, "gate") if cfg.combine_option == "gate" else None
        zero = torch.zeros(1, 2 * cfg.hidden_size // cfg.n_heads)
        r = O.core_from_projected(
            mq, mk, mv, qa, ka, gl, mask,
            P.get("attack_attention.order_affine.weight", zero), P.get("attack_attention.order_affine.bias"),
            P.get("attack_attention.distance_affine.weight", zero), P.get("attack_attention.distance_affine.bias"),
            P.get("attack_attention.scalar"), cfg, c.t("in.noise.0"))
    assert (r["M"] - c.t("out.0.M")).abs().max() <= 1e-6
    assert (r["combined"] - c.t("out.0.calibrated_attention")).abs().max() <= 2e-6
    assert (r["ctx_calibrated"] - c.t("out.0.ctx_calibrated")).abs().max() <= 2e-5
    assert (r["ctx_attacked"] - c.t("out.0.ctx_attacked")).abs().max() <= 2e-5


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_losses_and_two_pass_grads(name):
    c = Case(name)
    cfg = c.model_cfg()
    train = bool(int(c.raw["meta.train"]))
    rnds = c.layer_randomness(cfg.enc.n_layers, train)
    keep_emb = c.t("in.keep_emb").float() if train else None
    att, cal, grads = O.two_pass_grads(c.batch(), c.params(), cfg, train, rnds, keep_emb)
    assert abs(att.item() - float(c.raw["out.att_loss"])) <= 2e-5
    assert abs(cal.item() - float(c.raw["out.cal_loss"])) <= 2e-5
    for n, g in c.grads().items():
        scale = max(g.abs().max().item(), 1e-6)
        assert ((grads[n] - g).abs().max() / scale) <= 2e-3, n
        if O.is_attack_param(n):
            assert g.abs().max() > 0  # attack transforms do receive the attacked-loss gradient


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if "eval" in n])
def test_model_logits(name):
    c = Case(name)
    cfg = c.model_cfg()
    with torch.no_grad():
        logits = O.full_sort_predict(c.batch(), c.params(), cfg, c.layer_randomness(cfg.enc.n_layers))
    assert (logits - c.t("out.logits")).abs().max() <= 2e-5


def test_degenerate_identities():
    """identities that need no oracle."""
    c = Case("enc_plain")  # use_order = use_distance = False
    cfg = c.encoder_cfg()
    with torch.no_grad():
        _, _, dbgs = O.encoder_forward(c.t("in.x"), c.t("in.mask"), c.params(), cfg, c.layer_randomness(cfg.n_layers))
    assert torch.equal(dbgs[0]["after_spatial"], dbgs[0]["before_spatial"])
    for k in ("after_spatial", "perturbed_mask", "perturbed_attention", "calibrated_attention"):
        assert (dbgs[0][k].sum(-1) - 1).abs().max() <= 1e-5
