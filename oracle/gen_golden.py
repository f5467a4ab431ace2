#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE GENUINE REFERENCE.

TEST INFRASTRUCTURE ONLY (see oracle/ac_tsr_ref.py header).  Runs in the build container only:
it imports /root/reference (read-only; nothing is copied from it) with three logging-only modules
(`colorlog`, `colorama`, `torch.utils.tensorboard`) registered as empty stand-ins because
recbole/utils/__init__.py pulls them in for its colour logger.  Only tensors
(inputs, parameters by state-dict key, expected outputs) are written; the reference never travels.

While generating, every case is also evaluated with the CPU restatement (oracle/ac_tsr_ref.py)
and the script aborts if the two disagree, so a committed fixture is by construction a vector on
which restatement == reference.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = os.environ.get("ACTSR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

for _name in ("colorlog", "colorama"):
    _m = types.ModuleType(_name)
    _m.init = lambda *a, **k: None
    sys.modules.setdefault(_name, _m)
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules.setdefault("torch.utils.tensorboard", _tb)

from recbole.model.layers import AttackRTransformerEncoder  # noqa: E402  (the reference)
from recbole.model.sequential_recommender.acsasrec import ACSASRec  # noqa: E402
from recbole.model.sequential_recommender.acbert4rec import AcBERT4Rec  # noqa: E402

from oracle import ac_tsr_ref as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(4)


class Cfg(dict):
    """recbole Config look-alike: missing keys read as None (configurator.py:405-409)."""

    def __getitem__(self, k):
        return self.get(k, None)


class FakeDataset:
    def __init__(self, n_items):
        self.n = n_items

    def num(self, field):
        return self.n


def make_item_seq(B, L, n_items, lens, gen, left_pad_rows=()):
    seq = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        ids = torch.randint(1, n_items, (n,), generator=gen)
        if b in left_pad_rows:  # not produced by RecBole's loader; exercises fully-masked causal rows
            seq[b, L - n:] = ids
        else:
            seq[b, :n] = ids
    return seq


def reinit(module, sigma, gen):
    """Re-draw every parameter at scale sigma (reference init is N(0, 0.02^2): acsasrec.py:74-84)."""
    with torch.no_grad():
        for name, prm in module.named_parameters():
            if name.endswith("LayerNorm.weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=gen))
            elif name.endswith("scalar") or name.endswith("mask_loss_weight"):
                pass
            elif name.endswith("rich_calibrated_combine_ratio"):  # layers.py:874-875 starts it at 0.5: move it off
                prm.fill_(0.3)
            else:
                prm.copy_(sigma * torch.randn(prm.shape, generator=gen))


def save(name, arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


def check(tag, a, b, tol):
    d = (a - b).abs().max().item()
    if not d <= tol:
        raise SystemExit(f"restatement != reference at {tag}: max abs diff {d:g} > {tol:g}")
    return d


def encoder_case(name, *, B, L, H, h, inner, n_layers, combine, use_order=True, use_distance=True, two_level=True,
                 rich="fixed", bidirectional=False, sigma=0.02, seed=0, lens=None, left_pad_rows=(), keep="all",
                 n_items=300):
    print(f"[encoder] {name}")
    gen = torch.Generator().manual_seed(seed)
    ecfg = O.EncoderCfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner, hidden_dropout_prob=0.5,
                        attn_dropout_prob=0.5, hidden_act="gelu", layer_norm_eps=1e-12, combine_option=combine,
                        use_order=use_order, use_distance=use_distance, two_level=two_level,
                        rich_calibrated_combine=rich, seq_length=L)
    torch.manual_seed(seed)
    enc = AttackRTransformerEncoder(n_layers, h, H, inner, 0.5, 0.5, "gelu", 1e-12, combine, use_order, use_distance,
                                    two_level, rich, L)
    reinit(enc, sigma, gen)
    enc.eval()
    if lens is None:
        lens = [int(v) for v in torch.randint(1, L + 1, (B,), generator=gen)]
    item_seq = make_item_seq(B, L, n_items, lens, gen, left_pad_rows)
    x = torch.randn(B, L, H, generator=gen)
    x = torch.nn.functional.layer_norm(x, (H,))  # unit-variance rows like LN(embedding)
    mask = O.attention_mask(item_seq, bidirectional)
    # the reference's own mask builder must agree with the restated one: check through a tiny subclass-free call
    from recbole.model.abstract_recommender import SequentialRecommender
    ref_mask = SequentialRecommender.get_attention_mask(None, item_seq, bidirectional)
    check("attention_mask", mask, ref_mask, 0.0)

    torch.manual_seed(seed + 1000)
    with torch.no_grad():
        layers_out, masks, probs = enc(x, mask, output_all_encoded_layers=True, return_all_attention_prob=True)
    # same draws, regenerated in the reference's order
    torch.manual_seed(seed + 1000)
    rnds = [O.draw_layer_randomness((B, h, L, L), (B, L, H), ecfg, train=False) for _ in range(n_layers)]
    P = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        o_out, o_masks, o_dbg = O.encoder_forward(x, mask, P, ecfg, rnds)
        r_out, _, _ = O.encoder_forward(x, mask, P, ecfg, rnds, materialize=False)
    worst = 0.0
    for i in range(n_layers):
        worst = max(worst, check(f"{name}.L{i}.attacked", o_out[i][0], layers_out[i][0], 2e-5))
        worst = max(worst, check(f"{name}.L{i}.calibrated", o_out[i][1], layers_out[i][1], 2e-5))
        worst = max(worst, check(f"{name}.L{i}.M", o_masks[i], masks[i], 1e-6))
        for k in ("before_spatial", "after_spatial", "perturbed_mask", "perturbed_attention", "calibrated_attention"):
            worst = max(worst, check(f"{name}.L{i}.{k}", o_dbg[i][k], probs[i][k], 2e-6))
        check(f"{name}.L{i}.rank1", r_out[i][1], layers_out[i][1], 5e-4 if sigma > 0.1 else 5e-5)
    print(f"  restatement vs reference: max abs diff {worst:.3g}")

    arr = {"in.x": x, "in.item_seq": item_seq, "in.mask": mask, "meta.bidirectional": int(bidirectional),
           "meta.cfg": np.array([n_layers, h, H, inner, L, int(use_order), int(use_distance), int(two_level)]),
           "meta.combine": combine, "meta.rich": rich, "meta.sigma": sigma}
    for k, v in P.items():
        arr["p." + k] = v
    for i in range(n_layers):
        arr[f"in.noise.{i}"] = rnds[i].noise
        arr[f"out.{i}.attacked"] = layers_out[i][0]
        arr[f"out.{i}.calibrated"] = layers_out[i][1]
        arr[f"out.{i}.M"] = masks[i]
        arr[f"out.{i}.calibrated_attention"] = probs[i]["calibrated_attention"]
        if keep == "all":
            for k in ("before_spatial", "after_spatial", "perturbed_attention"):
                arr[f"out.{i}.{k}"] = probs[i][k]
        # projected-tensor boundary of the HIP core (derived from the reference's returned tensors only)
        arr[f"out.{i}.ctx_attacked"] = O.context_only(probs[i]["perturbed_attention"], o_dbg[i]["value"])
        final = o_dbg[i]["final_combined"]
        arr[f"out.{i}.ctx_calibrated"] = O.context_only(final, o_dbg[i]["value"])
    save(name, arr)


def model_case(name, *, B, L, H, h, inner, n_layers, n_items, combine="gate", sigma=0.02, seed=0, train=False,
               mask_loss_weight=0.03):
    print(f"[model] {name} train={train}")
    gen = torch.Generator().manual_seed(seed)
    cfg = Cfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner, hidden_dropout_prob=0.5,
              attn_dropout_prob=0.5, hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, loss_type="CE",
              combine_option=combine, rich_calibrated_combine="none", two_level=True, use_position_embedding=False,
              use_order=True, use_distance=True, trainable_mask_loss_weight=False, mask_loss_weight=mask_loss_weight,
              USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", LIST_SUFFIX="_list",
              ITEM_LIST_LENGTH_FIELD="item_length", NEG_PREFIX="neg_", MAX_ITEM_LIST_LENGTH=L, device="cpu")
    torch.manual_seed(seed)
    model = ACSASRec(cfg, FakeDataset(n_items))
    if sigma != 0.02:
        reinit(model, sigma, gen)
    lens = [int(v) for v in torch.randint(1, L + 1, (B,), generator=gen)]
    lens[0], lens[1] = 1, L
    item_seq = make_item_seq(B, L, n_items, lens, gen)
    batch = {"item_id_list": item_seq, "item_length": torch.tensor(lens, dtype=torch.long),
             "item_id": torch.randint(1, n_items, (B,), generator=gen)}
    ecfg = O.EncoderCfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner, combine_option=combine,
                        rich_calibrated_combine="none", seq_length=50)
    mcfg = O.ModelCfg(enc=ecfg, n_items=n_items, max_seq_length=L, mask_loss_weight=mask_loss_weight)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}

    model.train(train)
    # -- logits (eval) ---------------------------------------------------------------------
    arr = {"in.item_id_list": item_seq, "in.item_length": batch["item_length"], "in.item_id": batch["item_id"],
           "meta.cfg": np.array([n_layers, h, H, inner, L, n_items]), "meta.combine": combine,
           "meta.train": int(train), "meta.mask_loss_weight": mask_loss_weight}
    for k, v in P.items():
        arr["p." + k] = v
    if not train:
        torch.manual_seed(seed + 7)
        with torch.no_grad():
            _, scores = model.full_sort_predict(batch)
        with torch.no_grad():
            torch.manual_seed(seed + 7)
            o_scores = O.full_sort_predict(batch, P, mcfg)
        print(f"  logits: max abs diff {check(name + '.logits', o_scores, scores, 2e-5):.3g}")
        arr["out.logits"] = scores

    # -- two-pass trainer protocol (recbole/trainer/trainer.py:662-686) ----------------------
    model.zero_grad()
    torch.manual_seed(seed + 11)
    att_loss, cal_loss = model.calculate_loss(batch)
    for n, prm in model.named_parameters():
        prm.requires_grad = not O.is_attack_param(n)
    cal_loss.backward(retain_graph=True)
    for n, prm in model.named_parameters():
        prm.requires_grad = O.is_attack_param(n)
    att_loss.backward()
    for n, prm in model.named_parameters():
        prm.requires_grad = True
    ref_grads = {n: (prm.grad.detach().clone() if prm.grad is not None else torch.zeros_like(prm))
                 for n, prm in model.named_parameters()}

    # regenerate the same draws in the reference's order, run the restatement with them
    torch.manual_seed(seed + 11)
    keep_emb = torch.empty(B, L, H).bernoulli_(0.5) if train else None
    rnds = [O.draw_layer_randomness((B, h, L, L), (B, L, H), ecfg, train) for _ in range(n_layers)]
    o_att, o_cal, o_grads = O.two_pass_grads(batch, P, mcfg, train, rnds, keep_emb)
    print(f"  losses: ref ({att_loss.item():.6f}, {cal_loss.item():.6f})  restated ({o_att.item():.6f}, {o_cal.item():.6f})")
    check(name + ".att_loss", o_att, att_loss.detach(), 2e-5)
    check(name + ".cal_loss", o_cal, cal_loss.detach(), 2e-5)
    worst = 0.0
    for n, g in ref_grads.items():
        scale = max(g.abs().max().item(), 1e-6)
        worst = max(worst, check(name + ".grad." + n, o_grads[n] / scale, g / scale, 2e-3))
    print(f"  grads: worst relative-to-max diff {worst:.3g}")
    arr["out.att_loss"] = att_loss.detach()
    arr["out.cal_loss"] = cal_loss.detach()
    for n, g in ref_grads.items():
        arr["grad." + n] = g
    if train:
        arr["in.keep_emb"] = keep_emb.to(torch.uint8)
    for i, r in enumerate(rnds):
        arr[f"in.noise.{i}"] = r.noise
        if train:
            for f in ("keep_after", "keep_before", "keep_mask", "keep_out_att", "keep_out_cal", "keep_ffn_att",
                      "keep_ffn_cal"):
                arr[f"in.{f}.{i}"] = getattr(r, f).to(torch.uint8)
    save(name, arr)


def bert_case(name, *, B, L, H, h, inner, n_layers, n_items, combine="gate", sigma=0.02, seed=0, mask_ratio=0.2,
              mask_loss_weight=0.03, with_scores=False, use_pos=True):
    """AcBERT4Rec (acbert4rec.py): the cloze reconstruction under a seeded `random`, the two losses and the two-pass
    gradients in eval mode (the only RNG consumers are then `random` for the cloze and one randn per layer), and --
    for combine options that accept L+1 columns -- the full-sort scores."""
    import random
    print(f"[bert] {name}")
    gen = torch.Generator().manual_seed(seed)
    cfg = Cfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner, hidden_dropout_prob=0.5,
              attn_dropout_prob=0.5, hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, loss_type="CE",
              combine_option=combine, rich_calibrated_combine="none", two_level=True, use_position_embedding=use_pos,
              use_order=True, use_distance=True, trainable_mask_loss_weight=False, mask_loss_weight=mask_loss_weight,
              mask_ratio=mask_ratio, USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", LIST_SUFFIX="_list",
              ITEM_LIST_LENGTH_FIELD="item_length", NEG_PREFIX="neg_", MAX_ITEM_LIST_LENGTH=L, device="cpu")
    torch.manual_seed(seed)
    model = AcBERT4Rec(cfg, FakeDataset(n_items))
    if sigma != 0.02:
        reinit(model, sigma, gen)
    lens = [int(v) for v in torch.randint(2, L + 1, (B,), generator=gen)]
    lens[0], lens[1] = L, 2
    item_seq = make_item_seq(B, L, n_items, lens, gen)
    item_len = torch.tensor(lens, dtype=torch.long)
    ecfg = O.EncoderCfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner, combine_option=combine,
                        rich_calibrated_combine="none", seq_length=50)
    mcfg = O.ModelCfg(enc=ecfg, n_items=n_items, max_seq_length=L, mask_loss_weight=mask_loss_weight,
                      use_position_embedding=use_pos, bidirectional=True)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.eval()

    # -- cloze reconstruction: reference vs restatement under the same `random` seed ---------------
    random.seed(seed + 3)
    masked, pos, neg, idx = model.reconstruct_train_data(item_seq)
    random.seed(seed + 3)
    o_masked, o_pos, o_neg, o_idx = O.cloze_mask_host(item_seq, mask_ratio, n_items, n_items, int(mask_ratio * L))
    for a, b_ in ((masked, o_masked), (pos, o_pos), (neg, o_neg), (idx, o_idx)):
        assert torch.equal(a, b_), "cloze reconstruction differs"
    assert (masked == n_items).any()

    arr = {"in.item_id_list": item_seq, "in.item_length": item_len, "in.random_seed": seed + 3,
           "in.masked_seq": masked, "in.pos_items": pos, "in.neg_items": neg, "in.masked_index": idx,
           "meta.cfg": np.array([n_layers, h, H, inner, L, n_items]), "meta.combine": combine,
           "meta.mask_ratio": mask_ratio, "meta.mask_loss_weight": mask_loss_weight, "meta.use_pos": int(use_pos)}
    for k, v in P.items():
        arr["p." + k] = v

    # -- two-pass trainer protocol on calculate_loss ------------------------------------------------
    model.zero_grad()
    random.seed(seed + 3)
    torch.manual_seed(seed + 11)
    att_loss, cal_loss = model.calculate_loss({"item_id_list": item_seq})
    for n, prm in model.named_parameters():
        prm.requires_grad = not O.is_attack_param(n)
    cal_loss.backward(retain_graph=True)
    for n, prm in model.named_parameters():
        prm.requires_grad = O.is_attack_param(n)
    att_loss.backward()
    for n, prm in model.named_parameters():
        prm.requires_grad = True
    ref_grads = {n: (prm.grad.detach().clone() if prm.grad is not None else torch.zeros_like(prm))
                 for n, prm in model.named_parameters()}
    torch.manual_seed(seed + 11)
    rnds = [O.draw_layer_randomness((B, h, L, L), (B, L, H), ecfg, False) for _ in range(n_layers)]
    o_att, o_cal, o_grads = O.bert_two_pass_grads(masked, pos, idx, P, mcfg, False, rnds)
    print(f"  losses: ref ({att_loss.item():.6f}, {cal_loss.item():.6f})  restated ({o_att.item():.6f}, {o_cal.item():.6f})")
    check(name + ".att_loss", o_att, att_loss.detach(), 2e-5)
    check(name + ".cal_loss", o_cal, cal_loss.detach(), 2e-5)
    worst = 0.0
    for n, g in ref_grads.items():
        scale = max(g.abs().max().item(), 1e-6)
        worst = max(worst, check(name + ".grad." + n, o_grads[n] / scale, g / scale, 2e-3))
    print(f"  grads: worst relative-to-max diff {worst:.3g}")
    arr["out.att_loss"] = att_loss.detach()
    arr["out.cal_loss"] = cal_loss.detach()
    for n, g in ref_grads.items():
        arr["grad." + n] = g
    for i, r in enumerate(rnds):
        arr[f"in.noise.{i}"] = r.noise

    # -- full-sort scores (L + 1 columns: impossible with the gate, whose width is pinned to L, and with the
    #    position embedding, which has only L rows: acbert4rec.py:47,152-160) ----------------------------
    if with_scores:
        torch.manual_seed(seed + 7)
        with torch.no_grad():
            att_scores, scores = model.full_sort_predict({"item_id_list": item_seq, "item_length": item_len})
        torch.manual_seed(seed + 7)
        rn = [O.draw_layer_randomness((B, h, L + 1, L + 1), (B, L + 1, H), ecfg, False) for _ in range(n_layers)]
        with torch.no_grad():
            o_as, o_s = O.bert_full_sort_predict(item_seq, item_len, P, mcfg, rn)
        print(f"  scores: max abs diff {check(name + '.scores', o_s, scores, 2e-5):.3g}, "
              f"attacked {check(name + '.att_scores', o_as, att_scores, 2e-5):.3g}")
        arr["out.scores"] = scores
        arr["out.att_scores"] = att_scores
        for i, r in enumerate(rn):
            arr[f"in.noise_eval.{i}"] = r.noise
    save(name, arr)


def main():
    only = set(sys.argv[1:])

    def want(n):
        return not only or n in only

    E = dict(B=4, L=50, H=64, h=2, inner=256, n_layers=2)
    if want("enc_gate_init"):
        encoder_case("enc_gate_init", **E, combine="gate", sigma=0.02, seed=0, lens=[1, 50, 17, 33])
    if want("enc_gate_stress"):
        encoder_case("enc_gate_stress", **E, combine="gate", sigma=0.3, seed=1, lens=[50, 1, 26, 41])
    if want("enc_gate_h4"):
        encoder_case("enc_gate_h4", B=3, L=50, H=64, h=4, inner=128, n_layers=3, combine="gate", sigma=0.1, seed=42,
                     keep="few")
    if want("enc_fixed_dist"):
        encoder_case("enc_fixed_dist", **E, combine="fixed", use_order=False, sigma=0.3, seed=2, keep="few")
    if want("enc_fixed_order_bidir"):
        encoder_case("enc_fixed_order_bidir", **E, combine="fixed", use_distance=False, bidirectional=True, sigma=0.3,
                     seed=3, keep="few")
    if want("enc_gate_bidir"):
        encoder_case("enc_gate_bidir", **E, combine="gate", bidirectional=True, sigma=0.2, seed=4, keep="few")
    if want("enc_plain"):
        encoder_case("enc_plain", **E, combine="gate", use_order=False, use_distance=False, sigma=0.2, seed=5,
                     keep="few")
    if want("enc_onelevel"):
        encoder_case("enc_onelevel", **E, combine="gate", two_level=False, rich="fixed", sigma=0.2, seed=6, keep="few")
    if want("enc_onelevel_trainable"):
        encoder_case("enc_onelevel_trainable", **E, combine="gate", two_level=False, rich="trainable", sigma=0.2,
                     seed=12, keep="few")
    if want("enc_anneal"):
        encoder_case("enc_anneal", B=2, L=50, H=64, h=2, inner=256, n_layers=1, combine="annealing", sigma=0.2, seed=7,
                     keep="few")
    if want("enc_leftpad"):
        encoder_case("enc_leftpad", B=3, L=50, H=64, h=2, inner=256, n_layers=1, combine="gate", sigma=0.2, seed=8,
                     lens=[20, 50, 3], left_pad_rows=(0, 2), keep="few")
    if want("enc_L200_h4"):
        encoder_case("enc_L200_h4", B=2, L=200, H=128, h=4, inner=256, n_layers=1, combine="gate", sigma=0.1, seed=9,
                     lens=[200, 77], keep="few")
    if want("enc_L200_d64_bidir"):
        encoder_case("enc_L200_d64_bidir", B=1, L=200, H=256, h=4, inner=256, n_layers=1, combine="gate",
                     bidirectional=True, sigma=0.05, seed=10, lens=[131], keep="few")
    if want("enc_L37_ragged"):
        encoder_case("enc_L37_ragged", B=3, L=37, H=64, h=2, inner=128, n_layers=1, combine="gate", sigma=0.2, seed=11,
                     keep="few")

    if want("model_eval"):
        model_case("model_eval", B=8, L=50, H=64, h=2, inner=256, n_layers=2, n_items=500, seed=0, train=False)
    if want("model_eval_stress"):
        model_case("model_eval_stress", B=6, L=50, H=64, h=2, inner=256, n_layers=2, n_items=400, seed=1, sigma=0.15,
                   train=False)
    if want("model_train"):
        model_case("model_train", B=6, L=50, H=64, h=2, inner=256, n_layers=2, n_items=400, seed=2, train=True)

    # the hyper-parameters the reference ships for Amazon-Beauty (config/amazon-beauty.yaml:33-36: 3 layers, 4 heads -> head
    # size 16, inner 128), BASELINE configs[0]'s dataset; eval logits + the two-pass gradients
    if want("model_beauty"):
        model_case("model_beauty", B=6, L=50, H=64, h=4, inner=128, n_layers=3, n_items=400, seed=5, sigma=0.05, train=False)

    if want("bert_gate"):
        bert_case("bert_gate", B=6, L=50, H=64, h=2, inner=256, n_layers=2, n_items=400, combine="gate", seed=3)
    if want("bert_fixed_scores"):
        bert_case("bert_fixed_scores", B=5, L=50, H=64, h=2, inner=256, n_layers=2, n_items=300, combine="fixed",
                  sigma=0.1, seed=4, mask_ratio=0.3, with_scores=True, use_pos=False)


if __name__ == "__main__":
    main()
