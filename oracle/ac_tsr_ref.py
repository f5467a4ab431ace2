"""CPU oracle for the AC-TSR calibrated-attention hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, torch-CPU restatement of the reference
algorithm.  It is NOT part of the product: only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it, and there only as the
checker / the timed CPU baseline.  The product path (`ac_tsr_amd/`) never imports it.

Parity status: PINNED.  `oracle/gen_golden.py` imports the genuine reference
(`/root/reference`, three logging-only stub modules) in the build container,
runs it on seeded inputs and commits inputs + expected outputs under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function below
against those vectors (the reference ships no tests / golden vectors of its own,
SURVEY.md section 4).

Every function cites the reference lines it follows (paths relative to
/root/reference/).  Parameters travel as a flat dict keyed by the reference's
state-dict names, e.g. ``p["attack_attention.query.weight"]``
for one layer, ``P["trm_encoder.layer.0.gate.weight"]`` for a model.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
MASK_FILL = -10000.0  # recbole/model/abstract_recommender.py:142
LOG_EPS = 1e-24  # recbole/model/layers.py:719


@dataclass
class EncoderCfg:
    """Constructor arguments of AttackRTransformerEncoder (recbole/model/layers.py:1071-1087)."""

    n_layers: int = 2
    n_heads: int = 2
    hidden_size: int = 64
    inner_size: int = 256
    hidden_dropout_prob: float = 0.5
    attn_dropout_prob: float = 0.5
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-12
    combine_option: str = "fixed"
    use_order: bool = True
    use_distance: bool = True
    two_level: bool = True
    rich_calibrated_combine: str = "fixed"
    seq_length: int = 50


@dataclass
class LayerRandomness:
    """Explicit randomness of ONE AttackRTransformerLayer.forward call.

    The reference consumes the global CPU generator in this order (training mode):
    dropout(after_spatial) layers.py:736/739, dropout(before_spatial) :736/740,
    dropout(attack mask) :672, randn noise :917, out_dropout(attacked) :682,
    out_dropout(calibrated) :682, ffn dropout(attacked) :795, ffn dropout(calibrated) :795.
    Keep-masks hold 0/1; scaling by 1/(1-p) is applied here like nn.Dropout does.
    In eval mode only `noise` is drawn.
    """

    noise: Optional[Tensor] = None  # [B,h,L,L]
    keep_after: Optional[Tensor] = None  # [B,h,L,L]
    keep_before: Optional[Tensor] = None  # [B,h,L,L]
    keep_mask: Optional[Tensor] = None  # [B,h,L,L]
    keep_out_att: Optional[Tensor] = None  # [B,L,H]
    keep_out_cal: Optional[Tensor] = None  # [B,L,H]
    keep_ffn_att: Optional[Tensor] = None  # [B,L,H]
    keep_ffn_cal: Optional[Tensor] = None  # [B,L,H]


def _drop(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """nn.Dropout in training mode with an explicit keep mask (identity when keep is None)."""
    if keep is None:
        return x
    return x * (keep.to(x.dtype) / (1.0 - p))


def draw_layer_randomness(shape_bhll, shape_blh, cfg: EncoderCfg, train: bool) -> LayerRandomness:
    """Draw from the global torch generator in exactly the reference's order (see LayerRandomness)."""
    r = LayerRandomness()
    pa, ph = cfg.attn_dropout_prob, cfg.hidden_dropout_prob

    def bern(shape, p):
        return torch.empty(shape, dtype=torch.float32).bernoulli_(1.0 - p)

    if train:
        r.keep_after = bern(shape_bhll, pa)
        r.keep_before = bern(shape_bhll, pa)
        r.keep_mask = bern(shape_bhll, pa)
    r.noise = torch.randn(shape_bhll)
    if train:
        r.keep_out_att = bern(shape_blh, ph)
        r.keep_out_cal = bern(shape_blh, ph)
        r.keep_ffn_att = bern(shape_blh, ph)
        r.keep_ffn_cal = bern(shape_blh, ph)
    return r


# ----------------------------------------------------------------------------------------------
# recbole/model/abstract_recommender.py:130-143
# ----------------------------------------------------------------------------------------------
def attention_mask(item_seq: Tensor, bidirectional: bool = False) -> Tensor:
    """get_attention_mask: additive 0 / -10000 mask, [B,1,L,L] causal or [B,1,1,L] bidirectional."""
    valid = (item_seq != 0)[:, None, None, :]
    if not bidirectional:
        L = item_seq.size(-1)
        valid = torch.tril(valid.expand(-1, -1, L, -1))
    return torch.where(valid, 0.0, MASK_FILL)


def gather_indexes(output: Tensor, gather_index: Tensor) -> Tensor:
    """gather_indexes (abstract_recommender.py:130-134): output[b, gather_index[b], :]."""
    idx = gather_index.view(-1, 1, 1).expand(-1, -1, output.shape[-1])
    return output.gather(dim=1, index=idx).squeeze(1)


# ----------------------------------------------------------------------------------------------
# recbole/model/layers.py:614-742  AttackRMultiHeadAttention
# ----------------------------------------------------------------------------------------------
def _heads(x: Tensor, n_heads: int) -> Tensor:
    """transpose_for_scores (layers.py:652-655): [B,L,H] -> [B,L,h,dh] view."""
    B, L, H = x.shape
    return x.view(B, L, n_heads, H // n_heads)


def _lin(x: Tensor, p: Dict[str, Tensor], name: str) -> Tensor:
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def spatial_errors(q: Tensor, k: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg, materialize: bool = True
                   ) -> Tuple[Tensor, Tensor]:
    """Spatial calibrator terms error_order, error_distance (layers.py:705-727).

    q, k: [B,h,L,dh].  materialize=True follows the reference literally (builds the
    [B,h,L,L,2dh] concatenation, :705-708); materialize=False uses the rank-1 identity
    affine(q_i || k_j) = q_i.w[:dh] + k_j.w[dh:] + b that the HIP kernel uses.
    """
    B, h, L, dh = q.shape
    zeros = torch.zeros(B, h, L, L)
    e_order, e_dist = zeros, zeros

    def affine(name):
        w, b = p[name + ".weight"], p[name + ".bias"]
        if materialize:
            qv = q.unsqueeze(3).expand(B, h, L, L, dh)
            kv = k.unsqueeze(2).expand(B, h, L, L, dh)
            return F.linear(torch.cat((qv, kv), dim=-1), w, b).squeeze(-1)
        return (q @ w[0, :dh]).unsqueeze(-1) + (k @ w[0, dh:]).unsqueeze(-2) + b

    if cfg.use_order:
        gd = torch.triu(torch.ones(L, L), diagonal=1)[None, None].expand(B, h, L, L)
        pr = torch.sigmoid(affine("attack_attention.order_affine"))
        e_order = torch.log(pr + LOG_EPS) * gd + torch.log(1 - pr + LOG_EPS) * (1 - gd)
    if cfg.use_distance:
        ar = torch.arange(0, L, 1)
        gd = torch.log(torch.abs(ar[None, :] - ar[:, None]) + 1)[None, None].expand(B, h, L, L)
        pr = affine("attack_attention.distance_affine")
        e_dist = -torch.square(gd - pr) * torch.square(p["attack_attention.scalar"]) / 2
    return e_order, e_dist


def origin_qkv(x: Tensor, mask: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg,
               keep_after: Optional[Tensor] = None, keep_before: Optional[Tensor] = None,
               materialize: bool = True):
    """cal_origin_qkv (layers.py:686-742).

    Returns (mixed_query [B,L,H], mixed_key [B,L,H], value [B,h,L,dh],
             after_spatial [B,h,L,L], before_spatial [B,h,L,L]).
    """
    h = cfg.n_heads
    mq = _lin(x, p, "attack_attention.query")
    mk = _lin(x, p, "attack_attention.key")
    mv = _lin(x, p, "attack_attention.value")
    q = _heads(mq, h).permute(0, 2, 1, 3)
    k = _heads(mk, h).permute(0, 2, 1, 3)
    v = _heads(mv, h).permute(0, 2, 1, 3)
    raw = torch.matmul(q, k.transpose(-1, -2))  # unscaled, layers.py:695
    e_order, e_dist = spatial_errors(q, k, p, cfg, materialize)
    calibrated = raw + e_order + e_dist  # layers.py:729
    sqrt_dh = math.sqrt(q.shape[-1])
    pa = cfg.attn_dropout_prob

    def prob(scores, keep):  # _func, layers.py:731-737
        return _drop(torch.softmax(scores / sqrt_dh + mask, dim=-1), keep, pa)

    return mq, mk, v, prob(calibrated, keep_after), prob(raw, keep_before)


def attack_mask(mq: Tensor, mk: Tensor, mask: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg,
                keep: Optional[Tensor] = None) -> Tensor:
    """cal_attack_mask (layers.py:657-674): softmax(Qa.Ka^T/sqrt(dh) + mask) from the MIXED q/k."""
    h = cfg.n_heads
    qa = _heads(_lin(mq, p, "attack_attention.attack_query_transform"), h).permute(0, 2, 1, 3)
    ka = _heads(_lin(mk, p, "attack_attention.attack_key_transform"), h).permute(0, 2, 3, 1)
    s = torch.matmul(qa, ka) / math.sqrt(qa.shape[-1]) + mask
    return _drop(torch.softmax(s, dim=-1), keep, cfg.attn_dropout_prob)


def adjusted_outputs(prob: Tensor, x: Tensor, v: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg,
                     keep: Optional[Tensor] = None) -> Tensor:
    """cal_adjusted_outputs (layers.py:676-684): P.V, head merge, dense, dropout, LN(+residual)."""
    ctx = torch.matmul(prob, v).permute(0, 2, 1, 3).contiguous()
    ctx = ctx.view(ctx.shape[0], ctx.shape[1], -1)
    hid = _drop(_lin(ctx, p, "attack_attention.dense"), keep, cfg.hidden_dropout_prob)
    return F.layer_norm(hid + x, (x.shape[-1],), p["attack_attention.LayerNorm.weight"],
                        p["attack_attention.LayerNorm.bias"], cfg.layer_norm_eps)


def context_only(prob: Tensor, v: Tensor) -> Tensor:
    """First two lines of cal_adjusted_outputs (layers.py:677-680): head-merged P.V -> [B,L,H]."""
    ctx = torch.matmul(prob, v).permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.shape[0], ctx.shape[1], -1)


# ----------------------------------------------------------------------------------------------
# recbole/model/layers.py:745-798  FeedForward
# ----------------------------------------------------------------------------------------------
def _act(name: str):
    if name == "gelu":  # exact-erf gelu, layers.py:776-785
        return lambda t: t * 0.5 * (1.0 + torch.erf(t / math.sqrt(2.0)))
    if name == "relu":
        return F.relu
    if name == "swish":
        return lambda t: t * torch.sigmoid(t)
    if name == "tanh":
        return torch.tanh
    if name == "sigmoid":
        return torch.sigmoid
    raise KeyError(name)


def feed_forward(x: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg, keep: Optional[Tensor] = None) -> Tensor:
    """FeedForward.forward (layers.py:790-798)."""
    hid = _act(cfg.hidden_act)(_lin(x, p, "feed_forward.dense_1"))
    hid = _drop(_lin(hid, p, "feed_forward.dense_2"), keep, cfg.hidden_dropout_prob)
    return F.layer_norm(hid + x, (x.shape[-1],), p["feed_forward.LayerNorm.weight"],
                        p["feed_forward.LayerNorm.bias"], cfg.layer_norm_eps)


# ----------------------------------------------------------------------------------------------
# recbole/model/layers.py:859-951  AttackRTransformerLayer
# ----------------------------------------------------------------------------------------------
def combine_attention(origin: Tensor, calibrated: Tensor, mq: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg,
                      anneal_step: int = 0) -> Tensor:
    """combine_attention (layers.py:883-896)."""
    if cfg.combine_option == "fixed":
        return torch.softmax(origin + 0.5 * calibrated, dim=-1)
    if cfg.combine_option == "gate":
        g = torch.sigmoid(_lin(mq, p, "gate")).unsqueeze(1)
        return g * origin + (1 - g) * calibrated
    if cfg.combine_option == "annealing":
        rate = math.exp(-anneal_step / 100000)
        return rate * origin + (1 - rate) * calibrated
    raise KeyError(cfg.combine_option)


def layer_forward(x: Tensor, mask: Tensor, p: Dict[str, Tensor], cfg: EncoderCfg, rnd: LayerRandomness,
                  materialize: bool = True, anneal_step: int = 0):
    """AttackRTransformerLayer.forward (layers.py:898-951).

    Returns (attacked_out [B,L,H], calibrated_out [B,L,H], attack_mask M [B,h,L,L],
             combined prob [B,h,L,L], dict of intermediates).  `dbg` additionally carries the
    head-merged contexts (ctx_attacked / ctx_calibrated, [B,L,H]) and mixed q/k/v that the HIP
    core consumes / produces.
    """
    mq, mk, v, after, before = origin_qkv(x, mask, p, cfg, rnd.keep_after, rnd.keep_before, materialize)
    origin = after if cfg.two_level else before  # layers.py:911-914
    M = attack_mask(mq, mk, mask, p, cfg, rnd.keep_mask)  # :915
    noise = rnd.noise
    attacked = torch.softmax(origin * M + noise * (1 - M) + mask, dim=-1)  # :918-919
    calibrated = torch.softmax(origin * torch.exp(1 - M) + mask, dim=-1)  # :920-921
    combined = combine_attention(origin, calibrated, mq, p, cfg, anneal_step)  # :922-924
    combined = torch.softmax(combined + mask, dim=-1)  # :925
    probs = {
        "before_spatial": before, "after_spatial": after, "perturbed_mask": M,
        "perturbed_attention": attacked, "calibrated_attention": combined,
    }
    if not cfg.two_level:  # :929-936
        if cfg.rich_calibrated_combine == "fixed":
            combined = (combined + after) / 2
        elif cfg.rich_calibrated_combine == "trainable":
            r = p["rich_calibrated_combine_ratio"]
            combined = r * combined + (1 - r) * after
        else:
            raise KeyError(cfg.rich_calibrated_combine)
    att_attn = adjusted_outputs(attacked, x, v, p, cfg, rnd.keep_out_att)  # :938-940
    cal_attn = adjusted_outputs(combined, x, v, p, cfg, rnd.keep_out_cal)  # :942-944
    att_out = feed_forward(att_attn, p, cfg, rnd.keep_ffn_att)  # :946
    cal_out = feed_forward(cal_attn, p, cfg, rnd.keep_ffn_cal)  # :947
    dbg = dict(probs)
    dbg.update(mixed_query=mq, mixed_key=mk, value=v, ctx_attacked=context_only(attacked, v),
               ctx_calibrated=context_only(combined, v), final_combined=combined)
    return att_out, cal_out, M, combined, dbg


# ----------------------------------------------------------------------------------------------
# recbole/model/layers.py:1070-1131  AttackRTransformerEncoder
# ----------------------------------------------------------------------------------------------
def layer_params(P: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


def encoder_forward(x: Tensor, mask: Tensor, P: Dict[str, Tensor], cfg: EncoderCfg,
                    rnds: Optional[List[LayerRandomness]] = None, train: bool = False,
                    materialize: bool = True, prefix: str = "layer."):
    """AttackRTransformerEncoder.forward (layers.py:1097-1131), output_all_encoded_layers=True.

    rnds=None draws the randomness from the global generator in reference order.
    Returns (list[(attacked, calibrated)], list[M], list[dbg]).
    """
    B, L, H = x.shape
    outs, masks, dbgs = [], [], []
    hidden = x
    for i in range(cfg.n_layers):
        rnd = rnds[i] if rnds is not None else draw_layer_randomness((B, cfg.n_heads, L, L), (B, L, H), cfg, train)
        att, cal, M, _, dbg = layer_forward(hidden, mask, layer_params(P, f"{prefix}{i}."), cfg, rnd, materialize)
        hidden = cal  # layers.py:1112
        outs.append((att, cal))
        masks.append(M)
        dbgs.append(dbg)
    return outs, masks, dbgs


# ----------------------------------------------------------------------------------------------
# recbole/model/sequential_recommender/acsasrec.py:86-164  ACSASRec
# ----------------------------------------------------------------------------------------------
@dataclass
class ModelCfg:
    enc: EncoderCfg = field(default_factory=EncoderCfg)
    n_items: int = 1000
    max_seq_length: int = 50
    use_position_embedding: bool = False
    loss_type: str = "CE"
    mask_loss_weight: float = 0.3
    trainable_mask_loss_weight: bool = False
    bidirectional: bool = False  # ACSASRec is causal; True gives AcBERT4Rec's mask flavour (acbert4rec.py:173)


def model_forward(item_seq: Tensor, item_seq_len: Tensor, P: Dict[str, Tensor], cfg: ModelCfg,
                  train: bool = False, rnds=None, keep_emb: Optional[Tensor] = None, materialize: bool = True):
    """ACSASRec.forward (acsasrec.py:86-104) -> (attacked [B,H], calibrated [B,H], list[M], dbgs)."""
    emb = F.embedding(item_seq, P["item_embedding.weight"])
    if cfg.use_position_embedding:
        pos = torch.arange(item_seq.size(1), dtype=torch.long)
        emb = emb + F.embedding(pos, P["position_embedding.weight"]).unsqueeze(0)
    H = emb.shape[-1]
    emb = F.layer_norm(emb, (H,), P["LayerNorm.weight"], P["LayerNorm.bias"], cfg.enc.layer_norm_eps)
    if train and keep_emb is None and rnds is None:
        keep_emb = torch.empty(emb.shape).bernoulli_(1.0 - cfg.enc.hidden_dropout_prob)  # acsasrec.py:95
    emb = _drop(emb, keep_emb, cfg.enc.hidden_dropout_prob)
    mask = attention_mask(item_seq, cfg.bidirectional)
    outs, masks, dbgs = encoder_forward(emb, mask, P, cfg.enc, rnds, train, materialize, prefix="trm_encoder.layer.")
    att, cal = outs[-1]
    return gather_indexes(att, item_seq_len - 1), gather_indexes(cal, item_seq_len - 1), masks, dbgs


def calculate_loss(batch: Dict[str, Tensor], P: Dict[str, Tensor], cfg: ModelCfg, train: bool = True, rnds=None,
                   keep_emb=None, materialize: bool = True):
    """ACSASRec.calculate_loss (acsasrec.py:107-144), CE loss -> (final_attacked_loss, calibrated_loss)."""
    att, cal, masks, _ = model_forward(batch["item_id_list"], batch["item_length"], P, cfg, train, rnds, keep_emb,
                                       materialize)
    E = P["item_embedding.weight"]
    pos = batch["item_id"]
    attacked_loss = -F.cross_entropy(att @ E.t(), pos)
    penalty = torch.stack([torch.norm(1 - M, p=2) for M in masks]).mean()
    w = P["mask_loss_weight"][0] if cfg.trainable_mask_loss_weight else cfg.mask_loss_weight
    return attacked_loss + penalty * w, F.cross_entropy(cal @ E.t(), pos)


def full_sort_predict(batch: Dict[str, Tensor], P: Dict[str, Tensor], cfg: ModelCfg, rnds=None) -> Tensor:
    """ACSASRec.full_sort_predict (acsasrec.py:157-164): calibrated_out @ E^T -> [B, n_items]."""
    _, cal, _, _ = model_forward(batch["item_id_list"], batch["item_length"], P, cfg, False, rnds)
    return cal @ P["item_embedding.weight"].t()


def is_attack_param(name: str) -> bool:
    """Parameter partition used by the two-pass trainer (recbole/trainer/trainer.py:672-683)."""
    return "attack_key_transform" in name or "attack_query_transform" in name


def two_pass_grads(batch, P: Dict[str, Tensor], cfg: ModelCfg, train: bool = True, rnds=None, keep_emb=None,
                   materialize: bool = True):
    """Gradients left in .grad by AttackSASRecTrainer._train_epoch's two backward passes
    (recbole/trainer/trainer.py:672-686): non-attack params get d(calibrated_loss), attack
    transforms get d(final_attacked_loss).  Returns (attacked_loss, calibrated_loss, grads dict)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    att_loss, cal_loss = calculate_loss(batch, leaves, cfg, train, rnds, keep_emb, materialize)
    names = list(leaves)
    g_cal = torch.autograd.grad(cal_loss, [leaves[n] for n in names], retain_graph=True, allow_unused=True)
    g_att = torch.autograd.grad(att_loss, [leaves[n] for n in names], allow_unused=True)
    grads = {}
    for n, gc, ga in zip(names, g_cal, g_att):
        g = ga if is_attack_param(n) else gc
        grads[n] = torch.zeros_like(P[n]) if g is None else g
    return att_loss.detach(), cal_loss.detach(), grads


# ----------------------------------------------------------------------------------------------
# recbole/model/sequential_recommender/acbert4rec.py  AcBERT4Rec (cloze training over the same encoder)
# ----------------------------------------------------------------------------------------------
def cloze_mask_host(item_seq: Tensor, mask_ratio: float, mask_token: int, n_items: int, mask_item_length: int,
                    rng=None):
    """AcBERT4Rec.reconstruct_train_data (acbert4rec.py:105-150): walks every sequence up to its first padding,
    masks an item with probability mask_ratio and draws one negative per masked item.  Consumes Python's `random`
    stream call for call like the reference (one random() per real item, then randint() until the draw is not in
    the sequence), so seeding `random` reproduces its output.  Returns (masked_seq [B,L], pos [B,ml], neg [B,ml],
    masked_index [B,ml]); the three short lists are LEFT-padded with zeros and keep their LAST ml entries
    (acbert4rec.py:98-102)."""
    import random as _random
    rng = rng or _random
    rows = item_seq.cpu().tolist()
    masked_rows, pos_rows, neg_rows, idx_rows = [], [], [], []

    def fit(values):
        return ([0] * (mask_item_length - len(values)) + values)[-mask_item_length:] if mask_item_length > 0 else []

    for row in rows:
        masked = list(row)
        pos, neg, idx = [], [], []
        for j, item in enumerate(row):
            if item == 0:
                break
            if rng.random() < mask_ratio:
                draw = rng.randint(1, n_items - 1)
                while draw in row:
                    draw = rng.randint(1, n_items - 1)
                pos.append(item)
                neg.append(draw)
                idx.append(j)
                masked[j] = mask_token
        masked_rows.append(masked)
        pos_rows.append(fit(pos))
        neg_rows.append(fit(neg))
        idx_rows.append(fit(idx))
    as_t = lambda v: torch.tensor(v, dtype=torch.long).view(len(rows), -1)
    return as_t(masked_rows), as_t(pos_rows), as_t(neg_rows), as_t(idx_rows)


def bert_forward(item_seq: Tensor, P: Dict[str, Tensor], cfg: ModelCfg, train: bool = False, rnds=None,
                 keep_emb: Optional[Tensor] = None, materialize: bool = True):
    """AcBERT4Rec.forward (acbert4rec.py:162-178) -> (attacked [B,L,H], calibrated [B,L,H], list[M])."""
    emb = F.embedding(item_seq, P["item_embedding.weight"])
    if cfg.use_position_embedding:
        pos = torch.arange(item_seq.size(1), dtype=torch.long)
        emb = emb + F.embedding(pos, P["position_embedding.weight"]).unsqueeze(0)
    H = emb.shape[-1]
    emb = F.layer_norm(emb, (H,), P["LayerNorm.weight"], P["LayerNorm.bias"], cfg.enc.layer_norm_eps)
    if train and keep_emb is None and rnds is None:
        keep_emb = torch.empty(emb.shape).bernoulli_(1.0 - cfg.enc.hidden_dropout_prob)
    emb = _drop(emb, keep_emb, cfg.enc.hidden_dropout_prob)
    mask = attention_mask(item_seq, bidirectional=True)
    outs, masks, _ = encoder_forward(emb, mask, P, cfg.enc, rnds, train, materialize, prefix="trm_encoder.layer.")
    att, cal = outs[-1]
    return att, cal, masks


def bert_masked_ce(seq_output: Tensor, pos_items: Tensor, targets: Tensor, P: Dict[str, Tensor], cfg: ModelCfg) -> Tensor:
    """AcBERT4Rec._cal_loss (acbert4rec.py:201-209): CE over the catalogue WITHOUT the mask-token row, averaged over
    the real (non-padding) masked slots."""
    E = P["item_embedding.weight"][:cfg.n_items]
    logits = seq_output @ E.t()
    per_slot = F.cross_entropy(logits.view(-1, E.size(0)), pos_items.view(-1), reduction="none")
    return torch.sum(per_slot * targets) / torch.sum(targets)


def bert_calculate_loss(masked_seq: Tensor, pos_items: Tensor, masked_index: Tensor, P: Dict[str, Tensor],
                        cfg: ModelCfg, train: bool = True, rnds=None, keep_emb=None, materialize: bool = True):
    """AcBERT4Rec.calculate_loss after the cloze reconstruction (acbert4rec.py:215-240): the hidden rows at the
    masked positions are picked with the reference's multi-hot bmm (acbert4rec.py:180-199, 219-225)."""
    att, cal, masks = bert_forward(masked_seq, P, cfg, train, rnds, keep_emb, materialize)
    B, ml = masked_index.shape
    multi_hot = torch.zeros(B * ml, masked_seq.size(-1))
    multi_hot[torch.arange(B * ml), masked_index.view(-1)] = 1
    multi_hot = multi_hot.view(B, ml, -1)
    att_rows, cal_rows = torch.bmm(multi_hot, att), torch.bmm(multi_hot, cal)
    targets = (masked_index > 0).float().view(-1)  # (index 0 doubles as the padding marker, as in the reference)
    attacked_loss = -bert_masked_ce(att_rows, pos_items, targets, P, cfg)
    penalty = torch.stack([torch.norm(1 - M, p=2) for M in masks]).mean()
    w = P["mask_loss_weight"][0] if cfg.trainable_mask_loss_weight else cfg.mask_loss_weight
    return attacked_loss + penalty * w, bert_masked_ce(cal_rows, pos_items, targets, P, cfg)


def bert_two_pass_grads(masked_seq, pos_items, masked_index, P: Dict[str, Tensor], cfg: ModelCfg, train: bool = True,
                        rnds=None, keep_emb=None, materialize: bool = True):
    """two_pass_grads for AcBERT4Rec (same trainer protocol, recbole/trainer/trainer.py:672-686)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    att_loss, cal_loss = bert_calculate_loss(masked_seq, pos_items, masked_index, leaves, cfg, train, rnds, keep_emb,
                                             materialize)
    names = list(leaves)
    g_cal = torch.autograd.grad(cal_loss, [leaves[n] for n in names], retain_graph=True, allow_unused=True)
    g_att = torch.autograd.grad(att_loss, [leaves[n] for n in names], allow_unused=True)
    grads = {}
    for n, gc, ga in zip(names, g_cal, g_att):
        g = ga if is_attack_param(n) else gc
        grads[n] = torch.zeros_like(P[n]) if g is None else g
    return att_loss.detach(), cal_loss.detach(), grads


def bert_append_mask_token(item_seq: Tensor, item_seq_len: Tensor, mask_token: int) -> Tensor:
    """AcBERT4Rec.reconstruct_test_data (acbert4rec.py:152-160): one more column, mask token right after the last item."""
    out = torch.cat((item_seq, torch.zeros(item_seq.size(0), 1, dtype=torch.long)), dim=-1)
    out[torch.arange(item_seq.size(0)), item_seq_len] = mask_token
    return out


def bert_full_sort_predict(item_seq: Tensor, item_seq_len: Tensor, P: Dict[str, Tensor], cfg: ModelCfg, rnds=None):
    """AcBERT4Rec.full_sort_predict (acbert4rec.py:257-267) -> (attacked_scores, scores), [B, n_items] each."""
    seq = bert_append_mask_token(item_seq, item_seq_len, cfg.n_items)
    att, cal, _ = bert_forward(seq, P, cfg, False, rnds)
    E = P["item_embedding.weight"][:cfg.n_items]
    return gather_indexes(att, item_seq_len) @ E.t(), gather_indexes(cal, item_seq_len) @ E.t()


# ----------------------------------------------------------------------------------------------
# The attention core at the projected-tensor boundary (what the HIP kernel computes), expressed
# with the same reference ops.  Used by tests to check the C-ABI entry points directly.
# ----------------------------------------------------------------------------------------------
def core_from_projected(mq: Tensor, mk: Tensor, mv: Tensor, qa: Tensor, ka: Tensor, gate_logits: Optional[Tensor],
                        mask: Tensor, w_order, b_order, w_dist, b_dist, scalar, cfg: EncoderCfg,
                        noise: Tensor, keep_after=None, keep_mask=None, keep_before=None, anneal_rate: float = 1.0,
                        rich_ratio=None, materialize: bool = True):
    """Everything between the projections and the output dense, following layers.py:695-740,
    664-672, 917-936, 677-680 on already-projected tensors.

    mq/mk/mv/qa/ka: [B,L,H]; gate_logits: [B,L,L] (pre-sigmoid) or None; mask additive.
    Returns dict(ctx_attacked, ctx_calibrated [B,L,H]; M, after, before, attacked, combined [B,h,L,L]).
    """
    h = cfg.n_heads
    q = _heads(mq, h).permute(0, 2, 1, 3)
    k = _heads(mk, h).permute(0, 2, 1, 3)
    v = _heads(mv, h).permute(0, 2, 1, 3)
    dh = q.shape[-1]
    p = {
        "attack_attention.order_affine.weight": w_order, "attack_attention.order_affine.bias": b_order,
        "attack_attention.distance_affine.weight": w_dist, "attack_attention.distance_affine.bias": b_dist,
        "attack_attention.scalar": scalar,
    }
    raw = torch.matmul(q, k.transpose(-1, -2))
    # materialize=False: the affine over (q_i || k_j) in its rank-1 form (the [B,h,L,L,2dh] tensor of layers.py:705-708 is
    # 655 MB at the benchmark's batch); tests/test_oracle_golden.py pins both forms to the reference
    e_o, e_d = spatial_errors(q, k, p, cfg, materialize=materialize)
    pa = cfg.attn_dropout_prob
    after = _drop(torch.softmax((raw + e_o + e_d) / math.sqrt(dh) + mask, dim=-1), keep_after, pa)
    before = _drop(torch.softmax(raw / math.sqrt(dh) + mask, dim=-1), keep_before, pa)
    origin = after if cfg.two_level else before
    qah = _heads(qa, h).permute(0, 2, 1, 3)
    kah = _heads(ka, h).permute(0, 2, 3, 1)
    M = _drop(torch.softmax(torch.matmul(qah, kah) / math.sqrt(dh) + mask, dim=-1), keep_mask, pa)
    attacked = torch.softmax(origin * M + noise * (1 - M) + mask, dim=-1)
    calibrated = torch.softmax(origin * torch.exp(1 - M) + mask, dim=-1)
    if cfg.combine_option == "fixed":
        comb = torch.softmax(origin + 0.5 * calibrated, dim=-1)
    elif cfg.combine_option == "gate":
        g = torch.sigmoid(gate_logits).unsqueeze(1)
        comb = g * origin + (1 - g) * calibrated
    elif cfg.combine_option == "annealing":
        comb = anneal_rate * origin + (1 - anneal_rate) * calibrated
    else:
        raise KeyError(cfg.combine_option)
    comb = torch.softmax(comb + mask, dim=-1)
    final = comb
    if not cfg.two_level:
        if cfg.rich_calibrated_combine == "fixed":
            final = (comb + after) / 2
        elif cfg.rich_calibrated_combine == "trainable":  # :932-934
            final = rich_ratio * comb + (1 - rich_ratio) * after
        else:
            raise KeyError(cfg.rich_calibrated_combine)
    return dict(ctx_attacked=context_only(attacked, v), ctx_calibrated=context_only(final, v), M=M, after=after,
                before=before, attacked=attacked, combined=comb, final=final)
