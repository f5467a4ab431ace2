"""MI355X-native modules with the reference's operator surface for the calibrated encoder.

Class names, constructor signatures, forward signatures / return arities and state-dict keys are
those of recbole/model/layers.py (AttackRMultiHeadAttention :614-742, FeedForward :745-798,
AttackRTransformerLayer :859-951, AttackRTransformerEncoder :1070-1131), so a checkpoint of the
reference loads with `load_state_dict` and the AC-SASRec trainer can drive these modules unchanged.
The attention core itself is NOT a chain of torch ops: every layer makes one call into the HIP
library (`ops.calibrated_attention`); the six projections in front of it are one launch (`linear.projections`: hidden 64, else
hipBLASLt GEMMs in one autograd node), the position-wise tail behind it another (`tail.layer_tail`).

Extra, optional keyword arguments (absent from the reference) are prefixed with an underscore:
`_rnd` feeds explicit randomness for parity tests.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_ln, ops
from . import tail
from .linear import projections, skinny_linear
from .ops import AttentionConfig, ExplicitRandomness, StructuredMask
from .state import state_of


class AttackRMultiHeadAttention(nn.Module):
    """Parameters and projections of recbole/model/layers.py:614-650.  The three methods of the
    reference (`cal_origin_qkv`, `cal_attack_mask`, `cal_adjusted_outputs`) are fused: `project` yields
    the five projected tensors the HIP core consumes, `output` is the dense + residual LayerNorm tail
    of `cal_adjusted_outputs` (:681-683)."""

    def __init__(self, n_heads, hidden_size, hidden_dropout_prob, attn_dropout_prob, layer_norm_eps, use_order,
                 use_distance):
        super().__init__()
        if hidden_size % n_heads != 0:
            raise ValueError(
                "The hidden size (%d) is not a multiple of the number of attention "
                "heads (%d)" % (hidden_size, n_heads))
        self.num_attention_heads = n_heads
        self.attention_head_size = int(hidden_size / n_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.sqrt_attention_head_size = math.sqrt(self.attention_head_size)

        self.query = nn.Linear(hidden_size, self.all_head_size)
        self.key = nn.Linear(hidden_size, self.all_head_size)
        self.value = nn.Linear(hidden_size, self.all_head_size)

        self.use_order = use_order
        self.use_distance = use_distance
        if self.use_order:
            self.order_affine = nn.Linear(2 * self.attention_head_size, 1)
        if self.use_distance:
            self.distance_affine = nn.Linear(2 * self.attention_head_size, 1)
            self.scalar = nn.Parameter(torch.randn(1))

        self.attack_query_transform = nn.Linear(self.all_head_size, self.all_head_size)
        self.attack_key_transform = nn.Linear(self.all_head_size, self.all_head_size)

        self.attn_dropout_prob = attn_dropout_prob
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.out_dropout = nn.Dropout(hidden_dropout_prob)

    def project(self, input_tensor):
        """mixed q/k/v (layers.py:687-689) and the attack transforms of the MIXED q/k (:658-659)."""
        mq = skinny_linear(input_tensor, self.query)
        mk = skinny_linear(input_tensor, self.key)
        mv = skinny_linear(input_tensor, self.value)
        qa = skinny_linear(mq, self.attack_query_transform)
        ka = skinny_linear(mk, self.attack_key_transform)
        return mq, mk, mv, qa, ka

    def calibrator_params(self):
        p = {}
        if self.use_order:
            p.update(w_order=self.order_affine.weight, b_order=self.order_affine.bias)
        if self.use_distance:
            p.update(w_dist=self.distance_affine.weight, b_dist=self.distance_affine.bias, scalar=self.scalar)
        return p

    def output(self, context_layer, input_tensor, _keep=None):
        """dense -> dropout -> LayerNorm(+residual) of cal_adjusted_outputs (layers.py:681-683)."""
        hidden_states = skinny_linear(context_layer, self.dense)
        if hidden_states.is_cuda and fused_ln.supported(hidden_states.shape[-1]):
            return fused_ln.dropout_add_layer_norm(hidden_states, input_tensor, self.LayerNorm, self.out_dropout.p,
                                                   self.training, _keep)
        if _keep is not None:
            hidden_states = hidden_states * (_keep.to(hidden_states.dtype) / (1.0 - self.out_dropout.p))
        else:
            hidden_states = self.out_dropout(hidden_states)
        return self.LayerNorm(hidden_states + input_tensor)


class FeedForward(nn.Module):
    """recbole/model/layers.py:745-798 (exact-erf gelu)."""

    def __init__(self, hidden_size, inner_size, hidden_dropout_prob, hidden_act, layer_norm_eps):
        super().__init__()
        self.dense_1 = nn.Linear(hidden_size, inner_size)
        self.intermediate_act_fn = self.get_hidden_act(hidden_act)
        self.dense_2 = nn.Linear(inner_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.dropout = nn.Dropout(hidden_dropout_prob)

    def get_hidden_act(self, act):
        ACT2FN = {
            "gelu": self.gelu,
            "relu": F.relu,
            "swish": self.swish,
            "tanh": torch.tanh,
            "sigmoid": torch.sigmoid,
        }
        return ACT2FN[act]

    def gelu(self, x):
        return F.gelu(x)  # erf form == x * 0.5 * (1 + erf(x / sqrt(2)))  (layers.py:785)

    def swish(self, x):
        return x * torch.sigmoid(x)

    def forward(self, input_tensor, _keep=None):
        hidden_states = skinny_linear(self.intermediate_act_fn(skinny_linear(input_tensor, self.dense_1)), self.dense_2)
        if hidden_states.is_cuda and fused_ln.supported(hidden_states.shape[-1]):
            return fused_ln.dropout_add_layer_norm(hidden_states, input_tensor, self.LayerNorm, self.dropout.p,
                                                   self.training, _keep)
        if _keep is not None:
            hidden_states = hidden_states * (_keep.to(hidden_states.dtype) / (1.0 - self.dropout.p))
        else:
            hidden_states = self.dropout(hidden_states)
        return self.LayerNorm(hidden_states + input_tensor)


class AttackRTransformerLayer(nn.Module):
    """recbole/model/layers.py:859-951 with the attention core as one HIP launch."""

    def __init__(self, n_heads, hidden_size, intermediate_size, hidden_dropout_prob, attn_dropout_prob, hidden_act,
                 layer_norm_eps, combine_option='fixed', use_order=True, use_distance=True, two_level=True,
                 rich_calibrated_combine='fixed', seq_length=50):
        super().__init__()
        self.hidden_size = hidden_size
        self.attack_attention = AttackRMultiHeadAttention(
            n_heads, hidden_size, hidden_dropout_prob, attn_dropout_prob, layer_norm_eps,
            use_order=use_order, use_distance=use_distance)
        self.two_level = two_level
        self.rich_calibrated_combine = rich_calibrated_combine
        if self.rich_calibrated_combine == 'trainable':
            self.rich_calibrated_combine_ratio = torch.nn.Parameter(torch.FloatTensor([0.5]), requires_grad=True)
        self.combine_option = combine_option
        if self.combine_option == 'gate':
            self.gate = torch.nn.Linear(hidden_size, seq_length)
        self.combine_ratio = 0.5
        self.feed_forward = FeedForward(hidden_size, intermediate_size, hidden_dropout_prob, hidden_act, layer_norm_eps)
        self.anneal_step = 0

    def _config(self) -> AttentionConfig:
        rate = 1.0
        if self.combine_option == 'annealing':  # stateful, one tick per forward (layers.py:890-891)
            rate = math.exp(-self.anneal_step / 100000)
            self.anneal_step += 1
        return AttentionConfig(n_heads=self.attack_attention.num_attention_heads, combine_option=self.combine_option,
                               two_level=self.two_level, rich_calibrated_combine=self.rich_calibrated_combine,
                               adversarial=True, anneal_rate=rate)

    def forward(self, hidden_states, attention_mask, return_attention_prob=False, return_all_attention_prob=False,
                _rnd=None, _need_attacked=True, _attack_upstream=True, _rows=None):
        """`_need_attacked=False` (set by the encoder for layers whose attacked output nobody can observe) skips the
        dense / LayerNorm / feed-forward tail of the attacked branch and returns None in its place.
        `_rows` ([B, R] positions) makes the layer return only those positions of its two outputs ([B, R, H]): the tail
        is position-wise, so it then runs on the selected rows alone (the models read one position per sequence of the
        last layer, abstract_recommender.py:130-134; AcBERT4Rec the masked positions, acbert4rec.py:219-225)."""
        att = self.attack_attention
        cal = att.calibrator_params()
        mq, mk, mv, qa, ka, gate_logits, hidden_res, extras = projections(
            hidden_states, att.query, att.key, att.value, att.attack_query_transform, att.attack_key_transform,
            self.gate if self.combine_option == 'gate' else None, attack_upstream=_attack_upstream,
            spatial=(cal.get("w_order"), cal.get("b_order"), cal.get("w_dist"), cal.get("b_dist"), att.num_attention_heads))
        cfg = self._config()
        core_rnd = None
        if _rnd is not None:
            core_rnd = ExplicitRandomness(noise=_rnd.noise, keep_after=getattr(_rnd, "keep_after", None),
                                          keep_mask=getattr(_rnd, "keep_mask", None),
                                          keep_before=getattr(_rnd, "keep_before", None))
        p_drop = att.attn_dropout_prob if (self.training or (core_rnd is not None and core_rnd.keep_after is not None)) else 0.0
        want_probs = return_all_attention_prob or return_attention_prob
        ctx_att, ctx_cal, attack_mask, probs = ops.calibrated_attention(
            mq, mk, mv, qa, ka, gate_logits, attention_mask, cfg, p_drop=p_drop, rnd=core_rnd, want_probs=want_probs,
            seed_tensor=state_of(self).seed_tensor if core_rnd is None else None, read_rows=_rows,
            attack_upstream=_attack_upstream, state=state_of(self), rich_ratio=getattr(self, "rich_calibrated_combine_ratio", None),
            **extras, **cal)
        residual = hidden_res  # hidden_states, via the projection node (its backward launch takes the residual gradient)
        if _rows is not None:
            index = _rows.unsqueeze(-1).expand(-1, -1, hidden_states.shape[-1])
            pick = lambda t: None if t is None else t.gather(1, index)
            picks_itself = ctx_cal.is_cuda and tail.supported(att, self.feed_forward) and tail.fused_supported(att, self.feed_forward)
            if not picks_itself:
                residual = pick(hidden_res)
        else:
            pick = lambda t: t

        def branch(ctx_layer, keep_out, keep_ffn):
            one_launch = ctx_layer.is_cuda and tail.supported(att, self.feed_forward) and tail.fused_supported(att, self.feed_forward)
            fused = one_launch or (ctx_layer.is_cuda and torch.is_grad_enabled() and tail.supported(att, self.feed_forward))
            if one_launch and _rows is not None:
                # the fused tail picks the rows itself: no gather launches in front of it, no scatter launches behind
                return tail.layer_tail(ctx_layer, hidden_res, att, self.feed_forward, pick(keep_out), pick(keep_ffn),
                                       pick=_rows)
            ctx_layer, keep_out, keep_ffn = pick(ctx_layer), pick(keep_out), pick(keep_ffn)
            if fused:
                return tail.layer_tail(ctx_layer, residual, att, self.feed_forward, keep_out, keep_ffn)
            return self.feed_forward(att.output(ctx_layer, residual, keep_out), keep_ffn)

        attacked_feedforward_output = None
        if _need_attacked:
            attacked_feedforward_output = branch(ctx_att, getattr(_rnd, "keep_out_att", None),
                                                 getattr(_rnd, "keep_ffn_att", None))
        calibrated_feedforward_output = branch(ctx_cal, getattr(_rnd, "keep_out_cal", None),
                                               getattr(_rnd, "keep_ffn_cal", None))
        combined_attention_prob = probs.get("calibrated_attention")
        if return_all_attention_prob:
            all_attention_prob = {
                'before_spatial': probs["before_spatial"],
                'after_spatial': probs["after_spatial"],
                'perturbed_mask': attack_mask,
                'perturbed_attention': probs["perturbed_attention"],
                'calibrated_attention': probs["calibrated_attention"],
            }
            return (attacked_feedforward_output, calibrated_feedforward_output, attack_mask, combined_attention_prob,
                    all_attention_prob)
        return attacked_feedforward_output, calibrated_feedforward_output, attack_mask, combined_attention_prob



class AttackRTransformerEncoder(nn.Module):
    """recbole/model/layers.py:1070-1131."""

    def __init__(self, n_layers=2, n_heads=2, hidden_size=64, inner_size=256, hidden_dropout_prob=0.5,
                 attn_dropout_prob=0.5, hidden_act='gelu', layer_norm_eps=1e-12, combine_option='fixed', use_order=True,
                 use_distance=True, two_level=True, rich_calibrated_combine='fixed', seq_length=50):
        super().__init__()
        layer = AttackRTransformerLayer(
            n_heads, hidden_size, inner_size, hidden_dropout_prob, attn_dropout_prob, hidden_act, layer_norm_eps,
            combine_option, use_order=use_order, use_distance=use_distance, two_level=two_level,
            rich_calibrated_combine=rich_calibrated_combine, seq_length=seq_length)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(n_layers)])

    def forward(self, hidden_states, attention_mask, output_all_encoded_layers=True, return_attention_prob=False,
                return_all_attention_prob=False, _rnds: Optional[List] = None, _last_rows=None):
        """attention_mask: the reference's dense additive tensor ([B,1,L,L] / [B,1,1,L]) or an
        `ops.StructuredMask` (same values, derived in-kernel).  `_last_rows` ([B, R] positions, only with
        output_all_encoded_layers=False): the last layer returns just those positions, [B, R, H]."""
        if _last_rows is not None and output_all_encoded_layers:
            raise ValueError("_last_rows needs output_all_encoded_layers=False")
        all_encoder_layers = []
        attacked_hidden_states = None
        calibrated_hidden_states = None
        all_attack_masks = []
        all_attention_prob = [] if return_attention_prob else None
        all_probs = [] if return_all_attention_prob else None
        for layer_idx, layer_module in enumerate(self.layer):
            rnd = _rnds[layer_idx] if _rnds is not None else None
            # only the calibrated output feeds the next layer (layers.py:1112): with output_all_encoded_layers=False the
            # attacked tail (dense, LayerNorm, feed-forward) of every layer but the last is unobservable and skipped
            need_attacked = output_all_encoded_layers or layer_idx == len(self.layer) - 1
            outs = layer_module(hidden_states, attention_mask, return_attention_prob, return_all_attention_prob,
                                _rnd=rnd, _need_attacked=need_attacked,
                                # the first layer owes no input gradient to an attack transform
                                _attack_upstream=layer_idx > 0 or not state_of(self).prune_dead_work,
                                _rows=_last_rows if layer_idx == len(self.layer) - 1 else None)
            attacked_hidden_states, calibrated_hidden_states, attack_mask, combined_attention_prob = outs[:4]
            hidden_states = calibrated_hidden_states  # layers.py:1112
            all_attack_masks.append(attack_mask)
            if output_all_encoded_layers:
                all_encoder_layers.append((attacked_hidden_states, calibrated_hidden_states))
            if return_attention_prob:
                all_attention_prob.append(combined_attention_prob)
            if return_all_attention_prob:
                all_probs.append(outs[4])
        if not output_all_encoded_layers:
            all_encoder_layers.append((attacked_hidden_states, calibrated_hidden_states))
        if return_all_attention_prob:
            return all_encoder_layers, all_attack_masks, all_probs
        if return_attention_prob:
            return all_encoder_layers, all_attack_masks, all_attention_prob
        return all_encoder_layers, all_attack_masks
