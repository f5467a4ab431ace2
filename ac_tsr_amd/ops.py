"""Host side of the fused calibrated-attention operator: tensors in, C-ABI call, tensors out.

`calibrated_attention` is the autograd-aware operator the layer modules call.  It launches
`acattn_calibrated_attention_fwd` / `_bwd` (include/acattn.h) on torch's current HIP stream with raw
device pointers; PyTorch only owns the memory.  Inputs must be CUDA(HIP) fp32 tensors -- there is no
CPU or eager fallback: a CPU tensor or a missing libacattn.so raises.

Reference semantics: recbole/model/layers.py:657-742 (AttackRMultiHeadAttention.cal_attack_mask /
cal_origin_qkv), :883-936 (AttackRTransformerLayer.combine_attention / forward), :677-680.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .state import DEFAULT as _DEFAULT_STATE
from ._lib import BwdIO, FwdOut, Problem


@dataclass
class StructuredMask:
    """The attention mask of SequentialRecommender.get_attention_mask (abstract_recommender.py:136-143)
    in factored form: which keys are real items + whether attention is causal.  The kernel derives the
    additive 0 / -10000 values from it instead of reading a [B,1,L,L] tensor from HBM."""

    key_valid: torch.Tensor  # [B, L] uint8, item_seq != 0
    causal: bool = True

    def dense(self) -> torch.Tensor:
        valid = self.key_valid.bool()[:, None, None, :]
        if self.causal:
            L = self.key_valid.shape[-1]
            valid = torch.tril(valid.expand(-1, -1, L, -1))
        return torch.where(valid, 0.0, -10000.0)


@dataclass
class AttentionConfig:
    n_heads: int
    combine_option: str = "gate"  # layers.py:883-896
    two_level: bool = True  # layers.py:911-914
    rich_calibrated_combine: str = "fixed"  # layers.py:929-936 (only read when two_level is False)
    adversarial: bool = True  # False = spatial calibrator only (BASELINE config 2)
    anneal_rate: float = 1.0


@dataclass
class ExplicitRandomness:
    """Parity mode: the layer's draws as tensors ([B,h,L,L]); keep masks None = no dropout."""

    noise: Optional[torch.Tensor] = None
    keep_after: Optional[torch.Tensor] = None
    keep_mask: Optional[torch.Tensor] = None
    keep_before: Optional[torch.Tensor] = None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_cuda(name: str, t: torch.Tensor, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.AcattnError(
            f"{name} lives on {t.device}: the calibrated-attention core runs only as HIP kernels on an MI355X "
            "(no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fill_problem(q, k, v, qa, ka, gate_logits, mask, w_order, b_order, w_dist, b_dist, scalar, rich_ratio,
                  cfg: AttentionConfig, p_drop: float, rnd, seed: int, keepalive: list, seed_tensor=None,
                  gate_is_prob: bool = False, affine=None) -> Problem:
    B, L, H = q.shape
    prob = Problem()
    prob.B, prob.L, prob.H, prob.n_heads = B, L, H, cfg.n_heads
    for name, t in (("q", q), ("k", k), ("v", v)):
        _need_cuda(name, t)
        assert t.shape == (B, L, H), f"{name} must be [B,L,H]"
    prob.q, prob.k, prob.v = _ptr(q), _ptr(k), _ptr(v)
    prob.adversarial = int(cfg.adversarial)
    if cfg.adversarial:
        for name, t in (("qa", qa), ("ka", ka)):
            _need_cuda(name, t)
            assert t.shape == (B, L, H)
        prob.qa, prob.ka = _ptr(qa), _ptr(ka)
    if cfg.combine_option not in _lib.COMBINE:
        raise KeyError(cfg.combine_option)  # layers.py:894-895
    prob.combine_option = _lib.COMBINE[cfg.combine_option]
    if cfg.adversarial and cfg.combine_option == "gate":
        _need_cuda("gate_logits", gate_logits)
        if gate_logits.shape != (B, L, L):
            # same failure as the reference's broadcast at layers.py:888 when seq_length != L
            raise RuntimeError(f"The size of tensor a ({gate_logits.shape[-1]}) must match the size of tensor b ({L})")
        prob.gate_logits = _ptr(gate_logits)
        prob.gate_is_prob = int(bool(gate_is_prob))
    if affine is not None:
        _need_cuda("affine", affine)
        assert affine.shape == (B, cfg.n_heads, 4, 16 * ((L + 15) // 16)), "affine must be [B, n_heads, 4, 16*ceil(L/16)]"
        prob.affine = _ptr(affine)
    # mask
    if isinstance(mask, StructuredMask):
        kv = mask.key_valid
        _need_cuda("key_valid", kv, torch.uint8)
        assert kv.shape == (B, L)
        prob.mask_mode, prob.causal, prob.key_valid = _lib.MASK_STRUCTURED, int(mask.causal), _ptr(kv)
    else:
        _need_cuda("attention_mask", mask)
        if mask.shape == (B, 1, L, L):
            prob.mask_mode = _lib.MASK_DENSE_LL
        elif mask.shape == (B, 1, 1, L):
            prob.mask_mode = _lib.MASK_DENSE_L
        else:
            raise ValueError(f"attention_mask must be [B,1,L,L] or [B,1,1,L], got {tuple(mask.shape)}")
        prob.mask = _ptr(mask)
    # spatial calibrator
    if w_order is not None:
        for t in (w_order, b_order):
            _need_cuda("order_affine", t)
        prob.w_order, prob.b_order = _ptr(w_order), _ptr(b_order)
    if w_dist is not None:
        for t in (w_dist, b_dist, scalar):
            _need_cuda("distance_affine", t)
        prob.w_dist, prob.b_dist, prob.scalar = _ptr(w_dist), _ptr(b_dist), _ptr(scalar)
    prob.anneal_rate = float(cfg.anneal_rate)
    prob.two_level = int(cfg.two_level)
    if not cfg.two_level:
        if cfg.rich_calibrated_combine not in ("fixed", "trainable"):
            raise KeyError(cfg.rich_calibrated_combine)  # layers.py:935-936
        prob.rich_combine = _lib.RICH[cfg.rich_calibrated_combine]
        if cfg.rich_calibrated_combine == "trainable":
            _need_cuda("rich_ratio", rich_ratio)
            prob.rich_ratio = _ptr(rich_ratio)
    # randomness
    prob.p_drop = float(p_drop)
    if rnd is not None:
        prob.rng_mode = _lib.RNG_EXPLICIT
        shape = (B, cfg.n_heads, L, L)
        if rnd.noise is not None:
            _need_cuda("noise", rnd.noise)
            assert rnd.noise.shape == shape
            prob.noise = _ptr(rnd.noise)
        for field in ("keep_after", "keep_mask", "keep_before"):
            t = getattr(rnd, field)
            if t is not None:
                if t.dtype != torch.uint8:
                    t = t.to(torch.uint8)
                    keepalive.append(t)
                _need_cuda(field, t, torch.uint8)
                assert t.shape == shape
                setattr(prob, field, _ptr(t))
        if p_drop > 0 and rnd.keep_after is None:
            prob.p_drop = 0.0
    else:
        prob.rng_mode = _lib.RNG_COUNTER
        prob.seed = seed & 0xFFFFFFFFFFFFFFFF
        if seed_tensor is not None:
            _need_cuda("seed_tensor", seed_tensor, torch.int64)
            prob.seed_device = _ptr(seed_tensor)
    return prob


# Route the launches of the training form (structured mask, counter RNG, `gate`, two_level, both spatial terms, no
# probability dumps) through the dispatcher operators of dispatch.py (torch.ops.acattn.*); everything else, and every call
# when this is False, goes to the C ABI directly.  Same kernels either way.
USE_DISPATCHER = True
# The attention node also returns, as its last output, acattn_mask_penalty_rows of its attack mask: sum (1 - M)^2 per
# (sequence, head, query block).  A loss that takes the mask penalty || 1 - M ||_2 (acsasrec.py:131-137) from these sums
# sends back a [B, nh, ceil(L/16)] cotangent instead of a dense [B, nh, L, L] one, and the backward kernels form
# d M = 2 d_pen (M - 1) from the tile they rebuild (acattn_bwd_io.d_penalty_part).  The mask tensor carries the sums as
# `M._acattn_pen` (ce.attacked_loss looks for it); False = the node returns None there.
PENALTY_ROWS = True


def _dispatcher_form(cfg, mask, rnd, want_probs, w_order, w_dist) -> bool:
    return (USE_DISPATCHER and isinstance(mask, StructuredMask) and rnd is None and not want_probs and cfg.two_level
            and w_order is not None and w_dist is not None and (not cfg.adversarial or cfg.combine_option == "gate"))


class _CalibratedAttention(torch.autograd.Function):
    """Inputs that can carry gradients are the first 12 positional tensors; the rest is configuration."""

    @staticmethod
    def forward(ctx, q, k, v, qa, ka, gate_logits, w_order, b_order, w_dist, b_dist, scalar, rich_ratio, mask,
                cfg: AttentionConfig, p_drop: float, rnd, seed: int, want_probs: bool, seed_tensor=None,
                read_rows=None, attack_upstream=True, state=_DEFAULT_STATE, gate_is_prob=False, affine=None):
        lib = _lib.load()
        B, L, H = q.shape
        ctx.attack_upstream = attack_upstream
        ctx.state = state
        ctx.set_materialize_grads(False)  # an output nobody differentiated arrives as None, not as a zero tensor
        ctx.active_qblocks = None
        ctx.read_rows = None
        if read_rows is not None:
            # bit q of entry b: some read position of sequence b lies in query block q (16 rows per block)
            if read_rows.shape[1] <= 4:
                ctx.read_rows = read_rows.contiguous()  # the kernels derive the block bits from the positions themselves
            else:
                blocks = torch.zeros(B, (L + 15) // 16, dtype=torch.int32, device=q.device).scatter_(
                    1, (read_rows >> 4), 1)
                weights = (1 << torch.arange(blocks.shape[1], dtype=torch.int32, device=q.device))
                ctx.active_qblocks = (blocks * weights).sum(dim=1, dtype=torch.int32)
        nh = cfg.n_heads
        keep = []
        wo = w_order.reshape(-1) if w_order is not None else None
        wd = w_dist.reshape(-1) if w_dist is not None else None
        ctx.gate_is_prob = gate_is_prob  # the backward reads the same tensor the same way; the planes are forward-only
        ctx.via_dispatcher = _dispatcher_form(cfg, mask, rnd, want_probs, w_order, w_dist)
        if ctx.via_dispatcher:
            from . import dispatch  # noqa: F401  (registers torch.ops.acattn.*)
            for name, t in (("q", q), ("k", k), ("v", v)):
                _need_cuda(name, t)
            if cfg.adversarial and gate_logits.shape != (B, L, L):
                # same failure as the reference's broadcast at layers.py:888 when seq_length != L
                raise RuntimeError(f"The size of tensor a ({gate_logits.shape[-1]}) must match the size of tensor b ({L})")
            want_pen = bool(PENALTY_ROWS and cfg.adversarial and any(ctx.needs_input_grad[:12]))
            ctx_att, ctx_cal, M, stats, pen = torch.ops.acattn.calibrated_attention_fwd(
                q, k, v, qa if cfg.adversarial else None, ka if cfg.adversarial else None,
                gate_logits if cfg.adversarial else None, mask.key_valid, bool(mask.causal), wo.contiguous(), b_order,
                wd.contiguous(), b_dist, scalar, nh, float(p_drop), int(seed) & 0x7FFFFFFFFFFFFFFF, seed_tensor,
                bool(gate_is_prob), affine, bool(cfg.adversarial), want_pen)
            if not cfg.adversarial:
                ctx_att = M = stats = pen = None
            if not want_pen:
                pen = None
            ctx.cfg, ctx.p_drop, ctx.rnd, ctx.seed, ctx.mask, ctx.seed_tensor = cfg, p_drop, rnd, seed, mask, seed_tensor
            ctx.save_for_backward(q, k, v, qa, ka, gate_logits, w_order, b_order, w_dist, b_dist, scalar, rich_ratio, M, stats)
            return (ctx_att, ctx_cal, M, None, None, None, None, pen)
        prob = _fill_problem(q, k, v, qa, ka, gate_logits, mask, wo, b_order, wd, b_dist, scalar, rich_ratio, cfg,
                             p_drop, rnd, seed, keep, seed_tensor, gate_is_prob, affine)
        out = FwdOut()
        ctx_cal = torch.empty_like(q)
        out.ctx_calibrated = _ptr(ctx_cal)
        ctx_att = M = stats = None
        probs = {}
        if cfg.adversarial:
            ctx_att = torch.empty_like(q)
            M = torch.empty(B, nh, L, L, device=q.device, dtype=torch.float32)
            stats = torch.empty(B, nh, L, _lib.NSTAT, device=q.device, dtype=torch.float32)
            out.ctx_attacked, out.attack_mask, out.row_stats = _ptr(ctx_att), _ptr(M), _ptr(stats)
            if want_probs:
                for name in ("after_spatial", "before_spatial", "perturbed_attention", "calibrated_attention"):
                    probs[name] = torch.empty_like(M)
                    setattr(out, name, _ptr(probs[name]))
        pen = None
        if PENALTY_ROWS and cfg.adversarial and any(ctx.needs_input_grad[:12]):
            pen = torch.empty(B, nh, (L + 15) // 16, device=q.device, dtype=torch.float32)
            out.penalty_part = _ptr(pen)  # filled by the launch itself or by acattn_mask_penalty_rows behind it
        _lib.check(lib.acattn_calibrated_attention_fwd(C.byref(prob), C.byref(out), _stream()), "calibrated_attention_fwd")
        ctx.cfg, ctx.p_drop, ctx.rnd, ctx.seed, ctx.mask, ctx.seed_tensor = cfg, p_drop, rnd, seed, mask, seed_tensor
        ctx.save_for_backward(q, k, v, qa, ka, gate_logits, w_order, b_order, w_dist, b_dist, scalar, rich_ratio, M, stats)
        outs = [ctx_att, ctx_cal, M] + [probs.get(n) for n in
                                        ("after_spatial", "before_spatial", "perturbed_attention", "calibrated_attention")]
        ctx.mark_non_differentiable(*[t for t in outs[3:] if t is not None])
        return tuple(outs) + (pen,)

    @staticmethod
    def backward(ctx, d_att, d_cal, d_M, *_unused):
        # Never mutates saved state: the trainer walks this node twice (retain_graph=True,
        # recbole/trainer/trainer.py:677,684).
        (q, k, v, qa, ka, gate_logits, w_order, b_order, w_dist, b_dist, scalar, rich_ratio, M, stats) = ctx.saved_tensors
        cfg = ctx.cfg
        if not cfg.adversarial:
            raise _lib.AcattnError("backward of the spatial-only operator is not provided")
        lib = _lib.load()
        B, L, H = q.shape
        nh, dh = cfg.n_heads, H // cfg.n_heads
        keep = []
        wo = w_order.reshape(-1) if w_order is not None else None
        wd = w_dist.reshape(-1) if w_dist is not None else None
        d_att = None if d_att is None else d_att.contiguous()
        d_cal = None if d_cal is None else d_cal.contiguous()
        d_M = None if d_M is None else d_M.contiguous()
        d_pen = _unused[4].contiguous() if len(_unused) > 4 and _unused[4] is not None else None
        if ctx.via_dispatcher:
            attack_only = ctx.state.attack_pass_only and not ctx.attack_upstream
            dq, dk, dv, dqa, dka, dgate_part, part = torch.ops.acattn.calibrated_attention_bwd(
                q, k, v, qa, ka, gate_logits, ctx.mask.key_valid, bool(ctx.mask.causal), wo.contiguous(), b_order,
                wd.contiguous(), b_dist, scalar, nh, float(ctx.p_drop), int(ctx.seed) & 0x7FFFFFFFFFFFFFFF, ctx.seed_tensor,
                bool(ctx.gate_is_prob), M, stats, d_att, d_cal, d_M, ctx.read_rows, ctx.active_qblocks, bool(attack_only), d_pen)
            return _CalibratedAttention._finish_backward(lib, attack_only, dq, dk, dv, dqa, dka, dgate_part, part, dh,
                                                         w_order, b_order, w_dist, b_dist, scalar, rich_ratio, ctx.state)
        prob = _fill_problem(q, k, v, qa, ka, gate_logits, ctx.mask, wo, b_order, wd, b_dist, scalar, rich_ratio, cfg,
                             ctx.p_drop, ctx.rnd, ctx.seed, keep, ctx.seed_tensor, ctx.gate_is_prob)
        io = BwdIO()
        io.attack_mask, io.row_stats = _ptr(M), _ptr(stats)
        io.d_ctx_attacked, io.d_ctx_calibrated, io.d_attack_mask = _ptr(d_att), _ptr(d_cal), _ptr(d_M)
        io.d_penalty_part = _ptr(d_pen)
        dq, dk, dv, dqa, dka = (torch.empty_like(q) for _ in range(5))
        io.dq, io.dk, io.dv, io.dqa, io.dka = _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dqa), _ptr(dka)
        dgate = dgate_part = None
        if cfg.combine_option == "gate":
            io.dgate_logits = _ptr(q)  # (placeholder for the query below: only tested for NULL)
        # the three per-(b, head) partial sums share ONE [B*nh, 4*dh + 4] buffer, reduced in a single pass
        width = 4 * dh + 4
        part = torch.empty(B * nh, width, device=q.device, dtype=torch.float32)
        ws_bytes = int(lib.acattn_calibrated_attention_bwd_workspace_bytes(C.byref(prob)))
        ws = torch.empty(max(ws_bytes, 4) // 4, device=q.device, dtype=torch.float32)  # row scalars (streaming backward)
        io.workspace = _ptr(ws)
        base = part.data_ptr()
        io.dw_order_part, io.dw_dist_part, io.dsmall_part = base, base + 4 * 2 * dh, base + 4 * 4 * dh
        io.part_stride = width
        # the context cotangents are zero outside the read positions (the caller's promise, `read_rows`)
        # (with a mask cotangent every block stays active, but those without a read position only owe the mask path)
        io.active_qblocks = _ptr(ctx.active_qblocks) if ctx.active_qblocks is not None else None
        if ctx.read_rows is not None:
            io.read_rows, io.n_read_rows = _ptr(ctx.read_rows), ctx.read_rows.shape[1]
        # pass 2 through a layer with nothing attack-related upstream: only the attack transforms' inputs matter
        attack_only = ctx.state.attack_pass_only and not ctx.attack_upstream
        io.attack_only = int(attack_only)
        if cfg.combine_option == "gate":
            summed = bool(lib.acattn_calibrated_attention_bwd_gate_summed(C.byref(prob), C.byref(io)))  # see dispatch._bwd_cuda
            dgate_part = torch.empty(B, 1 if summed else nh, L, L, device=q.device, dtype=torch.float32)
            io.dgate_logits, io.dgate_summed = _ptr(dgate_part), int(summed)
        _lib.check(lib.acattn_calibrated_attention_bwd(C.byref(prob), C.byref(io), _stream()), "calibrated_attention_bwd")
        return _CalibratedAttention._finish_backward(lib, attack_only, dq, dk, dv, dqa, dka, dgate_part, part, dh, w_order,
                                                     b_order, w_dist, b_dist, scalar, rich_ratio, ctx.state)

    @staticmethod
    def backward_pair(ctx, g_cal, g_att):
        """Both cotangent sets of the single-pass combined backward (combined.py) in ONE launch pair, or None when this
        node's situation is not the one the kernels share the recomputation for (include/acattn.h: acattn_bwd_io.dqa2): the
        calibrated set without an attacked-context / mask cotangent, the attacked set in the form it has in a layer with
        no attack transform upstream (d ctx_calibrated and / or the penalty's row sums; only dqa, dka wanted), L > 64."""
        cfg = ctx.cfg
        pick = lambda g, i: g[i] if len(g) > i else None
        d_att1, d_cal1, d_M1, d_pen1 = pick(g_cal, 0), pick(g_cal, 1), pick(g_cal, 2), pick(g_cal, 7)
        d_att2, d_cal2, d_M2, d_pen2 = pick(g_att, 0), pick(g_att, 1), pick(g_att, 2), pick(g_att, 7)
        if not (cfg.adversarial and getattr(ctx, "via_dispatcher", False) and ctx.state.prune_dead_work and not ctx.attack_upstream
                and d_att1 is None and d_att2 is None and d_M1 is None and d_M2 is None and d_pen1 is None
                and d_cal1 is not None and (d_cal2 is not None or d_pen2 is not None)
                and ctx.read_rows is None and ctx.active_qblocks is None):
            return None
        (q, k, v, qa, ka, gate_logits, w_order, b_order, w_dist, b_dist, scalar, rich_ratio, M, stats) = ctx.saved_tensors
        from . import dispatch
        lib = _lib.load()
        B, L, H = q.shape
        nh, dh = cfg.n_heads, H // cfg.n_heads
        wo, wd = w_order.reshape(-1).contiguous(), w_dist.reshape(-1).contiguous()
        prob = dispatch._problem(q, k, v, qa, ka, gate_logits, ctx.mask.key_valid, bool(ctx.mask.causal), wo, b_order, wd, b_dist,
                                 scalar, nh, float(ctx.p_drop), int(ctx.seed) & 0x7FFFFFFFFFFFFFFF, ctx.seed_tensor,
                                 bool(ctx.gate_is_prob), None, True)
        io = BwdIO()
        d_cal1 = d_cal1.contiguous()
        d_cal2 = None if d_cal2 is None else d_cal2.contiguous()
        d_pen2 = None if d_pen2 is None else d_pen2.contiguous()
        io.attack_mask, io.row_stats = _ptr(M), _ptr(stats)
        io.d_ctx_calibrated, io.d_ctx_calibrated2, io.d_penalty_part2 = _ptr(d_cal1), _ptr(d_cal2), _ptr(d_pen2)
        dq, dk, dv, dqa, dka, dqa2, dka2 = (torch.empty_like(q) for _ in range(7))
        io.dq, io.dk, io.dv, io.dqa, io.dka, io.dqa2, io.dka2 = (_ptr(t) for t in (dq, dk, dv, dqa, dka, dqa2, dka2))
        dgate_part = torch.empty(B, nh, L, L, device=q.device, dtype=torch.float32)
        io.dgate_logits = _ptr(dgate_part)
        width = 4 * dh + 4
        part = torch.empty(B * nh, width, device=q.device, dtype=torch.float32)
        ws_bytes = int(lib.acattn_calibrated_attention_bwd_workspace_bytes(C.byref(prob)))
        ws = torch.empty(max(ws_bytes, 4) // 4, device=q.device, dtype=torch.float32)
        io.workspace = _ptr(ws)
        base = part.data_ptr()
        io.dw_order_part, io.dw_dist_part, io.dsmall_part = base, base + 4 * 2 * dh, base + 4 * 4 * dh
        io.part_stride = width
        if not lib.acattn_calibrated_attention_bwd_pair_supported(C.byref(prob), C.byref(io)):
            return None  # (L <= 64: the row-resident kernel; head size 128; a pinned kernel)
        _lib.check(lib.acattn_calibrated_attention_bwd(C.byref(prob), C.byref(io), _stream()), "calibrated_attention_bwd (pair)")
        res_cal = _CalibratedAttention._finish_backward(lib, False, dq, dk, dv, dqa, dka, dgate_part, part, dh, w_order, b_order,
                                                        w_dist, b_dist, scalar, rich_ratio)
        res_att = (None, None, None, dqa2, dka2) + (None,) * 19
        return res_cal, res_att

    @staticmethod
    def _finish_backward(lib, attack_only, dq, dk, dv, dqa, dka, dgate_part, part, dh, w_order, b_order, w_dist, b_dist,
                         scalar, rich_ratio, state=None):
        """The reductions behind the backward launch: per-head gate partials and per-(b, head) parameter partials."""
        dgate = None
        if attack_only:
            return (None, None, None, dqa, dka) + (None,) * 19
        if dgate_part is not None and dgate_part.shape[1] > 1 and part.shape[0] < 4096:
            # the gate is shared by the heads (layers.py:887 unsqueeze(1)): its per-head gradients and the parameter
            # partials are summed by ONE launch
            Bq, nhq, Lq = dgate_part.shape[0], dgate_part.shape[1], dgate_part.shape[2]
            dgate = torch.empty(Bq, Lq, dgate_part.shape[3], device=part.device, dtype=torch.float32)
            tot = torch.empty(part.shape[1], device=part.device, dtype=torch.float32)
            _lib.check(lib.acattn_sum_rows_pair(_ptr(dgate_part), _ptr(dgate), Bq, nhq, Lq * dgate_part.shape[3],
                                                _ptr(part), _ptr(tot), 1, part.shape[0], part.shape[1], _stream()),
                       "sum_rows_pair")
        else:
            if dgate_part is not None:  # (one head, or the one-row form's head-summed tensor: nothing to add)
                dgate = dgate_part[:, 0] if dgate_part.shape[1] == 1 else sum_rows(dgate_part, 1)
            if state is not None and state.attack_pass_only:
                # pass 2 keeps only the attack transforms' gradients (trainer.py:678-684): the calibrators' are dropped by
                # autograd on arrival, so they are not summed at all
                return (dq, dk, dv, dqa, dka, dgate) + (None,) * 18
            tot = sum_rows0(part, state)
        small = tot[4 * dh:]
        g_wo = tot[:2 * dh].view_as(w_order) if w_order is not None else None
        g_bo = small[0:1].view_as(b_order) if w_order is not None else None
        g_wd = tot[2 * dh:4 * dh].view_as(w_dist) if w_dist is not None else None
        g_bd = small[1:2].view_as(b_dist) if w_dist is not None else None
        g_sc = small[2:3].view_as(scalar) if w_dist is not None else None
        g_rr = small[3:4].view_as(rich_ratio) if rich_ratio is not None else None
        if state is not None:
            state.watch(tot, g_wo, g_bo, g_wd, g_bd, g_sc, g_rr)
        return (dq, dk, dv, dqa, dka, dgate, g_wo, g_bo, g_wd, g_bd, g_sc, g_rr, None, None, None, None, None, None, None,
                None, None, None, None, None)


def calibrated_attention(q, k, v, qa, ka, gate_logits, mask, cfg: AttentionConfig, *, w_order=None, b_order=None,
                         w_dist=None, b_dist=None, scalar=None, rich_ratio=None, p_drop: float = 0.0,
                         rnd: Optional[ExplicitRandomness] = None, seed: Optional[int] = None,
                         want_probs: bool = False, seed_tensor: Optional[torch.Tensor] = None,
                         read_rows: Optional[torch.Tensor] = None, attack_upstream: bool = True,
                         state=_DEFAULT_STATE, gate_is_prob: bool = False, affine: Optional[torch.Tensor] = None):
    """Fused core of one AttackRTransformerLayer between the projections and the output dense.

    `attack_upstream=False` declares that nothing that produced q, k, v holds attack transforms (first encoder layer):
    in pass 2 of the two-pass trainer the backward then returns only the gradients of qa and ka.

    `state` (state.StepState) tells the backward which pass of the two-pass trainer is running.

    `read_rows` ([B, R] int64, optional) promises that the two context outputs are only ever read at those positions
    of each sequence: the backward then skips query blocks that cannot carry a cotangent (a speed hint; results are
    the same because the skipped rows' cotangents are zero).

    `gate_is_prob` / `affine`: what the producer of q, k and the gate already computed (acattn_problem.gate_is_prob,
    .affine; linear.projections returns both as `extras`): `gate_logits` then holds sigmoid(gate(mixed_query)) -- the
    gradient returned for it is still that of the logits -- and `affine` [B, n_heads, 4, 16*ceil(L/16)] the rank-1
    halves of the spatial calibrator's affines.

    Returns (ctx_attacked [B,L,H] | None, ctx_calibrated [B,L,H], M [B,h,L,L] | None, probs dict).
    """
    if rnd is None and seed is None:
        seed = state.draw_seed()  # one 63-bit seed per call, salted per data-parallel rank
    # seed_tensor (device int64[1]) is added to `seed` inside the kernels: under hipGraph capture `seed` is frozen
    # into the graph, the tensor is what changes between replays (trainer.enable_graph)
    outs = _CalibratedAttention.apply(q, k, v, qa, ka, gate_logits, w_order, b_order, w_dist, b_dist, scalar,
                                      rich_ratio, mask, cfg, p_drop, rnd, seed or 0, want_probs, seed_tensor, read_rows,
                                      attack_upstream, state, gate_is_prob, affine)
    names = ("after_spatial", "before_spatial", "perturbed_attention", "calibrated_attention")
    probs = {n: t for n, t in zip(names, outs[3:7]) if t is not None}
    if outs[7] is not None:
        outs[2]._acattn_pen = outs[7]  # see PENALTY_ROWS
    return outs[0], outs[1], outs[2], probs


def materialize_randomness(B: int, n_heads: int, L: int, seed: int, p_drop: float, device) -> ExplicitRandomness:
    """The counter-mode draws for (seed, shape) as tensors (acattn_rng_materialize)."""
    lib = _lib.load()
    shape = (B, n_heads, L, L)
    noise = torch.empty(shape, device=device, dtype=torch.float32)
    ka, km, kb = (torch.empty(shape, device=device, dtype=torch.uint8) for _ in range(3))
    _lib.check(lib.acattn_rng_materialize(B, n_heads, L, seed, p_drop, _ptr(noise), _ptr(ka), _ptr(km), _ptr(kb),
                                          _stream()), "rng_materialize")
    return ExplicitRandomness(noise=noise, keep_after=ka, keep_mask=km, keep_before=kb)


def fwd_algorithmic_bytes(B, L, H, n_heads, adversarial=True, combine_option="gate") -> int:
    prob = Problem()
    prob.B, prob.L, prob.H, prob.n_heads = B, L, H, n_heads
    prob.adversarial, prob.combine_option = int(adversarial), _lib.COMBINE[combine_option]
    return int(_lib.load().acattn_fwd_algorithmic_bytes(C.byref(prob)))


def _launch_sum_rows(x, batch, R, Cn, out_shape):
    out = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().acattn_sum_rows(_ptr(x), _ptr(out), batch, R, Cn, _stream()), "sum_rows")
    return out


def sum_rows(x: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """x.sum(dim) for a contiguous fp32 HIP tensor through acattn_sum_rows (dims before `dim` form the batch, dims
    after it the columns).  A long reduction with few output columns (a bias gradient: 25,600 rows x 64) is done in
    two stages, first into `s` partial rows per batch, so that the first stage fills the chip.  No autograd: used
    inside backward functions."""
    if not x.is_cuda or x.dtype != torch.float32 or x.numel() == 0:
        return x.sum(dim)
    x = x.contiguous()
    dim = dim % x.dim()
    batch = 1
    for s_ in x.shape[:dim]:
        batch *= s_
    R = x.shape[dim]
    out_shape = tuple(x.shape[:dim]) + tuple(x.shape[dim + 1:])
    Cn = x.numel() // (batch * R)
    wgs = batch * ((Cn + 1023) // 1024)
    # (medium reductions with few columns are spread over ~32 workgroups by the launcher itself)
    if wgs < 128 and R >= 4096:
        s = 1
        for cand in range(2, 257):
            if R % cand == 0 and R // cand >= 8:
                s = cand
        if s > 1:
            part = _launch_sum_rows(x, batch * s, R // s, Cn, (batch, s, Cn))
            return _launch_sum_rows(part, batch, s, Cn, out_shape)
    return _launch_sum_rows(x, batch, R, Cn, out_shape)


def sum_rows0(x: torch.Tensor, state=None) -> torch.Tensor:
    """sum_rows(x, 0) of a parameter-partials tensor.  Inside a trainer's backward walk (`state.deferring()`) the sum is
    left to the walk's one reduction launch: the tensor comes back unwritten and the caller names the views of it that it
    returns as gradients with state.watch(...) (state.py)."""
    if state is not None and state.deferring() and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.shape[0] > 1:
        x = x.contiguous()
        out = torch.empty(x.shape[1:], device=x.device, dtype=torch.float32)
        state.defer_sum(x, out)
        return out
    return sum_rows(x, 0)


def linear_wgrad(x2: torch.Tensor, g2: torch.Tensor, want_bias: bool, state=None):
    """(dW [N,K], db [N] or None) of y = x W^T + b from x2 [M,K] and dy g2 [M,N] through acattn_linear_wgrad
    (one pass over both matrices, fp32 MFMA).  No autograd: used inside backward functions."""
    if state is not None and state.deferring():
        return linear_wgrad_grouped([(x2, g2, want_bias)], state)[0]
    assert x2.is_cuda and x2.dtype == torch.float32 and g2.dtype == torch.float32 and x2.shape[0] == g2.shape[0]
    x2, g2 = x2.contiguous(), g2.contiguous()
    M, K = x2.shape
    N = g2.shape[1]
    lib = _lib.load()
    ws = torch.empty(lib.acattn_linear_wgrad_workspace_bytes(M, K, N) // 4, device=x2.device, dtype=torch.float32)
    dw = torch.empty(N, K, device=x2.device, dtype=torch.float32)
    db = torch.empty(N, device=x2.device, dtype=torch.float32) if want_bias else None
    _lib.check(lib.acattn_linear_wgrad(_ptr(x2), _ptr(g2), M, K, N, _ptr(ws), _ptr(dw), _ptr(db) if want_bias else None,
                                       _stream()), "linear_wgrad")
    return dw, db


class _MaskPenalty(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m):
        _need_cuda("attack mask", m)
        m = m.contiguous()
        ws = torch.empty(_lib.PENALTY_WS_FLOATS, device=m.device, dtype=torch.float32)
        norm = torch.empty((), device=m.device, dtype=torch.float32)
        _lib.check(_lib.load().acattn_mask_penalty_fwd(_ptr(m), m.numel(), _ptr(ws), _ptr(norm), _stream()), "mask_penalty_fwd")
        ctx.save_for_backward(m, norm)
        return norm

    @staticmethod
    def backward(ctx, d_norm):
        m, norm = ctx.saved_tensors
        d_m = torch.empty_like(m)
        _lib.check(_lib.load().acattn_mask_penalty_bwd(_ptr(m), _ptr(norm), _ptr(d_norm.contiguous()), m.numel(), _ptr(d_m),
                                                       _stream()), "mask_penalty_bwd")
        return d_m


def mask_penalty(attack_mask: torch.Tensor) -> torch.Tensor:
    """torch.norm(1 - attack_mask, p=2) (acsasrec.py:131-137) as one pass over the mask each way."""
    return _MaskPenalty.apply(attack_mask)


def linear_wgrad_grouped(items, state=None):
    """[(dW, db or None), ...] for items = [(x2 [M,K_i], g2 [M,N_i], want_bias), ...]: items that share M go through
    ONE acattn_linear_wgrad_grouped launch pair (at most WGRAD_MAX_GROUP per launch).  With a `state` that is
    `deferring()` (a trainer's backward walk) only stage 1 is launched: the returned tensors are written by the
    trainer's StepState.flush_deferred() at the end of the walk (state.py)."""
    lib = _lib.load()
    defer = state is not None and state.deferring()
    out = [None] * len(items)
    buckets = {}
    for pos, (x, g, wb) in enumerate(items):
        assert x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32 and x.shape[0] == g.shape[0]
        buckets.setdefault(x.shape[0], []).append((pos, x.contiguous(), g.contiguous(), wb))
    for M, members in buckets.items():
        for s0 in range(0, len(members), _lib.WGRAD_MAX_GROUP):
            chunk = members[s0:s0 + _lib.WGRAD_MAX_GROUP]
            n = len(chunk)
            dev = chunk[0][1].device
            ws_bytes = sum(lib.acattn_linear_wgrad_workspace_bytes(M, x.shape[1], g.shape[1]) for _, x, g, _ in chunk)
            ws = torch.empty(ws_bytes // 4, device=dev, dtype=torch.float32)
            dws = [torch.empty(g.shape[1], x.shape[1], device=dev, dtype=torch.float32) for _, x, g, _ in chunk]
            dbs = [torch.empty(g.shape[1], device=dev, dtype=torch.float32) if wb else None for _, _, g, wb in chunk]
            arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
            ints = lambda vals: (C.c_int32 * n)(*vals)
            if defer:
                n_part = C.c_int32(0)
                w_off, b_off = (C.c_int64 * n)(), (C.c_int64 * n)()
                _lib.check(lib.acattn_linear_wgrad_grouped_partial(
                    arr([x.data_ptr() for _, x, _, _ in chunk]), arr([g.data_ptr() for _, _, g, _ in chunk]),
                    ints([x.shape[1] for _, x, _, _ in chunk]), ints([g.shape[1] for _, _, g, _ in chunk]),
                    ints([1 if wb else 0 for _, _, _, wb in chunk]), n, M, _ptr(ws), C.byref(n_part), w_off, b_off, _stream()),
                    "linear_wgrad_grouped_partial")
                base = ws.data_ptr()
                for k, ((pos, x, g, wb), gw, gb) in enumerate(zip(chunk, dws, dbs)):
                    # the queue keeps the memory alive (storages), not the tensors: autograd adopts a gradient tensor as
                    # a leaf's .grad without copying only while nobody else holds it
                    state.defer({"part_w": base + 4 * w_off[k], "part_b": base + 4 * b_off[k], "K": x.shape[1], "N": g.shape[1],
                                 "P": n_part.value, "dw": gw.data_ptr(), "db": None if gb is None else gb.data_ptr(),
                                 "keep": (ws.untyped_storage(), gw.untyped_storage(), None if gb is None else gb.untyped_storage())})
                    out[pos] = (gw, gb)
                continue
            _lib.check(lib.acattn_linear_wgrad_grouped(
                arr([x.data_ptr() for _, x, _, _ in chunk]), arr([g.data_ptr() for _, _, g, _ in chunk]),
                ints([x.shape[1] for _, x, _, _ in chunk]), ints([g.shape[1] for _, _, g, _ in chunk]),
                arr([t.data_ptr() for t in dws]), arr([t.data_ptr() if t is not None else None for t in dbs]), n, M,
                _ptr(ws), _stream()), "linear_wgrad_grouped")
            for (pos, _, _, _), gw, gb in zip(chunk, dws, dbs):
                out[pos] = (gw, gb)
    return out
