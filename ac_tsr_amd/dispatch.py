"""The fused core as torch dispatcher operators: `torch.ops.acattn.calibrated_attention_fwd / _bwd`.

SURVEY.md section 8(b) asks for the native side to sit behind a torch custom op.  The library itself stays a plain C ABI
(include/acattn.h: no torch types in it, any host can bind it); this module registers that ABI's two attention entry
points with the dispatcher (`torch.library`), in the form the training step uses -- structured mask, counter RNG, `gate`
combine, two_level -- so that

  * the launches are visible to the dispatcher (profiler op names, `torch.library.opcheck`, FakeTensor / meta tracing: each
    op has a Meta implementation that only computes shapes),
  * `calibrated_attention_fwd` carries an autograd formula of its own (`register_autograd`: a caller of the raw op gets
    gradients without ops._CalibratedAttention), and
  * ops._CalibratedAttention routes its launches through them when the call has that form (`ops.USE_DISPATCHER`), so the
    operators tested here are the operators that train.

The options outside that form (dense masks, explicit randomness for parity tests, 'fixed' / 'annealing', one-level,
probability dumps) keep the direct C-ABI call in ops.py.  Reference semantics: recbole/model/layers.py:657-742, 883-936,
677-680 (see ops.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

_LIB = torch.library.Library("acattn", "DEF")
_LIB.define(
    "calibrated_attention_fwd(Tensor q, Tensor k, Tensor v, Tensor? qa, Tensor? ka, Tensor? gate, Tensor key_valid, "
    "bool causal, Tensor w_order, Tensor b_order, Tensor w_dist, Tensor b_dist, Tensor scalar, int n_heads, "
    "float p_drop, int seed, Tensor? seed_tensor, bool gate_is_prob, Tensor? affine, bool adversarial, "
    "bool want_penalty=True) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_LIB.define(
    "calibrated_attention_bwd(Tensor q, Tensor k, Tensor v, Tensor qa, Tensor ka, Tensor gate, Tensor key_valid, "
    "bool causal, Tensor w_order, Tensor b_order, Tensor w_dist, Tensor b_dist, Tensor scalar, int n_heads, "
    "float p_drop, int seed, Tensor? seed_tensor, bool gate_is_prob, Tensor attack_mask, Tensor row_stats, "
    "Tensor? d_ctx_attacked, Tensor? d_ctx_calibrated, Tensor? d_attack_mask, Tensor? read_rows, Tensor? active_qblocks, "
    "bool attack_only, Tensor? d_penalty_part=None) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")


import os

# ACATTN_POISON_OUTPUTS=1 (tests): every output buffer of the two operators starts as NaN instead of uninitialised memory, so
# an output element a kernel forgets to write shows up instead of reading what an earlier launch left in a reused buffer
# (round 4 found the one-row backward at head size 128 writing half of dq's columns that way).
_POISON = os.environ.get("ACATTN_POISON_OUTPUTS") == "1"


def _out_like(t: torch.Tensor) -> torch.Tensor:
    return torch.full_like(t, float("nan")) if _POISON else torch.empty_like(t)


def _out(*shape, device) -> torch.Tensor:
    if _POISON:
        return torch.full(shape, float("nan"), device=device, dtype=torch.float32)
    return torch.empty(*shape, device=device, dtype=torch.float32)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _problem(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed,
             seed_tensor, gate_is_prob, affine, adversarial) -> _lib.Problem:
    B, L, H = q.shape
    p = _lib.Problem()
    p.B, p.L, p.H, p.n_heads = B, L, H, n_heads
    p.q, p.k, p.v = _ptr(q), _ptr(k), _ptr(v)
    p.adversarial = int(adversarial)
    if adversarial:
        p.qa, p.ka, p.gate_logits = _ptr(qa), _ptr(ka), _ptr(gate)
        p.gate_is_prob = int(gate_is_prob)
    p.combine_option, p.two_level = _lib.COMBINE["gate"], 1
    p.mask_mode, p.causal, p.key_valid = _lib.MASK_STRUCTURED, int(causal), _ptr(key_valid)
    p.w_order, p.b_order, p.w_dist, p.b_dist, p.scalar = (_ptr(t) for t in (w_order, b_order, w_dist, b_dist, scalar))
    p.affine = _ptr(affine)
    p.rng_mode, p.p_drop, p.seed = _lib.RNG_COUNTER, float(p_drop), seed & 0xFFFFFFFFFFFFFFFF
    p.seed_device = _ptr(seed_tensor)
    return p


def _check(name, t, shape, dtype, device, optional=False):
    """The operators hand raw device pointers to the C ABI, which reads them as contiguous fp32 / uint8 / int of exactly the
    documented shape (include/acattn.h): anything else -- an autocast bf16 tensor, an int64 validity mask, a gate of
    another sequence length, a tensor on another GPU -- would be reinterpreted and read out of bounds.  TypeError /
    ValueError here instead."""
    if t is None:
        if optional:
            return
        raise TypeError(f"acattn: `{name}` is required")
    if not t.is_cuda:
        raise _lib.AcattnError(f"acattn operators run only as HIP kernels on an MI355X (`{name}` is on {t.device}): no CPU fallback")
    if t.device != device:
        raise ValueError(f"acattn: `{name}` is on {t.device}, the other tensors on {device}")
    if t.dtype != dtype:
        raise TypeError(f"acattn: `{name}` must be {dtype} (got {t.dtype}): the kernels compute in fp32, outside autocast")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"acattn: `{name}` must have shape {tuple(shape)} (got {tuple(t.shape)})")
    if not t.is_contiguous():
        raise ValueError(f"acattn: `{name}` must be contiguous")


def _check_problem(q, k, v, qa, ka, gate, key_valid, w_order, b_order, w_dist, b_dist, scalar, n_heads, seed_tensor, affine,
                   adversarial):
    if q.dim() != 3:
        raise ValueError(f"acattn: `q` must be [B, L, H] (got {tuple(q.shape)})")
    B, L, H = q.shape
    if n_heads <= 0 or H % n_heads:
        raise ValueError(f"The hidden size ({H}) is not a multiple of the number of attention heads ({n_heads})")  # layers.py:618-622
    dh, dev, f32 = H // n_heads, q.device, torch.float32
    _check("q", q, (B, L, H), f32, dev)
    _check("k", k, (B, L, H), f32, dev)
    _check("v", v, (B, L, H), f32, dev)
    if adversarial:
        _check("qa", qa, (B, L, H), f32, dev)
        _check("ka", ka, (B, L, H), f32, dev)
        _check("gate", gate, (B, L, L), f32, dev)
    _check("key_valid", key_valid, (B, L), torch.uint8, dev)
    for name, t, n in (("w_order", w_order, 2 * dh), ("w_dist", w_dist, 2 * dh), ("b_order", b_order, 1), ("b_dist", b_dist, 1),
                       ("scalar", scalar, 1)):
        _check(name, t, None, f32, dev)
        if t.numel() != n:
            raise ValueError(f"acattn: `{name}` must hold {n} value(s) (got shape {tuple(t.shape)})")
    _check("seed_tensor", seed_tensor, (1,), torch.int64, dev, optional=True)
    _check("affine", affine, (B, n_heads, 4, 16 * ((L + 15) // 16)), f32, dev, optional=True)
    return B, L, H, dh


def _fwd_cuda(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed,
              seed_tensor, gate_is_prob, affine, adversarial, want_penalty=True):
    B, L, H, _ = _check_problem(q, k, v, qa, ka, gate, key_valid, w_order, b_order, w_dist, b_dist, scalar, n_heads, seed_tensor,
                                affine, adversarial)
    prob = _problem(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop,
                    seed, seed_tensor, gate_is_prob, affine, adversarial)
    out = _lib.FwdOut()
    ctx_cal = _out_like(q)
    out.ctx_calibrated = _ptr(ctx_cal)
    if adversarial:
        ctx_att = _out_like(q)
        M = _out(B, n_heads, L, L, device=q.device)
        stats = torch.empty(B, n_heads, L, _lib.NSTAT, device=q.device, dtype=torch.float32)  # (3 of its 8 columns are spare)
        out.ctx_attacked, out.attack_mask, out.row_stats = _ptr(ctx_att), _ptr(M), _ptr(stats)
        # sum (1 - M)^2 per (sequence, head, query block): the mask penalty without another pass over M (include/acattn.h).
        # Only when a gradient can flow (evaluation / no_grad: for L <= 64 it is one more launch behind the kernel)
        if want_penalty:
            pen = _out(B, n_heads, (L + 15) // 16, device=q.device)
            out.penalty_part = _ptr(pen)
        else:
            pen = q.new_empty(0)
    else:  # the spatial-only operator writes one context; the other outputs are empty
        ctx_att, M, stats, pen = (q.new_empty(0) for _ in range(4))
    _lib.check(_lib.load().acattn_calibrated_attention_fwd(C.byref(prob), C.byref(out), _stream()), "calibrated_attention_fwd")
    return ctx_att, ctx_cal, M, stats, pen


def _fwd_meta(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed,
              seed_tensor, gate_is_prob, affine, adversarial, want_penalty=True):
    B, L, H = q.shape
    if adversarial:
        return (torch.empty_like(q), torch.empty_like(q), q.new_empty(B, n_heads, L, L), q.new_empty(B, n_heads, L, _lib.NSTAT),
                q.new_empty(B, n_heads, (L + 15) // 16) if want_penalty else q.new_empty(0))
    return q.new_empty(0), torch.empty_like(q), q.new_empty(0), q.new_empty(0), q.new_empty(0)


def _bwd_cuda(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed,
              seed_tensor, gate_is_prob, attack_mask, row_stats, d_ctx_attacked, d_ctx_calibrated, d_attack_mask, read_rows,
              active_qblocks, attack_only, d_penalty_part=None):
    B, L, H, dh = _check_problem(q, k, v, qa, ka, gate, key_valid, w_order, b_order, w_dist, b_dist, scalar, n_heads, seed_tensor,
                                 None, True)
    dev, f32 = q.device, torch.float32
    _check("attack_mask", attack_mask, (B, n_heads, L, L), f32, dev)
    _check("row_stats", row_stats, (B, n_heads, L, _lib.NSTAT), f32, dev)
    _check("d_ctx_attacked", d_ctx_attacked, (B, L, H), f32, dev, optional=True)
    _check("d_ctx_calibrated", d_ctx_calibrated, (B, L, H), f32, dev, optional=True)
    _check("d_attack_mask", d_attack_mask, (B, n_heads, L, L), f32, dev, optional=True)
    _check("d_penalty_part", d_penalty_part, (B, n_heads, (L + 15) // 16), f32, dev, optional=True)
    _check("active_qblocks", active_qblocks, (B,), torch.int32, dev, optional=True)
    if read_rows is not None:
        if read_rows.dim() != 2 or read_rows.shape[0] != B:
            raise ValueError(f"acattn: `read_rows` must be [B, n] (got {tuple(read_rows.shape)})")
        _check("read_rows", read_rows, None, torch.int64, dev)
    lib = _lib.load()
    prob = _problem(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop,
                    seed, seed_tensor, gate_is_prob, None, True)
    io = _lib.BwdIO()
    io.attack_mask, io.row_stats = _ptr(attack_mask), _ptr(row_stats)
    io.d_ctx_attacked, io.d_ctx_calibrated, io.d_attack_mask = _ptr(d_ctx_attacked), _ptr(d_ctx_calibrated), _ptr(d_attack_mask)
    dq, dk, dv, dqa, dka = ((_out_like(q) if not attack_only or n in (3, 4) else torch.empty_like(q)) for n in range(5))
    io.dq, io.dk, io.dv, io.dqa, io.dka = _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dqa), _ptr(dka)
    io.dgate_logits = _ptr(q)  # (placeholder for the query below: only tested for NULL)
    # the three per-(b, head) partial sums share ONE [B*nh, 4*dh + 4] buffer, reduced in a single pass by the caller
    width = 4 * dh + 4
    part = _out(B * n_heads, width, device=q.device) if not attack_only else torch.empty(B * n_heads, width, device=q.device)
    ws_bytes = int(lib.acattn_calibrated_attention_bwd_workspace_bytes(C.byref(prob)))
    ws = torch.empty(max(ws_bytes, 4) // 4, device=q.device, dtype=torch.float32)
    io.workspace = _ptr(ws)
    base = part.data_ptr()
    io.dw_order_part, io.dw_dist_part, io.dsmall_part = base, base + 4 * 2 * dh, base + 4 * 4 * dh
    io.part_stride = width
    io.active_qblocks = _ptr(active_qblocks)
    if read_rows is not None:
        io.read_rows, io.n_read_rows = _ptr(read_rows), read_rows.shape[1]
    io.attack_only = int(attack_only)
    io.d_penalty_part = _ptr(d_penalty_part)  # [B, n_heads, ceil(L/16)]: include/acattn.h
    # one read position per sequence: ONE row of each sequence's gate gradient is non-zero, and the one-row form of the
    # backward adds it straight into the head-summed [B,L,L] tensor (acattn_bwd_io.dgate_summed); returned as [B,1,L,L]
    summed = bool(lib.acattn_calibrated_attention_bwd_gate_summed(C.byref(prob), C.byref(io)))
    dgate_part = (_out if not attack_only else torch.empty)(B, 1 if summed else n_heads, L, L, device=q.device) if _POISON else \
        torch.empty(B, 1 if summed else n_heads, L, L, device=q.device, dtype=torch.float32)
    io.dgate_logits, io.dgate_summed = _ptr(dgate_part), int(summed)
    _lib.check(lib.acattn_calibrated_attention_bwd(C.byref(prob), C.byref(io), _stream()), "calibrated_attention_bwd")
    return dq, dk, dv, dqa, dka, dgate_part, part


def _bwd_meta(q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed,
              seed_tensor, gate_is_prob, attack_mask, row_stats, d_ctx_attacked, d_ctx_calibrated, d_attack_mask, read_rows,
              active_qblocks, attack_only, d_penalty_part=None):
    B, L, H = q.shape
    dh = H // n_heads
    e = lambda: torch.empty_like(q)
    # (the one-row form returns the gate gradient already summed over the heads: same rule as the library's query for the
    # dispatcher form -- one read position, no block bitmap, no mask cotangent or L <= 64 for the split, not attack-only)
    one = (read_rows is not None and read_rows.shape[1] == 1 and active_qblocks is None and not attack_only and L <= 208
           and dh in (16, 32, 64, 128)
           and ((d_attack_mask is None and d_penalty_part is None) or (L <= 64 and dh <= 64)))
    return e(), e(), e(), e(), e(), q.new_empty(B, 1 if one else n_heads, L, L), q.new_empty(B * n_heads, 4 * dh + 4)


_LIB.impl("calibrated_attention_fwd", _fwd_cuda, "CUDA")
_LIB.impl("calibrated_attention_fwd", _fwd_meta, "Meta")
_LIB.impl("calibrated_attention_bwd", _bwd_cuda, "CUDA")
_LIB.impl("calibrated_attention_bwd", _bwd_meta, "Meta")


# ---- autograd formula of the raw forward op (a caller that bypasses ops._CalibratedAttention) ---------------------------------
def _setup_context(ctx, inputs, output):
    (q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed, seed_tensor,
     gate_is_prob, affine, adversarial) = inputs[:20]
    if not adversarial:
        ctx.adversarial = False
        return
    ctx.adversarial = True
    ctx.args = (causal, n_heads, p_drop, seed, gate_is_prob)
    ctx.save_for_backward(q, k, v, qa, ka, gate, key_valid, w_order, b_order, w_dist, b_dist, scalar,
                          seed_tensor if seed_tensor is not None else q.new_empty(0), output[2], output[3])
    ctx.has_seed_tensor = seed_tensor is not None
    ctx.set_materialize_grads(False)


def _backward(ctx, d_att, d_cal, d_M, _d_stats, d_pen=None):
    if not ctx.adversarial:
        raise _lib.AcattnError("backward of the spatial-only operator is not provided")
    q, k, v, qa, ka, gate, key_valid, w_order, b_order, w_dist, b_dist, scalar, seed_t, M, stats = ctx.saved_tensors
    causal, n_heads, p_drop, seed, gate_is_prob = ctx.args
    con = lambda t: None if t is None else t.contiguous()
    dq, dk, dv, dqa, dka, dgate_part, part = torch.ops.acattn.calibrated_attention_bwd(
        q, k, v, qa, ka, gate, key_valid, causal, w_order, b_order, w_dist, b_dist, scalar, n_heads, p_drop, seed,
        seed_t if ctx.has_seed_tensor else None, gate_is_prob, M, stats, con(d_att), con(d_cal), con(d_M), None, None, False,
        con(d_pen))
    dh = q.shape[-1] // n_heads
    tot = part.sum(0)
    small = tot[4 * dh:]
    return (dq, dk, dv, dqa, dka, (dgate_part[:, 0] if dgate_part.shape[1] == 1 else dgate_part.sum(1)), None, None, tot[:2 * dh].view_as(w_order), small[0:1].view_as(b_order),
            tot[2 * dh:4 * dh].view_as(w_dist), small[1:2].view_as(b_dist), small[2:3].view_as(scalar), None, None, None, None,
            None, None, None, None)


torch.library.register_autograd("acattn::calibrated_attention_fwd", _backward, setup_context=_setup_context, lib=_LIB)
