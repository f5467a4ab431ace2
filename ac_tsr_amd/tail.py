"""The tail of one branch of an encoder layer as ONE autograd node:

    a   = LayerNorm(dropout(dense(ctx)) + x)                  cal_adjusted_outputs, recbole/model/layers.py:681-683
    out = LayerNorm(dropout(dense_2(gelu(dense_1(a)))) + a)   FeedForward.forward,  layers.py:790-798

`_FusedLayerTail` (hidden 64, inner 256 / 128: the shipped configuration): ONE HIP launch forward
(acattn_layer_tail_fwd: the three products on the fp32 matrix cores with the chain held in registers, csrc/acattn_tail.hip)
and, backward, one launch for every input gradient and the LayerNorm partials (acattn_layer_tail_bwd) + the grouped
weight-gradient launch pair + one reduction of the partials.

`_LayerTail` (other sizes): the same three GEMMs through hipBLASLt, torch's erf-GELU and two fused
dropout+residual+LayerNorm launches; backward written out by hand: the two LayerNorm backward launches, GELU backward,
three input-gradient GEMMs of which the one that meets the residual stream accumulates in place (beta = 1), ONE grouped
launch pair for the three weight/bias gradients and ONE reduction for both LayerNorms' (dgamma, dbeta) partials --
instead of six autograd nodes with their own reductions and the elementwise adds autograd inserts where `a` fans out.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib, fused_ln, ops
from .ops import _ptr, _stream
from .state import state_of

# measurement switch (bench.py --tail unfused): route supported sizes through the unfused node as well
FUSED_KERNEL = True
FUSED_MAX_HIDDEN = 128  # widest hidden size routed to the single-launch node by default (256 is supported, see fused_supported)


def _tail_problem(c, x, wd, bd, g1, b1, w1, bb1, w2, bb2, g2, b2, eps1, eps2, p1, p2, k1, k2, seed1, seed2, seed_t,
                  rows=None):
    p = _lib.TailProblem()
    H, I = wd.shape[0], w1.shape[0]
    p.rows, p.H, p.I = c.numel() // H, H, I
    if rows is not None:  # [B, R] positions picked out of c / x [B, L, H]
        p.rows, p.src_index, p.src_R, p.src_L = rows.numel(), _ptr(rows), rows.shape[-1], c.shape[-2]
    p.ctx, p.x = _ptr(c), _ptr(x)
    p.wd, p.bd, p.g1, p.b1 = _ptr(wd), _ptr(bd), _ptr(g1), _ptr(b1)
    p.w1, p.bb1, p.w2, p.bb2, p.g2, p.b2 = _ptr(w1), _ptr(bb1), _ptr(w2), _ptr(bb2), _ptr(g2), _ptr(b2)
    p.eps1, p.eps2, p.p1, p.p2 = float(eps1), float(eps2), float(p1), float(p2)
    p.keep1, p.keep2 = _ptr(k1), _ptr(k2)
    p.seed1, p.seed2 = seed1 & 0xFFFFFFFFFFFFFFFF, seed2 & 0xFFFFFFFFFFFFFFFF
    p.seed_device = _ptr(seed_t)
    return p


class _FusedLayerTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, x, wd, bd, g1, b1, w1, bb1, w2, bb2, g2, b2, eps1, eps2, p1, p2, keep1, keep2, seed1, seed2,
                seed_tensor, state, pick=None):
        """`pick` ([B, R] int64 positions): the tail runs on those positions of c / x ([B, L, H]) only and returns
        [B, R, H]; explicit keep masks are then [B, R, H] as well."""
        ctx.state = state
        c, x = c.contiguous(), x.contiguous()
        params = tuple(t.contiguous() for t in (wd, bd, g1, b1, w1, bb1, w2, bb2, g2, b2))
        k1 = None if keep1 is None else keep1.to(torch.uint8).contiguous()
        k2 = None if keep2 is None else keep2.to(torch.uint8).contiguous()
        H, I = c.shape[-1], w1.shape[0]
        if pick is not None:
            pick = pick.contiguous()
            assert c.dim() == 3 and x.shape == c.shape and pick.dtype == torch.int64 and pick.shape[0] == c.shape[0]
            out_shape = (*pick.shape, H)
        else:
            out_shape = c.shape
        rows = 1
        for d in out_shape[:-1]:
            rows *= d
        new = lambda *shape: torch.empty(*shape, device=c.device, dtype=torch.float32)
        h1, a, h3, out = (new(*out_shape) for _ in range(4))
        st1, st2, act = new(rows, 2), new(rows, 2), new(rows, I)
        # hidden 128: gelu'(dense_1(a)) is saved as well (the backward would spend a fourth of its matrix work rebuilding it)
        dgelu = new(rows, I) if (H > 64 and any(ctx.needs_input_grad)) else None
        p = _tail_problem(c, x, *params, eps1, eps2, p1, p2, k1, k2, seed1, seed2, seed_tensor, pick)
        sv = _lib.TailSaved()
        sv.h1, sv.st1, sv.a, sv.act, sv.h3, sv.st2, sv.out = (_ptr(t) for t in (h1, st1, a, act, h3, st2, out))
        sv.gelu_grad = _ptr(dgelu)
        _lib.check(_lib.load().acattn_layer_tail_fwd(C.byref(p), C.byref(sv), _stream()), "layer_tail_fwd")
        empty = c.new_empty(0)
        ctx.save_for_backward(c, x, h1, st1, a, act, h3, st2, *params, k1 if k1 is not None else empty,
                              k2 if k2 is not None else empty, seed_tensor if seed_tensor is not None else empty,
                              pick if pick is not None else empty, dgelu if dgelu is not None else empty)
        ctx.args = (eps1, eps2, p1, p2, k1 is not None, k2 is not None, seed1, seed2, seed_tensor is not None,
                    pick is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        c, x, h1, st1, a, act, h3, st2 = ctx.saved_tensors[:8]
        params = ctx.saved_tensors[8:18]
        k1, k2, seed_t, pick, dgelu = ctx.saved_tensors[18:]
        eps1, eps2, p1, p2, has_k1, has_k2, seed1, seed2, has_seed_t, has_pick = ctx.args
        k1, k2, seed_t = (k1 if has_k1 else None), (k2 if has_k2 else None), (seed_t if has_seed_t else None)
        pick = pick if has_pick else None
        want_params = not ctx.state.attack_pass_only  # none of these is an attack transform (trainer.py:678-684)
        need_c, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        lib = _lib.load()
        rows, H, I = act.shape[0], c.shape[-1], act.shape[-1]
        new = lambda *shape: torch.empty(*shape, device=c.device, dtype=torch.float32)
        d_out = d_out.contiguous()
        p = _tail_problem(c, x, *params, eps1, eps2, p1, p2, k1, k2, seed1, seed2, seed_t, pick)
        sv = _lib.TailSaved()
        sv.h1, sv.st1, sv.a, sv.act, sv.h3, sv.st2 = (_ptr(t) for t in (h1, st1, a, act, h3, st2))
        sv.gelu_grad = _ptr(dgelu) if dgelu.numel() else None
        io = _lib.TailBwdIO()
        # with a row selection the kernel writes the picked positions only: the rest of the gradient is zero
        if pick is not None and need_c and need_x:
            d_c, d_x = torch.zeros(2, *c.shape, device=c.device, dtype=c.dtype).unbind(0)  # one fill launch for both
        else:
            alloc = torch.zeros_like if pick is not None else torch.empty_like
            d_c = alloc(c) if need_c else None
            d_x = alloc(x) if need_x else None
        io.d_out, io.d_ctx, io.d_x = _ptr(d_out), _ptr(d_c), _ptr(d_x)
        d_h1 = d_h2 = d_h3 = part = None
        if want_params:
            d_h1, d_h2, d_h3 = new(rows, H), new(rows, I), new(rows, H)
            part = new(int(lib.acattn_layer_tail_bwd_partial_rows_for(rows, H)), 4 * H)
            io.d_h1, io.d_h2, io.d_h3, io.dgb_part = _ptr(d_h1), _ptr(d_h2), _ptr(d_h3), _ptr(part)
        ws_bytes = int(lib.acattn_layer_tail_bwd_workspace_bytes(H, I))
        ws = new(ws_bytes // 4) if ws_bytes > 0 else None  # transposed weight copies at hidden 128
        io.workspace = _ptr(ws)
        _lib.check(lib.acattn_layer_tail_bwd(C.byref(p), C.byref(sv), C.byref(io), _stream()), "layer_tail_bwd")
        grads = [None] * 10  # wd, bd, g1, b1, w1, bb1, w2, bb2, g2, b2
        if want_params:
            two = lambda t: t.reshape(-1, t.shape[-1])
            c_rows = two(c) if pick is None else c.gather(1, pick.unsqueeze(-1).expand(-1, -1, H)).view(-1, H)
            (gwd, gbd), (gw1, gb1), (gw2, gb2) = ops.linear_wgrad_grouped(
                [(c_rows, d_h1, True), (two(a), d_h2, True), (act, d_h3, True)], ctx.state)
            gb = ops.sum_rows0(part, ctx.state).view(4, H)  # (dgamma1, dbeta1, dgamma2, dbeta2)
            grads = [gwd, gbd, gb[0], gb[1], gw1, gb1, gw2, gb2, gb[2], gb[3]]
            ctx.state.watch(gb, grads[2], grads[3], grads[8], grads[9])
        return (d_c, d_x, *grads, None, None, None, None, None, None, None, None, None, None, None)



class _LayerTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, x, wd, bd, g1, b1, w1, bb1, w2, bb2, g2, b2, eps1, eps2, p1, p2, keep1, keep2, seed1, seed2,
                seed_tensor, state):
        ctx.state = state
        c, x = c.contiguous(), x.contiguous()
        k1 = None if keep1 is None else keep1.to(torch.uint8).contiguous()
        k2 = None if keep2 is None else keep2.to(torch.uint8).contiguous()
        h1 = F.linear(c, wd, bd)
        a, st1 = fused_ln.forward_raw(h1, x, g1, b1, eps1, p1, k1, seed1, seed_tensor)
        h2 = F.linear(a, w1, bb1)
        act = F.gelu(h2)
        h3 = F.linear(act, w2, bb2)
        out, st2 = fused_ln.forward_raw(h3, a, g2, b2, eps2, p2, k2, seed2, seed_tensor)
        empty = c.new_empty(0)
        ctx.save_for_backward(c, x, h1, st1, a, h2, act, h3, st2, wd, g1, b1, w1, w2, g2, b2,
                              k1 if k1 is not None else empty, k2 if k2 is not None else empty,
                              seed_tensor if seed_tensor is not None else empty)
        ctx.args = (eps1, eps2, p1, p2, k1 is not None, k2 is not None, seed1, seed2, seed_tensor is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (c, x, h1, st1, a, h2, act, h3, st2, wd, g1, b1, w1, w2, g2, b2, k1, k2, seed_t) = ctx.saved_tensors
        eps1, eps2, p1, p2, has_k1, has_k2, seed1, seed2, has_seed_t = ctx.args
        k1, k2, seed_t = (k1 if has_k1 else None), (k2 if has_k2 else None), (seed_t if has_seed_t else None)
        params = not ctx.state.attack_pass_only  # none of these parameters is an attack transform (trainer.py:678-684)
        two = lambda t: t.reshape(-1, t.shape[-1])
        # feed-forward block
        d_h3, d_a, part2 = fused_ln.backward_raw(h3, a, g2, b2, eps2, p2, k2, seed2, seed_t, st2, d_out, want_gb=params)
        d_act = two(d_h3) @ w2
        d_h2 = torch.ops.aten.gelu_backward(d_act.view_as(h2), h2)
        two(d_a).addmm_(two(d_h2), w1)  # d a: residual path + through dense_1, accumulated by the GEMM
        # attention-output block
        need_c, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_h1, d_x, part1 = fused_ln.backward_raw(h1, x, g1, b1, eps1, p1, k1, seed1, seed_t, st1, d_a, want_dres=need_x,
                                                 want_gb=params)
        d_c = (two(d_h1) @ wd).view_as(c) if need_c else None
        grads = [None] * 10  # wd, bd, g1, b1, w1, bb1, w2, bb2, g2, b2
        if params:
            (gwd, gbd), (gw1, gb1), (gw2, gb2) = ops.linear_wgrad_grouped(
                [(two(c), two(d_h1), True), (two(a), two(d_h2), True), (two(act), two(d_h3), True)], ctx.state)
            gb = ops.sum_rows(torch.stack((part1, part2)), 1)  # [2 norms, (dgamma, dbeta), H] in one pass
            grads = [gwd, gbd, gb[0, 0], gb[0, 1], gw1, gb1, gw2, gb2, gb[1, 0], gb[1, 1]]
        return (d_c, d_x, *grads, None, None, None, None, None, None, None, None, None, None)


def supported(att, ffn) -> bool:
    """The fused node covers the shipped configuration: erf-GELU, biased linears, hidden sizes of the fused LayerNorm."""
    return (ffn.intermediate_act_fn == ffn.gelu and fused_ln.supported(att.dense.out_features)
            and att.dense.bias is not None and ffn.dense_1.bias is not None and ffn.dense_2.bias is not None)


def fused_supported(att, ffn) -> bool:
    """The single-launch node covers this (hidden, inner) pair (it can also pick rows itself: layer_tail(pick=...))."""
    H = att.dense.out_features
    # hidden 256: the launch exists and is parity-tested, but measured slower than the hipBLASLt node on MI355X (one wave
    # per SIMD: 4.72 against 3.91 ms forward + backward at 102,400 rows; configs[4] 45.6 against 43.1 ms per step)
    if H > FUSED_MAX_HIDDEN:
        return False
    return FUSED_KERNEL and bool(_lib.load().acattn_layer_tail_supported(H, ffn.dense_1.out_features))


def layer_tail(ctx_layer, input_tensor, att, ffn, keep_out=None, keep_ffn=None, pick=None):
    """FeedForward(attention_output(ctx_layer, input_tensor)) for one branch; `att` is the AttackRMultiHeadAttention
    (dense, LayerNorm, out_dropout), `ffn` the FeedForward module.  keep_* feed explicit dropout masks (parity).
    `pick` ([B, R] positions; fused node only): run on those positions of the two [B, L, H] inputs, return [B, R, H]."""
    training = att.training
    p1 = att.out_dropout.p if (training or keep_out is not None) else 0.0
    p2 = ffn.dropout.p if (training or keep_ffn is not None) else 0.0
    state = state_of(att)
    seed1 = state.draw_seed() if (p1 > 0 and keep_out is None) else 0
    seed2 = state.draw_seed() if (p2 > 0 and keep_ffn is None) else 0
    seed_t = state.seed_tensor if (keep_out is None and keep_ffn is None) else None
    extra = ()
    node = _LayerTail
    if fused_supported(att, ffn):
        node, extra = _FusedLayerTail, (pick,)
    else:
        assert pick is None, "row selection inside the tail needs the fused node"
    return node.apply(ctx_layer, input_tensor, att.dense.weight, att.dense.bias, att.LayerNorm.weight,
                            att.LayerNorm.bias, ffn.dense_1.weight, ffn.dense_1.bias, ffn.dense_2.weight,
                            ffn.dense_2.bias, ffn.LayerNorm.weight, ffn.LayerNorm.bias, att.LayerNorm.eps,
                            ffn.LayerNorm.eps, p1, p2, keep_out, keep_ffn, seed1, seed2, seed_t, state, *extra)
