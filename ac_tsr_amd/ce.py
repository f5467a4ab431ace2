"""Full-catalogue cross-entropy as one fused HIP operator (include/acattn.h: acattn_full_sort_ce_*).

`full_sort_cross_entropy(output, table, target)` == `CrossEntropyLoss()(output @ table.T, target)`
(recbole/model/sequential_recommender/acsasrec.py:117-120) but the [B, N] logits never exist in HBM:
forward and backward each sweep the item table once on the matrix cores.  No CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .state import DEFAULT as _DEFAULT_STATE
from .ops import _need_cuda, _ptr, _stream


def _problem(out, table, target) -> _lib.CeProblem:
    B, H = out.shape
    p = _lib.CeProblem()
    p.B, p.N, p.H = B, table.shape[0], H
    p.out, p.table, p.target = _ptr(out), _ptr(table), _ptr(target)
    return p


class _FullSortCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, table, target, state):
        ctx.state = state
        for name, t in (("output", out), ("item table", table)):
            _need_cuda(name, t)
        _need_cuda("target", target, torch.int64)
        assert out.dim() == 2 and table.dim() == 2 and out.shape[1] == table.shape[1] and target.shape == (out.shape[0],)
        lib = _lib.load()
        p = _problem(out, table, target)
        nbytes = lib.acattn_full_sort_ce_workspace_bytes(C.byref(p))
        if nbytes < 0:
            raise _lib.AcattnError(f"fused cross-entropy supports hidden sizes 64 and 128, got {out.shape[1]}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
        lse = torch.empty(out.shape[0], device=out.device, dtype=torch.float32)
        row_loss = torch.empty_like(lse)
        _lib.check(lib.acattn_full_sort_ce_fwd(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _stream()), "full_sort_ce_fwd")
        ctx.save_for_backward(out, table, target, lse)
        ctx.ws_bytes = nbytes
        ctx.tick = state.next_tick()
        return row_loss

    @staticmethod
    def backward(ctx, d_row_loss):
        out, table, target, lse = ctx.saved_tensors
        lib = _lib.load()
        p = _problem(out, table, target)
        ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=out.device)
        coef = d_row_loss.contiguous()
        d_out = torch.empty_like(out)
        # the item table is not an attack parameter: its gradient is dropped in the attacked-loss pass
        want_table = ctx.needs_input_grad[1] and not ctx.state.attack_pass_only
        d_table = ctx.state.grad_buffer_for(table) if want_table else None  # data parallel: straight into the flat buffer
        if want_table and d_table is None:
            d_table = torch.empty_like(table)
        _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(coef), _ptr(ws), _ptr(d_out), _ptr(d_table),
                                               _stream()), "full_sort_ce_bwd")
        ctx.state.publish_table_grad(getattr(ctx, "tick", -1), table, d_table)  # (see StepState.table_grad)
        return d_out, d_table, None, None


class _FullSortCEDir(torch.autograd.Function):
    """row losses whose TABLE gradient is never taken (the attacked loss under the two-pass protocol,
    recbole/trainer/trainer.py:678-684): the forward sweep also produces d row_loss / d output
    (acattn_full_sort_ce_fwd_dir), so the backward for `output` is one elementwise product instead of a second
    sweep of the catalogue.  If a caller does ask for the table gradient, it is computed by the regular backward."""

    @staticmethod
    def forward(ctx, out, table, target, state):
        ctx.state = state
        for name, t in (("output", out), ("item table", table)):
            _need_cuda(name, t)
        _need_cuda("target", target, torch.int64)
        lib = _lib.load()
        p = _problem(out, table, target)
        nbytes = lib.acattn_full_sort_ce_workspace_bytes(C.byref(p))
        if nbytes < 0:
            raise _lib.AcattnError(f"fused cross-entropy supports hidden sizes 64 and 128, got {out.shape[1]}")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
        lse = torch.empty(out.shape[0], device=out.device, dtype=torch.float32)
        row_loss = torch.empty_like(lse)
        direction = torch.empty_like(out)
        rc = lib.acattn_full_sort_ce_fwd_dir(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _ptr(direction), _stream())
        if rc == -100:  # too many rows for the per-workgroup slabs: plain forward, regular backward
            direction = None
            _lib.check(lib.acattn_full_sort_ce_fwd(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _stream()), "full_sort_ce_fwd")
        else:
            _lib.check(rc, "full_sort_ce_fwd_dir")
        ctx.has_dir = direction is not None
        ctx.save_for_backward(out, table, target, lse, direction if direction is not None else lse)
        ctx.tick = state.next_tick()
        ctx.ws_bytes = nbytes
        return row_loss

    @staticmethod
    def backward(ctx, d_row_loss):
        out, table, target, lse, direction = ctx.saved_tensors
        want_table = ctx.needs_input_grad[1] and not ctx.state.attack_pass_only
        if ctx.has_dir and not want_table:
            return direction * d_row_loss.unsqueeze(1), None, None, None
        lib = _lib.load()
        p = _problem(out, table, target)
        ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=out.device)
        d_out = torch.empty_like(out)
        d_table = ctx.state.grad_buffer_for(table) if want_table else None  # data parallel: straight into the flat buffer
        if want_table and d_table is None:
            d_table = torch.empty_like(table)
        _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(d_row_loss.contiguous()), _ptr(ws), _ptr(d_out),
                                               _ptr(d_table), _stream()), "full_sort_ce_bwd")
        ctx.state.publish_table_grad(getattr(ctx, "tick", -1), table, d_table)  # (see StepState.table_grad)
        return d_out, d_table, None, None


class _FullSortCEMean(torch.autograd.Function):
    """mean(_FullSortCE rows) as one node: the cotangent of the mean reaches the backward sweep as a device scalar
    (acattn_ce_problem.coef_is_scalar / coef_scale = 1/B) instead of being broadcast to a [B] tensor by two more
    launches."""

    @staticmethod
    def forward(ctx, out, table, target, state):
        row_loss = _FullSortCE.forward(ctx, out, table, target, state)
        return row_loss.mean()

    @staticmethod
    def backward(ctx, d_loss):
        out, table, target, lse = ctx.saved_tensors
        lib = _lib.load()
        p = _problem(out, table, target)
        p.coef_is_scalar, p.coef_scale = 1, 1.0 / out.shape[0]
        ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=out.device)
        d_out = torch.empty_like(out)
        want_table = ctx.needs_input_grad[1] and not ctx.state.attack_pass_only
        d_table = ctx.state.grad_buffer_for(table) if want_table else None  # data parallel: straight into the flat buffer
        if want_table and d_table is None:
            d_table = torch.empty_like(table)
        _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(d_loss.contiguous()), _ptr(ws), _ptr(d_out),
                                               _ptr(d_table), _stream()), "full_sort_ce_bwd")
        ctx.state.publish_table_grad(getattr(ctx, "tick", -1), table, d_table)  # (see StepState.table_grad)
        return d_out, d_table, None, None


class _AttackedLoss(torch.autograd.Function):
    """final_attacked_loss of ACSASRec.calculate_loss (acsasrec.py:129-137) as one node:

        -CrossEntropyLoss(output @ table^T, target) + weight * mean_l torch.norm(1 - M_l, p=2)

    Forward: the CE sweep that also yields d row_loss / d output (acattn_full_sort_ce_fwd_dir), one partial-sum
    launch per attack mask and ONE finishing launch for both means, the square roots and the combination (as torch
    ops: mean, neg, 2 x norm-finish, stack, mean, mul, add).  Backward: d output = direction * d_loss (one product;
    the -1/B is folded into the saved direction), d M_l = d_loss * weight / n * (M_l - 1) / ||1 - M_l||, one launch
    per mask.  The table gradient is never taken under the two-pass protocol (recbole/trainer/trainer.py:678-684);
    a caller that does ask for it gets it from the regular backward sweep."""

    @staticmethod
    def forward(ctx, out, table, target, weight, state, *masks):
        ctx.state = state
        lib = _lib.load()
        B = out.shape[0]
        p = _problem(out, table, target)
        nbytes = lib.acattn_full_sort_ce_workspace_bytes(C.byref(p))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
        lse = torch.empty(B, device=out.device, dtype=torch.float32)
        row_loss = torch.empty_like(lse)
        direction = torch.empty_like(out)
        rc = lib.acattn_full_sort_ce_fwd_dir(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _ptr(direction), _stream())
        if rc == -100:  # too many rows for the per-workgroup slabs: plain forward, regular backward sweep
            direction = None
            _lib.check(lib.acattn_full_sort_ce_fwd(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _stream()), "full_sort_ce_fwd")
        else:
            _lib.check(rc, "full_sort_ce_fwd_dir")
        masks = tuple(m.contiguous() for m in masks)
        part = torch.empty(len(masks), _lib.PENALTY_WS_FLOATS, device=out.device, dtype=torch.float32)
        if 1 < len(masks) <= _lib.MAX_MASKS:  # every mask in one launch
            ptrs = (C.c_void_p * len(masks))(*(m.data_ptr() for m in masks))
            _lib.check(lib.acattn_mask_penalty_partial_multi(ptrs, len(masks), masks[0].numel(), _ptr(part), _stream()),
                       "mask_penalty_partial_multi")
        else:
            for l, m in enumerate(masks):
                _lib.check(lib.acattn_mask_penalty_partial(_ptr(m), m.numel(), _ptr(part[l]), _stream()), "mask_penalty_partial")
        res = torch.empty(2 + len(masks), device=out.device, dtype=torch.float32)
        _lib.check(lib.acattn_attacked_loss_finish(_ptr(row_loss), B, _ptr(part), len(masks), masks[0].numel(), weight,
                                                   _ptr(res), _ptr(direction), 0 if direction is None else direction.numel(),
                                                   _stream()), "attacked_loss_finish")
        ctx.has_dir = direction is not None
        ctx.save_for_backward(out, table, target, lse, direction if direction is not None else lse, res, *masks)
        ctx.weight, ctx.ws_bytes = weight, nbytes
        return res[0]

    @staticmethod
    def backward(ctx, d_loss):
        out, table, target, lse, direction, res = ctx.saved_tensors[:6]
        masks = ctx.saved_tensors[6:]
        lib = _lib.load()
        d_loss = d_loss.contiguous()
        d_table = None
        want_table = ctx.needs_input_grad[1] and not ctx.state.attack_pass_only
        if want_table or not ctx.has_dir:
            p = _problem(out, table, target)
            p.coef_is_scalar, p.coef_scale = 1, -1.0 / out.shape[0]
            ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=out.device)
            d_out, d_table = torch.empty_like(out), (torch.empty_like(table) if want_table else None)
            _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(d_loss), _ptr(ws), _ptr(d_out), _ptr(d_table),
                                                   _stream()), "full_sort_ce_bwd")
        else:
            d_out = direction * d_loss  # direction already carries -1/B
        d_masks = []
        if 1 < len(masks) <= _lib.MAX_MASKS and all(ctx.needs_input_grad[5 + l] for l in range(len(masks))):
            d_masks = [torch.empty_like(m) for m in masks]  # every mask's gradient in one launch
            mp = (C.c_void_p * len(masks))(*(m.data_ptr() for m in masks))
            dp = (C.c_void_p * len(masks))(*(d.data_ptr() for d in d_masks))
            _lib.check(lib.acattn_mask_penalty_bwd_scaled_multi(mp, _ptr(res[2:]), _ptr(d_loss), ctx.weight / len(masks),
                                                                masks[0].numel(), dp, len(masks), _stream()),
                       "mask_penalty_bwd_scaled_multi")
            return (d_out, d_table, None, None, None, *d_masks)
        for l, m in enumerate(masks):
            if not ctx.needs_input_grad[5 + l]:
                d_masks.append(None)
                continue
            d_m = torch.empty_like(m)
            _lib.check(lib.acattn_mask_penalty_bwd_scaled(_ptr(m), _ptr(res[2 + l]), _ptr(d_loss), ctx.weight / len(masks),
                                                          m.numel(), _ptr(d_m), _stream()), "mask_penalty_bwd_scaled")
            d_masks.append(d_m)
        return (d_out, d_table, None, None, None, *d_masks)


class _AttackedLossRows(torch.autograd.Function):
    """_AttackedLoss with the mask penalty taken from the attention nodes' row sums (ops.PENALTY_ROWS: `pen_l` =
    acattn_mask_penalty_rows(M_l), [B, nh, ceil(L/16)], || 1 - M_l ||^2 = sum pen_l) instead of from the masks: no pass
    over M in the forward, and the backward returns d pen_l = d_loss * weight / n / (2 || 1 - M_l ||) -- a vector of a few
    thousand equal entries per layer -- instead of a dense d M_l; the attention backward kernels form
    d M = 2 d_pen (M - 1) from the tile they rebuild (acattn_bwd_io.d_penalty_part)."""

    @staticmethod
    def forward(ctx, out, table, target, weight, state, *pens):
        ctx.state = state
        lib = _lib.load()
        B = out.shape[0]
        p = _problem(out, table, target)
        nbytes = lib.acattn_full_sort_ce_workspace_bytes(C.byref(p))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
        lse = torch.empty(B, device=out.device, dtype=torch.float32)
        row_loss = torch.empty_like(lse)
        direction = torch.empty_like(out)
        rc = lib.acattn_full_sort_ce_fwd_dir(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _ptr(direction), _stream())
        if rc == -100:  # too many rows for the per-workgroup slabs: plain forward, regular backward sweep
            direction = None
            _lib.check(lib.acattn_full_sort_ce_fwd(C.byref(p), _ptr(ws), _ptr(lse), _ptr(row_loss), _stream()), "full_sort_ce_fwd")
        else:
            _lib.check(rc, "full_sort_ce_fwd_dir")
        pens = tuple(t.contiguous() for t in pens)
        ptrs = (C.c_void_p * len(pens))(*(t.data_ptr() for t in pens))
        res = torch.empty(2 + len(pens), device=out.device, dtype=torch.float32)
        _lib.check(lib.acattn_attacked_loss_finish_rows(_ptr(row_loss), B, ptrs, len(pens), pens[0].numel(), weight, _ptr(res),
                                                        _ptr(direction), 0 if direction is None else direction.numel(),
                                                        _stream()), "attacked_loss_finish_rows")
        ctx.has_dir = direction is not None
        ctx.save_for_backward(out, table, target, lse, direction if direction is not None else lse, res)
        ctx.weight, ctx.ws_bytes, ctx.pen_shapes = weight, nbytes, [t.shape for t in pens]
        return res[0]

    @staticmethod
    def backward(ctx, d_loss):
        out, table, target, lse, direction, res = ctx.saved_tensors
        lib = _lib.load()
        d_loss = d_loss.contiguous()
        d_table = None
        want_table = ctx.needs_input_grad[1] and not ctx.state.attack_pass_only
        if want_table or not ctx.has_dir:
            p = _problem(out, table, target)
            p.coef_is_scalar, p.coef_scale = 1, -1.0 / out.shape[0]
            ws = torch.empty(ctx.ws_bytes, dtype=torch.uint8, device=out.device)
            d_out, d_table = torch.empty_like(out), (torch.empty_like(table) if want_table else None)
            _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(d_loss), _ptr(ws), _ptr(d_out), _ptr(d_table),
                                                   _stream()), "full_sort_ce_bwd")
            from_dir = False
        else:
            d_out = torch.empty_like(direction)  # = direction * d_loss (direction already carries -1/B), in the launch below
            from_dir = True
        n = len(ctx.pen_shapes)
        count = 1
        for d in ctx.pen_shapes[0]:
            count *= d
        d_flat = torch.empty(n, count, device=out.device, dtype=torch.float32)  # every layer's vector in one launch
        dp = (C.c_void_p * n)(*(d_flat[l].data_ptr() for l in range(n)))
        if from_dir:
            _lib.check(lib.acattn_mask_penalty_drows_dir(_ptr(res[2:]), _ptr(d_loss), ctx.weight / n, count, dp, n, _ptr(direction),
                                                         _ptr(d_out), direction.numel(), _stream()), "mask_penalty_drows_dir")
        else:
            _lib.check(lib.acattn_mask_penalty_drows(_ptr(res[2:]), _ptr(d_loss), ctx.weight / n, count, dp, n, _stream()),
                       "mask_penalty_drows")
        d_pens = [d_flat[l].view(ctx.pen_shapes[l]) if ctx.needs_input_grad[5 + l] else None for l in range(n)]
        return (d_out, d_table, None, None, None, *d_pens)


def attacked_loss(output: torch.Tensor, table: torch.Tensor, target: torch.Tensor, masks, weight: float,
                  state=_DEFAULT_STATE):
    """-CE(output @ table^T, target) + weight * mean_l ||1 - M_l||_2 (acsasrec.py:129-137) as one autograd node, or None
    when the fused form does not apply (no mask, masks of different sizes)."""
    masks = [m for m in masks if m is not None]
    if not masks or any(m.numel() != masks[0].numel() or m.dtype != torch.float32 or not m.is_cuda for m in masks):
        return None
    _need_cuda("target", target, torch.int64)
    from . import ops
    pens = [getattr(m, "_acattn_pen", None) for m in masks]
    if (ops.PENALTY_ROWS and len(masks) <= _lib.MAX_MASKS and all(t is not None and t.is_cuda for t in pens)
            and all(t.shape == pens[0].shape for t in pens)):
        return _AttackedLossRows.apply(output.contiguous(), table, target, float(weight), state, *pens)
    return _AttackedLoss.apply(output.contiguous(), table, target, float(weight), state, *masks)


def full_sort_cross_entropy(output: torch.Tensor, table: torch.Tensor, target: torch.Tensor,
                            table_grad: bool = True, state=_DEFAULT_STATE) -> torch.Tensor:
    """mean_b [ logsumexp_n(output_b . table_n) - output_b . table_target(b) ].  `table_grad=False` declares that the
    table's gradient of this loss will not be taken (see _FullSortCEDir): same values, cheaper backward."""
    if table_grad:
        return _FullSortCEMean.apply(output.contiguous(), table, target, state)
    return _FullSortCEDir.apply(output.contiguous(), table, target, state).mean()


def full_sort_cross_entropy_rows(output: torch.Tensor, table: torch.Tensor, target: torch.Tensor,
                                 table_grad: bool = True, state=_DEFAULT_STATE) -> torch.Tensor:
    """Per-row losses `CrossEntropyLoss(reduction='none')(output @ table.T, target)` (acbert4rec.py:201-206)."""
    fn = _FullSortCE if table_grad else _FullSortCEDir
    return fn.apply(output.contiguous(), table, target, state)


class _DenseCE(torch.autograd.Function):
    """CrossEntropyLoss(reduction='none') over MATERIALISED logits [rows, N] (acbert4rec.py:201-209 at the widths where the
    catalogue product stays a library GEMM): row log-sum-exp in the forward (the row is read once, nothing of size
    [rows, N] is written), d logits = coef (softmax - onehot) in one elementwise pass in the backward -- torch's pair
    writes the [rows, N] log-probabilities, zero-fills a [rows, N] tensor and reads both back (1.6 GB each at
    20k x 20k: 0.6 + 0.23 + 0.9 ms per loss at configs[4])."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous()
        rows, n = logits.shape
        lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
        row_loss = torch.empty_like(lse)
        _lib.check(_lib.load().acattn_dense_ce_fwd(_ptr(logits), rows, n, _ptr(target), _ptr(lse), _ptr(row_loss), _stream()),
                   "dense_ce_fwd")
        ctx.save_for_backward(logits, target, lse)
        return row_loss

    @staticmethod
    def backward(ctx, d_rows):
        logits, target, lse = ctx.saved_tensors
        rows, n = logits.shape
        d_logits = torch.empty_like(logits)
        _lib.check(_lib.load().acattn_dense_ce_bwd(_ptr(logits), _ptr(lse), _ptr(target), _ptr(d_rows.contiguous()), rows, n,
                                                   _ptr(d_logits), _stream()), "dense_ce_bwd")
        return d_logits, None


def dense_cross_entropy_rows(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Per-row CE over materialised logits; torch's own on the host / for other dtypes."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and target.dtype == torch.int64:
        return _DenseCE.apply(logits, target.contiguous())
    return torch.nn.functional.cross_entropy(logits, target, reduction='none')


def supported(hidden_size: int) -> bool:
    return hidden_size in (64, 128, 256)
