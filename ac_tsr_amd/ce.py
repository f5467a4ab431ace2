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
        d_table = torch.empty_like(table) if want_table else None
        _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(coef), _ptr(ws), _ptr(d_out), _ptr(d_table),
                                               _stream()), "full_sort_ce_bwd")
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
        d_table = torch.empty_like(table) if want_table else None
        _lib.check(lib.acattn_full_sort_ce_bwd(C.byref(p), _ptr(lse), _ptr(d_row_loss.contiguous()), _ptr(ws), _ptr(d_out),
                                               _ptr(d_table), _stream()), "full_sort_ce_bwd")
        return d_out, d_table, None, None


def full_sort_cross_entropy(output: torch.Tensor, table: torch.Tensor, target: torch.Tensor,
                            table_grad: bool = True, state=_DEFAULT_STATE) -> torch.Tensor:
    """mean_b [ logsumexp_n(output_b . table_n) - output_b . table_target(b) ].  `table_grad=False` declares that the
    table's gradient of this loss will not be taken (see _FullSortCEDir): same values, cheaper backward."""
    fn = _FullSortCE if table_grad else _FullSortCEDir
    return fn.apply(output.contiguous(), table, target, state).mean()


def full_sort_cross_entropy_rows(output: torch.Tensor, table: torch.Tensor, target: torch.Tensor,
                                 table_grad: bool = True, state=_DEFAULT_STATE) -> torch.Tensor:
    """Per-row losses `CrossEntropyLoss(reduction='none')(output @ table.T, target)` (acbert4rec.py:201-206)."""
    fn = _FullSortCE if table_grad else _FullSortCEDir
    return fn.apply(output.contiguous(), table, target, state)


def supported(hidden_size: int) -> bool:
    return hidden_size in (64, 128)
