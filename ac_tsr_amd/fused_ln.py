"""LayerNorm(dropout(z) + residual) as one HIP launch each way (include/acattn.h: acattn_dropout_add_layernorm_*).

Used for the tail of both sub-blocks of a layer (recbole/model/layers.py:681-683, 794-796).  The dropout draws
come from the library's counter RNG (one 63-bit seed per call from torch's CPU generator + the trainer's
device-side step counter under hipGraph capture); `keep` feeds an explicit mask for parity tests.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops
from .state import state_of
from .ops import _need_cuda, _ptr, _stream


def _problem(z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor) -> _lib.LnProblem:
    H = z.shape[-1]
    p = _lib.LnProblem()
    p.rows, p.H, p.residual_rows = z.numel() // H, H, res.numel() // H
    p.z, p.residual, p.gamma, p.beta = _ptr(z), _ptr(res), _ptr(gamma), _ptr(beta)
    p.eps, p.p_drop = float(eps), float(p_drop)
    p.keep = _ptr(keep)
    p.seed = seed & 0xFFFFFFFFFFFFFFFF
    p.seed_device = _ptr(seed_tensor)
    return p


def forward_raw(z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor):
    """(y, stats) of one acattn_dropout_add_layernorm_fwd launch; no autograd (building block of fused nodes)."""
    p = _problem(z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor)
    y = torch.empty_like(z)
    stats = torch.empty(p.rows, 2, device=z.device, dtype=torch.float32)
    _lib.check(_lib.load().acattn_dropout_add_layernorm_fwd(C.byref(p), _ptr(y), _ptr(stats), _stream()),
               "dropout_add_layernorm_fwd")
    return y, stats


def backward_raw(z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor, stats, dy, want_dz=True, want_dres=True,
                 want_gb=True):
    """(dz, dres, partials [LN_BWD_GRID, 2, H] of (dgamma, dbeta)) of one acattn_dropout_add_layernorm_bwd launch."""
    p = _problem(z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor)
    dz = torch.empty_like(z) if want_dz else None
    dres = torch.empty_like(z) if want_dres else None
    part = torch.empty(_lib.LN_BWD_GRID, 2, z.shape[-1], device=z.device, dtype=torch.float32) if want_gb else None
    _lib.check(_lib.load().acattn_dropout_add_layernorm_bwd(C.byref(p), _ptr(dy.contiguous()), _ptr(stats), _ptr(dz),
                                                            _ptr(dres), _ptr(part), _stream()), "dropout_add_layernorm_bwd")
    return dz, dres, part


class _DropoutAddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor, state):
        ctx.state = state
        for name, t in (("z", z), ("residual", res), ("LayerNorm.weight", gamma), ("LayerNorm.bias", beta)):
            _need_cuda(name, t)
        if keep is not None:
            keep = keep.to(torch.uint8).contiguous()
            assert keep.shape == z.shape
        lib = _lib.load()
        p = _problem(z, res, gamma, beta, eps, p_drop, keep, seed, seed_tensor)
        y = torch.empty_like(z)
        stats = torch.empty(p.rows, 2, device=z.device, dtype=torch.float32)
        _lib.check(lib.acattn_dropout_add_layernorm_fwd(C.byref(p), _ptr(y), _ptr(stats), _stream()), "dropout_add_layernorm_fwd")
        ctx.save_for_backward(z, res, gamma, beta, stats, keep if keep is not None else torch.empty(0), seed_tensor
                              if seed_tensor is not None else torch.empty(0))
        ctx.args = (eps, p_drop, keep is not None, seed, seed_tensor is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, res, gamma, beta, stats, keep, seed_tensor = ctx.saved_tensors
        eps, p_drop, has_keep, seed, has_seed_t = ctx.args
        lib = _lib.load()
        p = _problem(z, res, gamma, beta, eps, p_drop, keep if has_keep else None, seed, seed_tensor if has_seed_t else None)
        dy = dy.contiguous()
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        same = res.numel() == z.numel()
        dres_full = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        want_gb = (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) and not ctx.state.attack_pass_only
        part = torch.empty(_lib.LN_BWD_GRID, 2, z.shape[-1], device=z.device, dtype=torch.float32) if want_gb else None
        _lib.check(lib.acattn_dropout_add_layernorm_bwd(C.byref(p), _ptr(dy), _ptr(stats), _ptr(dz), _ptr(dres_full),
                                                        _ptr(part), _stream()), "dropout_add_layernorm_bwd")
        dres = dres_full
        if dres_full is not None and not same:
            dres = ops.sum_rows(dres_full.view(-1, *res.shape), 0)
        dgamma = dbeta = None
        if part is not None:
            gb = ops.sum_rows(part, 0)
            dgamma, dbeta = gb[0], gb[1]
        return dz, dres, dgamma, dbeta, None, None, None, None, None, None


def dropout_add_layer_norm(z: torch.Tensor, residual: torch.Tensor, norm: torch.nn.LayerNorm, p_drop: float,
                           training: bool, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """norm(dropout(z, p_drop, training) + residual).  `keep` (0/1, shape of z) replaces the random draw."""
    p = p_drop if (training or keep is not None) else 0.0
    state = state_of(norm)
    seed = state.draw_seed() if (p > 0 and keep is None) else 0
    return _DropoutAddLayerNorm.apply(z.contiguous(), residual.contiguous(), norm.weight, norm.bias, norm.eps, p, keep, seed,
                                      state.seed_tensor if keep is None else None, state)


def supported(hidden_size: int) -> bool:
    return hidden_size in (64, 128, 256)
