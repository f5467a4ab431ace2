"""dropout(LayerNorm(item_embedding[item_seq] + position_embedding)) as one HIP launch each way
(include/acattn.h: acattn_embed_layernorm_*): the front end of ACSASRec.forward / AcBERT4Rec.forward
(recbole/model/sequential_recommender/acsasrec.py:87-95, acbert4rec.py:163-171).

The backward scatters the table gradient with float atomics into a zero-initialised [N, H] buffer, skipping
`padding_idx` rows like nn.Embedding; position and LayerNorm gradients come back as per-workgroup partials that are
summed here.  Dropout draws: the library's counter RNG (seed from torch's CPU generator + the trainer's device-side
step counter under hipGraph capture); `keep` feeds an explicit mask for parity tests.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops
from .state import state_of
from .ops import _need_cuda, _ptr, _stream


def _problem(idx, table, pos, gamma, beta, eps, p_drop, keep, seed, seed_tensor) -> _lib.EmbedProblem:
    p = _lib.EmbedProblem()
    p.rows, p.L, p.H = idx.numel(), idx.shape[-1], table.shape[1]
    p.n_table_rows = table.shape[0]
    p.idx, p.table, p.pos, p.gamma, p.beta = _ptr(idx), _ptr(table), _ptr(pos), _ptr(gamma), _ptr(beta)
    p.eps, p.p_drop = float(eps), float(p_drop)
    p.keep = _ptr(keep)
    p.seed = seed & 0xFFFFFFFFFFFFFFFF
    p.seed_device = _ptr(seed_tensor)
    return p


class _EmbedLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table, pos, gamma, beta, eps, p_drop, keep, seed, seed_tensor, padding_idx, state, hot_id=None):
        ctx.state = state
        ctx.hot_id = hot_id
        _need_cuda("item_seq", idx, torch.int64)
        for name, t in (("item_embedding.weight", table), ("LayerNorm.weight", gamma), ("LayerNorm.bias", beta)):
            _need_cuda(name, t)
        if pos is not None:
            _need_cuda("position_embedding.weight", pos)
            if idx.shape[-1] > pos.shape[0]:
                raise IndexError("index out of range in self")  # nn.Embedding's error for position ids >= rows
        if keep is not None:
            keep = keep.to(torch.uint8).contiguous()
            assert keep.shape == (*idx.shape, table.shape[1])
        lib = _lib.load()
        p = _problem(idx, table, pos, gamma, beta, eps, p_drop, keep, seed, seed_tensor)
        y = torch.empty(*idx.shape, table.shape[1], device=table.device, dtype=torch.float32)
        stats = torch.empty(p.rows, 2, device=table.device, dtype=torch.float32)
        nonzero = torch.empty(idx.shape, device=table.device, dtype=torch.uint8)  # idx != 0: the mask's validity bytes
        p.nonzero_out = _ptr(nonzero)
        _lib.check(lib.acattn_embed_layernorm_fwd(C.byref(p), _ptr(y), _ptr(stats), _stream()), "embed_layernorm_fwd")
        empty = torch.empty(0)
        ctx.save_for_backward(idx, table, pos if pos is not None else empty, gamma, beta, stats,
                              keep if keep is not None else empty, seed_tensor if seed_tensor is not None else empty)
        ctx.args = (eps, p_drop, pos is not None, keep is not None, seed, seed_tensor is not None, padding_idx)
        ctx.tick = state.next_tick()
        ctx.mark_non_differentiable(nonzero)
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" of the validity bytes (one launch per backward)
        return y, nonzero

    @staticmethod
    def backward(ctx, dy, _d_nonzero=None):
        idx, table, pos, gamma, beta, stats, keep, seed_tensor = ctx.saved_tensors
        eps, p_drop, has_pos, has_keep, seed, has_seed_t, padding_idx = ctx.args
        if ctx.state.attack_pass_only or dy is None:  # none of these parameters is an attack transform (trainer.py:678-684)
            return (None,) * 13
        lib = _lib.load()
        p = _problem(idx, table, pos if has_pos else None, gamma, beta, eps, p_drop, keep if has_keep else None, seed,
                     seed_tensor if has_seed_t else None)
        if ctx.hot_id is not None:
            p.hot_id_plus1 = int(ctx.hot_id) + 1  # (include/acattn.h: the row a large share of the lookups hit)
        L, H, chunks = p.L, p.H, _lib.EMBED_BWD_CHUNKS
        # the loss node's dense table gradient, if one was published in this walk: the rows are scattered into it and
        # the table gets no second gradient from here (no zero fill, no [N, H] add); else a zero-filled buffer of our own
        handed = ctx.state.take_table_grad(ctx.tick, table) if ctx.needs_input_grad[1] else None
        d_table = handed if handed is not None else (torch.zeros_like(table) if ctx.needs_input_grad[1] else None)
        want_pos = has_pos and ctx.needs_input_grad[2]
        pos_part = torch.empty(chunks, L, H, device=table.device, dtype=torch.float32) if want_pos else None
        want_gb = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        gb_part = torch.empty(chunks * L, 2, H, device=table.device, dtype=torch.float32) if want_gb else None
        _lib.check(lib.acattn_embed_layernorm_bwd(C.byref(p), _ptr(dy.contiguous()), _ptr(stats),
                                                  -1 if padding_idx is None else int(padding_idx), _ptr(d_table),
                                                  _ptr(pos_part), _ptr(gb_part), _stream()), "embed_layernorm_bwd")
        d_pos = None
        if want_pos:
            d_pos = torch.zeros_like(pos)
            d_pos[:L] = ops.sum_rows(pos_part, 0)
        dgamma = dbeta = None
        if want_gb:
            st = getattr(ctx, "state", None)
            gb = ops.sum_rows0(gb_part, st)
            dgamma, dbeta = gb[0], gb[1]
            if st is not None:
                st.watch(gb, dgamma if ctx.needs_input_grad[3] else None, dbeta if ctx.needs_input_grad[4] else None)
        return None, (None if handed is not None else d_table), d_pos, dgamma, dbeta, None, None, None, None, None, None, None, None


def embed_layer_norm(item_seq: torch.Tensor, item_embedding: torch.nn.Embedding,
                     position_embedding: Optional[torch.nn.Embedding], norm: torch.nn.LayerNorm, p_drop: float,
                     training: bool, keep: Optional[torch.Tensor] = None, return_nonzero: bool = False):
    """dropout(norm(item_embedding(item_seq) + position_embedding(arange(L))), p_drop, training) -> [B, L, H];
    with `return_nonzero` also `item_seq != 0` as bytes [B, L] (written by the same launch: the structured mask's
    key validity, abstract_recommender.py:137)."""
    p = p_drop if (training or keep is not None) else 0.0
    state = state_of(norm)
    seed = state.draw_seed() if (p > 0 and keep is None) else 0
    y, nonzero = _EmbedLayerNorm.apply(item_seq.contiguous(), item_embedding.weight,
                                       None if position_embedding is None else position_embedding.weight, norm.weight,
                                       norm.bias, norm.eps, p, keep, seed, state.seed_tensor if keep is None else None,
                                       item_embedding.padding_idx, state, getattr(item_embedding, "_acattn_hot_id", None))
    return (y, nonzero) if return_nonzero else y


def supported(hidden_size: int) -> bool:
    return hidden_size in (64, 128, 256)
