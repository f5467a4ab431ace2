"""y = x W^T + b for the tall-skinny activations of this model ([B*L, 64..256], B*L = 25,600 at the
benchmark shape) with a weight gradient that does not fall into a library pothole.

The forward and the input gradient are ordinary GEMMs (hipBLASLt via torch).  The weight gradient
dW[N,K] = sum_m g[m,n] x[m,k] is a GEMM whose reduction dimension is B*L: on MI355X / ROCm 7.2 hipBLASLt's
heuristic answers it with a stream-K kernel that takes 110-125 us, rocBLAS takes 25-50 us but is 20x
slower on the step's one large GEMM (tools/gemm_probe.py, profiles/).  dW and db therefore come from
acattn_linear_wgrad (csrc/acattn_linear.hip): one pass over x and dy feeding fp32 MFMAs, two launches per layer
instead of the five of a split-K built from library calls.  Parameters stay ordinary nn.Linear weights
(state-dict keys are load-bearing, recbole/trainer/trainer.py:672-683).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .state import DEFAULT as _DEFAULT_STATE, state_of


def _sum_rows(x, dim):
    from .ops import sum_rows  # late import: ops imports nothing from here, but keeps module load order simple
    return sum_rows(x, dim)


# Which backward pass of the two-pass trainer is running decides which parameter gradients are worth computing
# (recbole/trainer/trainer.py:672-684); the answer is per-model state captured by every node at forward time:
# state.StepState.

def _split(m: int) -> int:
    """Number of slabs: the largest divisor of m that is <= 128 and leaves slabs of >= 128 rows."""
    best = 1
    for s in range(2, 129):
        if m % s == 0 and m // s >= 128:
            best = s
    return best


class _SkinnyLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, is_attack, state):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.is_attack = is_attack
        ctx.state = state
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        g2 = g.reshape(-1, g.shape[-1])
        if ctx.needs_input_grad[0]:
            gx = (g2 @ weight).view_as(x)
        want_params = (not ctx.state.calibrated_pass_only) if ctx.is_attack else (not ctx.state.attack_pass_only)
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and want_params
        if ctx.needs_input_grad[1] and want_params:
            from .ops import linear_wgrad
            gw, gb = linear_wgrad(x.reshape(-1, x.shape[-1]), g2, want_b, ctx.state)
        elif want_b:
            gb = _sum_rows(g2, 0)
        return gx, gw, gb, None, None


def skinny_linear(x: torch.Tensor, layer: torch.nn.Linear) -> torch.Tensor:
    """layer(x) whose weight / bias gradients come from acattn_linear_wgrad.  CPU tensors (module construction,
    state-dict tests) take the stock path."""
    if not x.is_cuda or not torch.is_grad_enabled():
        return layer(x)
    return _SkinnyLinear.apply(x, layer.weight, layer.bias, getattr(layer, "_acattn_attack", False), state_of(layer))


class _Projections(torch.autograd.Function):
    """The six projections in front of the attention core as ONE autograd node:
        mq, mk, mv = query/key/value(x)                     recbole/model/layers.py:687-689
        qa, ka     = attack_query/key_transform(mq, mk)     layers.py:658-659
        gate       = gate(mq)                               layers.py:887 (combine_option 'gate')
    Forward: the same six GEMMs.  Backward: the cotangents that meet at mq, mk and x are accumulated by the GEMMs
    themselves (beta = 1) instead of by five separate elementwise adds per layer and pass, and the parameter
    gradients come from acattn_linear_wgrad.  Gradients reach exactly the leaves the six nn.Linear would feed."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, waq, baq, wak, bak, wg, bg, attack_upstream, state):
        mq, mk, mv = F.linear(x, wq, bq), F.linear(x, wk, bk), F.linear(x, wv, bv)
        qa, ka = F.linear(mq, waq, baq), F.linear(mk, wak, bak)
        gate = F.linear(mq, wg, bg) if wg is not None else None
        ctx.save_for_backward(x, mq, mk, wq, wk, wv, waq, wak, wg if wg is not None else x.new_empty(0))
        ctx.has_gate = wg is not None
        ctx.attack_upstream = attack_upstream
        ctx.state = state
        ctx.set_materialize_grads(False)  # unused outputs arrive as None in backward (handled there)
        return mq, mk, mv, qa, ka, gate

    @staticmethod
    def backward(ctx, dmq, dmk, dmv, dqa, dka, dgate):
        from .ops import linear_wgrad
        x, mq, mk, wq, wk, wv, waq, wak, wg = ctx.saved_tensors
        H = x.shape[-1]
        two = lambda t: None if t is None else t.reshape(-1, t.shape[-1])
        x2, mq2, mk2 = two(x), two(mq), two(mk)
        dmq, dmk, dmv, dqa, dka, dgate = (two(t) for t in (dmq, dmk, dmv, dqa, dka, dgate))

        def acc_(total, g, w):  # total += g @ w inside the GEMM (beta = 1, in place: no copy, no add kernel)
            if g is None:
                return total
            return g @ w if total is None else total.addmm_(g, w)

        grads = [None] * 12
        others = not ctx.state.attack_pass_only  # pass 2 keeps only the attack transforms (trainer.py:678-684)
        attack = not ctx.state.calibrated_pass_only  # pass 1 has them frozen (trainer.py:672-677)
        # In pass 2 a layer with no attack transform upstream (the first one) owes nobody an input gradient: the only
        # things left to compute are the two attack transforms' own gradients.
        need_dx = ctx.needs_input_grad[0] and (others or ctx.attack_upstream)
        # dmq / dmk are the attention node's freshly allocated dq / dk (mq and mk have no consumer outside this node
        # and the core), so accumulating into them in place touches nothing anyone else reads
        dx = dmq_t = dmk_t = None
        if need_dx or others:
            dmq_t = acc_(acc_(dmq, dqa, waq), dgate if ctx.has_gate else None, wg)
            dmk_t = acc_(dmk, dka, wak)
        if need_dx:
            dx = acc_(acc_(acc_(None, dmq_t, wq), dmk_t, wk), dmv, wv)
            dx = dx.view_as(x) if dx is not None else None
        jobs = []  # (slot of the weight gradient, input, cotangent, want bias): one grouped launch pair for all of them

        def params(slot, inp, g, want):
            if want and g is not None and (ctx.needs_input_grad[slot] or ctx.needs_input_grad[slot + 1]):
                jobs.append((slot, inp, g, ctx.needs_input_grad[slot + 1]))

        params(1, x2, dmq_t, others)
        params(3, x2, dmk_t, others)
        params(5, x2, dmv, others)
        params(7, mq2, dqa, attack)
        params(9, mk2, dka, attack)
        if ctx.has_gate:
            params(11, mq2, dgate, others)
        if jobs:
            from .ops import linear_wgrad_grouped
            for (slot, _, _, _), (gw, gb) in zip(jobs, linear_wgrad_grouped([(i, g, wb) for _, i, g, wb in jobs], ctx.state)):
                grads[slot - 1], grads[slot] = gw, gb
        return (dx, *grads, None, None)


# measurement switch (bench.py --projections library): route supported sizes through the hipBLASLt node as well
FUSED_PROJECTIONS = True
# measurement / test switch: let the fused launch hand affine planes and gate probabilities to the attention core
PRODUCER_EXTRAS = True


class _FusedProjections(torch.autograd.Function):
    """`_Projections` as ONE HIP launch each way (acattn_projections_fwd / _bwd, csrc/acattn_proj.hip; hidden 64):
    x is read once, mq / mk stay in registers between the product that makes them and the ones that consume them.
    The parameter gradients come from the same grouped acattn_linear_wgrad launch pair as in `_Projections`."""

    @staticmethod
    def _problem(x, wq, bq, wk, bk, wv, bv, waq, baq, wak, bak, wg, bg):
        from . import _lib
        from .ops import _ptr
        p = _lib.ProjProblem()
        p.rows, p.H, p.G = x.numel() // x.shape[-1], x.shape[-1], (wg.shape[0] if wg is not None else 0)
        p.x = _ptr(x)
        for name, t in (("wq", wq), ("bq", bq), ("wk", wk), ("bk", bk), ("wv", wv), ("bv", bv), ("waq", waq), ("baq", baq),
                        ("wak", wak), ("bak", bak), ("wg", wg), ("bg", bg)):
            setattr(p, name, _ptr(t))
        return p

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, waq, baq, wak, bak, wg, bg, attack_upstream, state, spatial=None):
        """`spatial` = (w_order, b_order, w_dist, b_dist, n_heads) or None.  With it (and a [B, L, H] input) the launch
        also writes what the attention core would otherwise re-derive per head / per query block: the rank-1 halves of
        the two spatial affines (acattn_problem.affine) and sigmoid(gate) in place of the gate logits
        (acattn_problem.gate_is_prob).  The gate output then HOLDS PROBABILITIES while the gradient that comes back for
        it is still the gradient of the logits (include/acattn.h: acattn_bwd_io.dgate_logits) -- an internal edge
        between this node and the attention node, not something a caller sees."""
        import ctypes as C
        from . import _lib
        from .ops import _ptr, _stream
        x = x.contiguous()
        params = [None if t is None else t.contiguous() for t in (wq, bq, wk, bk, wv, bv, waq, baq, wak, bak, wg, bg)]
        p = _FusedProjections._problem(x, *params)
        mq, mk, mv, qa, ka = (torch.empty_like(x) for _ in range(5))
        gate = x.new_empty(*x.shape[:-1], p.G) if wg is not None else None
        out = _lib.ProjOut()
        out.mq, out.mk, out.mv, out.qa, out.ka, out.gate = (_ptr(t) for t in (mq, mk, mv, qa, ka, gate))
        affine = None
        keep = []
        if spatial is not None and x.dim() == 3:
            w_order, b_order, w_dist, b_dist, n_heads = spatial
            B, L, H = x.shape
            affine = _affine_workspace(state, x.device, B, n_heads, L)
            keep = [t.detach().reshape(-1).contiguous() for t in (w_order, b_order, w_dist, b_dist)]
            p.w_order, p.b_order, p.w_dist, p.b_dist = (_ptr(t) for t in keep)
            p.n_heads, p.L = n_heads, L
            out.affine = _ptr(affine)
            out.gate_prob = 1 if gate is not None else 0
        _lib.check(_lib.load().acattn_projections_fwd(C.byref(p), C.byref(out), _stream()), "projections_fwd")
        ctx.save_for_backward(x, mq, mk, *(t if t is not None else x.new_empty(0) for t in params))
        ctx.has_gate = wg is not None
        ctx.attack_upstream = attack_upstream
        ctx.state = state
        ctx.set_materialize_grads(False)
        if affine is not None:
            ctx.mark_non_differentiable(affine)
        # x again, as an output of this node: the layer hands it to its tails as the residual (layers.py:683), so their
        # d_x arrives HERE and the backward launch starts dx from it (acattn_proj_bwd_io.dx_init) instead of autograd
        # adding the two [rows, H] gradients of x with an elementwise launch per layer and walk
        return mq, mk, mv, qa, ka, gate, x.view_as(x), affine

    @staticmethod
    def backward(ctx, dmq, dmk, dmv, dqa, dka, dgate, d_res=None, _d_affine=None):
        import ctypes as C
        from . import _lib
        from .ops import _ptr, _stream, linear_wgrad_grouped
        x, mq, mk = ctx.saved_tensors[:3]
        params = [t if t.numel() else None for t in ctx.saved_tensors[3:]]
        two = lambda t: None if t is None else t.reshape(-1, t.shape[-1])
        con = lambda t: None if t is None else t.contiguous()
        dmq, dmk, dmv, dqa, dka, dgate, d_res = (con(t) for t in (dmq, dmk, dmv, dqa, dka, dgate, d_res))
        if not ctx.has_gate:
            dgate = None
        others = not ctx.state.attack_pass_only  # pass 2 keeps only the attack transforms (trainer.py:678-684)
        attack = not ctx.state.calibrated_pass_only  # pass 1 has them frozen (trainer.py:672-677)
        need_dx = ctx.needs_input_grad[0] and (others or ctx.attack_upstream)
        dx = dmq_t = dmk_t = None
        if all(t is None for t in (dmq, dmk, dmv, dqa, dka, dgate)):  # only the residual path carries a gradient
            dx = d_res if need_dx else None
        elif need_dx or others:
            io = _lib.ProjBwdIO()
            io.dmq, io.dmk, io.dmv, io.dqa, io.dka, io.dgate = (_ptr(t) for t in (dmq, dmk, dmv, dqa, dka, dgate))
            io.dx_init = _ptr(d_res) if need_dx else None
            # dmq / dmk are the attention node's freshly allocated dq / dk: completing them in place touches nothing
            # anyone else reads
            dmq_t = dmq if dmq is not None else torch.empty_like(x)
            dmk_t = dmk if dmk is not None else torch.empty_like(x)
            dx = torch.empty_like(x) if need_dx else None
            io.dmq_total, io.dmk_total, io.dx = _ptr(dmq_t), _ptr(dmk_t), _ptr(dx)
            p = _FusedProjections._problem(x, *params)
            ws_bytes = int(_lib.load().acattn_projections_bwd_workspace_bytes(C.byref(p)))
            ws = torch.empty(ws_bytes // 4, device=x.device, dtype=torch.float32) if ws_bytes > 0 else None
            io.workspace = _ptr(ws)  # transposed weight copies at hidden 128 / 256
            _lib.check(_lib.load().acattn_projections_bwd(C.byref(p), C.byref(io), _stream()), "projections_bwd")
        grads = [None] * 12
        jobs = []

        def want(slot, inp, g, on):
            if on and g is not None and (ctx.needs_input_grad[slot] or ctx.needs_input_grad[slot + 1]):
                jobs.append((slot, two(inp), two(g), ctx.needs_input_grad[slot + 1]))

        want(1, x, dmq_t, others)
        want(3, x, dmk_t, others)
        want(5, x, dmv, others)
        want(7, mq, dqa, attack)
        want(9, mk, dka, attack)
        if ctx.has_gate:
            want(11, mq, dgate, others)
        if jobs:
            for (slot, _, _, _), (gw, gb) in zip(jobs, linear_wgrad_grouped([(i, g, wb) for _, i, g, wb in jobs], ctx.state)):
                grads[slot - 1], grads[slot] = gw, gb
        return (dx, *grads, None, None, None)


def _affine_workspace(state, device, B, n_heads, L):
    """[B, n_heads, 4, LP] zeros: the projections launch of a layer writes entries [0, L) of every plane, the attention
    launch right behind it on the same stream reads them, nobody else ever does (the backward recomputes from q, k and the
    parameters) -- so the layers of ONE model share a buffer per shape, and the padding entries [L, LP), which the kernels
    never write but do load, stay the zeros they were allocated as.  The buffer belongs to the model's StepState and is
    never evicted (a captured hipGraph holds its address; a process-global cache with eviction was a use-after-free
    under shape sweeps, and shared one buffer between models on different streams: ADVICE r3).  A module without a model
    state (the frozen default) gets a fresh buffer per call."""
    if getattr(state, "_frozen", True):
        return torch.zeros(B, n_heads, 4, 16 * ((L + 15) // 16), device=device, dtype=torch.float32)
    key = (str(device), B, n_heads, L)
    ws = state.affine_ws.get(key)
    if ws is None:
        ws = state.affine_ws[key] = torch.zeros(B, n_heads, 4, 16 * ((L + 15) // 16), device=device, dtype=torch.float32)
    return ws


def projections(x, query, key, value, attack_query, attack_key, gate=None, attack_upstream=True, spatial=None):
    """(mq, mk, mv, qa, ka, gate_logits or None, x_res, extras) of one encoder layer; see _Projections.
    `attack_upstream=False` tells the node that nothing that produced `x` holds attack transforms (the first encoder
    layer).  `x_res` is `x` for the residual connections of the layer (see _FusedProjections.forward; plain `x` on the
    other paths).  `spatial` = (w_order, b_order, w_dist, b_dist, n_heads): the single-launch path then also produces
    the affine planes and hands the gate over as probabilities; `extras` = {'affine': tensor, 'gate_is_prob': bool}
    says what it did (empty on the other paths) and goes to ops.calibrated_attention as keyword arguments."""
    node = _Projections if torch.is_grad_enabled() else None
    if x.is_cuda and FUSED_PROJECTIONS and x.dtype == torch.float32 and all(
            m.bias is not None for m in (query, key, value, attack_query, attack_key) + ((gate,) if gate is not None else ())):
        from . import _lib
        if _lib.load().acattn_projections_supported(x.shape[-1], gate.out_features if gate is not None else 0):
            node = _FusedProjections  # the single launch serves evaluation (no_grad) as well
    if not x.is_cuda or node is None:
        mq, mk, mv = query(x), key(x), value(x)
        return mq, mk, mv, attack_query(mq), attack_key(mk), (gate(mq) if gate is not None else None), x, {}
    args = (x, query.weight, query.bias, key.weight, key.bias, value.weight, value.bias,
            attack_query.weight, attack_query.bias, attack_key.weight, attack_key.bias,
            gate.weight if gate is not None else None, gate.bias if gate is not None else None,
            attack_upstream, state_of(query))
    if node is _FusedProjections:
        if not PRODUCER_EXTRAS or (spatial is not None and any(t is None for t in spatial[:4])):
            spatial = None
        *out, affine = node.apply(*args, spatial)
        extras = {} if affine is None else {"affine": affine, "gate_is_prob": gate is not None}
        return (*out, extras)
    out = node.apply(*args)
    return (*out, x, {})


class _FullSortScores(torch.autograd.Function):
    """scores = output @ E^T over the whole catalogue (acsasrec.py:118-119), with the input gradient
    d_output = d_scores @ E computed as a batched split-K product: its reduction runs over all N items
    (100k), the same library pothole as the weight gradients above (242 us -> see profiles/)."""

    @staticmethod
    def forward(ctx, output, table, state):
        ctx.save_for_backward(output, table)
        ctx.state = state
        return output @ table.t()

    @staticmethod
    def backward(ctx, g):
        output, table = ctx.saved_tensors
        g_out = g_tab = None
        if ctx.needs_input_grad[0]:
            n = table.shape[0]
            # split-K only for a SKINNY left operand (B = 512 rows against 100k items: the library pothole this was written
            # for).  With AcBERT4Rec's ~20k masked rows the product is an ordinary GEMM, and the slabs -- 113 of them at
            # 20,001 table rows: a 2.4 GB intermediate and two 0.49 ms reductions per step at configs[4] -- only cost [r4]
            s = _split(n) if g.shape[0] <= 4096 else 1
            if s > 1:
                b = g.shape[0]
                g_out = _sum_rows(torch.bmm(g.view(b, s, n // s).transpose(0, 1), table.view(s, n // s, -1)), 0)
            else:
                g_out = g @ table
        if ctx.needs_input_grad[1] and not ctx.state.attack_pass_only:
            g_tab = g.t() @ output
        return g_out, g_tab, None


def full_sort_scores(output: torch.Tensor, table: torch.Tensor, state=_DEFAULT_STATE) -> torch.Tensor:
    if not output.is_cuda or not torch.is_grad_enabled():
        return torch.matmul(output, table.transpose(0, 1))
    return _FullSortScores.apply(output, table, state)


class _EmbeddingLookup(torch.autograd.Function):
    """item_embedding(item_seq) (acsasrec.py:87) whose backward scatters with float atomics (index_add_) instead
    of torch's sort + segmented reduce (0.41 ms per step at B*L = 25,600 lookups; the atomic form is bounded by
    6.5 MB of atomic traffic).  Row `padding_idx` receives no gradient, like nn.Embedding."""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx, state):
        ctx.save_for_backward(idx)
        ctx.shape = weight.shape
        ctx.padding_idx = padding_idx
        ctx.state = state
        return F.embedding(idx, weight)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        if ctx.state.attack_pass_only:
            return None, None, None, None
        n, h = ctx.shape
        flat = idx.reshape(-1)
        if ctx.padding_idx is None:
            gw = g.new_zeros(n, h)
            gw.index_add_(0, flat, g.reshape(-1, h))
            return None, gw, None, None
        # about half of all lookups hit the padding row: thousands of atomic adds on ONE 256-byte row serialise
        # (measured 300 us).  Their sum is discarded anyway, so they are scattered over 4096 scratch rows instead.
        scratch = 4096
        gw = g.new_zeros(n + scratch, h)
        spread = n + (torch.arange(flat.numel(), device=flat.device) & (scratch - 1))
        gw.index_add_(0, torch.where(flat == ctx.padding_idx, spread, flat), g.reshape(-1, h))
        return None, gw[:n], None, None


def embedding_lookup(idx: torch.Tensor, emb: torch.nn.Embedding) -> torch.Tensor:
    if not idx.is_cuda or not torch.is_grad_enabled():
        return emb(idx)
    return _EmbeddingLookup.apply(idx, emb.weight, emb.padding_idx, state_of(emb))
