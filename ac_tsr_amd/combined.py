"""Single-pass combined backward: both cotangent sets of the two-pass protocol through ONE walk of the autograd graph.

The reference trainer walks one graph twice (recbole/trainer/trainer.py:672-686): `calibrated_loss.backward(retain_graph=True)`
with the attack transforms frozen, then `attacked_loss.backward()` with only the attack transforms live.  Both walks visit
(almost) every node: the calibrated loss reaches the attack transforms' INPUTS through the frozen transforms, the attacked
loss reaches the lower layers' attack transforms through everything.  The two gradients cannot be summed -- every leaf keeps
the gradient of ONE of the losses -- so a single walk has to carry two cotangent SETS:

  set "cal"  d calibrated_loss / d .   travels as autograd's own gradients (the walk is `calibrated_loss.backward`);
  set "att"  d attacked_loss / d .     travels beside it in `CombinedWalk.side`, keyed by (node, output index).

Every custom autograd node of this package is wrapped (`instrument`): when its backward is called by the engine with the cal
cotangents, it picks up the att cotangents deposited for its outputs by the nodes downstream (autograd's topological order
guarantees they have all run), evaluates its backward for both sets -- non-attack parameters take the cal set's gradients
through autograd, attack parameters the att set's directly -- and deposits the att set's input gradients on its producers.
Plain torch nodes between the custom ones (views, squeezes, the stack / mean of a penalty) are linear: the att cotangent is
pushed through them on the spot by calling the node.  The part of the graph only the attacked loss reaches (its loss node,
the last layer's attacked tail) is evaluated by hand before the engine starts (`prefix`).

What one walk shares: the engine's traversal, the launch stream (one hipGraph region), and -- where a node implements
`backward_pair` -- the recomputation both sets need (the attention core rebuilds its probability tiles and regenerates its
random numbers once per set otherwise).  Opt-in (`AttackSASRecTrainer(combined_backward=True)`); the default stays the
reference's two walks.  Losses, gradients and the update are the same in both modes (tests/test_hip_backward.py)."""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, Iterable, Optional

import torch
from torch.autograd.function import BackwardCFunction

_INSTRUMENTED = set()


def _is_accumulate(fn) -> bool:
    return type(fn).__name__ == "AccumulateGrad"


def instrument(*classes) -> None:
    """Wrap forward (records which arguments are tensors: autograd's edge list has one entry per TENSOR argument) and
    backward (the dual evaluation above) of custom autograd Functions.  Idempotent; a node outside a combined walk pays one
    attribute lookup."""
    for cls in classes:
        if cls in _INSTRUMENTED:
            continue
        _INSTRUMENTED.add(cls)
        fwd, bwd = cls.forward, cls.backward

        def forward(ctx, *args, _fwd=fwd):
            ctx._acattn_tensor_args = tuple(isinstance(a, torch.Tensor) for a in args)
            # (the node's materialize_grads flag cannot be read back from Python: record what forward sets)
            ctx._acattn_materialize = True
            setter = ctx.set_materialize_grads

            def record(value, _ctx=ctx, _setter=setter):
                _ctx._acattn_materialize = bool(value)
                _setter(value)

            ctx.set_materialize_grads = record
            try:
                return _fwd(ctx, *args)
            finally:  # back to the class method (no reference cycle through the closure)
                try:
                    del ctx.set_materialize_grads
                except AttributeError:
                    pass

        def backward(ctx, *grads, _bwd=bwd):
            walk = getattr(getattr(ctx, "state", None), "combined", None)
            if walk is None or walk.manual:
                return _bwd(ctx, *grads)
            return walk.node(ctx, _bwd, grads)

        cls.forward = staticmethod(forward)
        cls.backward = staticmethod(backward)


class CombinedWalk:
    def __init__(self, state, attack_params: Iterable[torch.nn.Parameter]):
        self.state = state
        self.attack_ids = {id(p) for p in attack_params}
        self.side: Dict[int, Dict[int, torch.Tensor]] = {}  # id(node) -> {output index: att cotangent}
        self.reach = set()   # ids of the nodes the engine will visit (reachable from the calibrated loss)
        self.keep = []       # node objects of `reach` (ids stay valid while the walk lasts)
        self.manual = False  # a node is being evaluated by hand: the wrapper passes through
        self.stats = {"dual_nodes": 0, "pair_nodes": 0, "prefix_nodes": 0, "pushed_through": 0}

    # ---- modes -----------------------------------------------------------------------------------------------------------
    @contextmanager
    def _attack(self):
        """pass_mode = 'attack' for one node evaluation (NOT StepState.attack_pass(): that opens a new walk and would drop
        the table gradient the calibrated set's loss node published for the embedding node)."""
        st = self.state
        prev = st.pass_mode
        st.pass_mode = "attack"
        try:
            yield
        finally:
            st.pass_mode = prev

    @contextmanager
    def _by_hand(self):
        prev = self.manual
        self.manual = True
        try:
            yield
        finally:
            self.manual = prev

    # ---- graph helpers ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _walk_graph(root):
        seen, order, stack = set(), [], [root]
        while stack:
            fn = stack.pop()
            if fn is None or id(fn) in seen:
                continue
            seen.add(id(fn))
            order.append(fn)
            for nxt, _ in fn.next_functions:
                if nxt is not None:
                    stack.append(nxt)
        return order

    @staticmethod
    def _n_outputs(fn) -> int:
        meta = getattr(fn, "_input_metadata", None)
        return len(meta) if meta is not None else 1

    @staticmethod
    def _zeros_for(fn, idx):
        m = fn._input_metadata[idx]
        return torch.zeros(tuple(m.shape), dtype=m.dtype, device=m.device)

    def _edges(self, fn, n_results: int):
        """next_functions aligned with the node's backward results: custom nodes return one result per forward ARGUMENT,
        the edge list has one entry per TENSOR argument."""
        mask = getattr(fn, "_acattn_tensor_args", None)
        if mask is None:
            return list(fn.next_functions)
        edges, it = [], iter(fn.next_functions)
        for is_tensor in mask:
            edges.append(next(it) if is_tensor else (None, 0))
        return edges

    def _evaluate(self, fn, grads, fill=True):
        """One node's backward, by hand, for the att set."""
        with torch.no_grad(), self._by_hand(), self._attack():
            if isinstance(fn, BackwardCFunction):
                if getattr(fn, "_acattn_materialize", True):
                    grads = [g if g is not None else self._zeros_for(fn, i) for i, g in enumerate(grads)]
                out = fn.apply(*grads)
            else:
                if fill and any(g is None for g in grads):
                    grads = [g if g is not None else self._zeros_for(fn, i) for i, g in enumerate(grads)]
                out = fn(*grads)
        return out if isinstance(out, (tuple, list)) else (out,)

    # ---- the att set's transport -----------------------------------------------------------------------------------------
    def deposit(self, fn, nr: int, g: Optional[torch.Tensor]) -> None:
        if fn is None or g is None:
            return
        if _is_accumulate(fn):
            p = fn.variable
            if id(p) in self.attack_ids:  # trainer.py:678-684: only the attack transforms take the attacked loss's gradient
                g = g.detach()
                if p.grad is None:
                    p.grad = g if g.shape == p.shape else g.reshape(p.shape)
                else:
                    p.grad.add_(g.reshape(p.shape))
            return
        if isinstance(fn, BackwardCFunction) and type(fn)._forward_cls in _INSTRUMENTED and id(fn) in self.reach:
            slot = self.side.setdefault(id(fn), {})
            slot[nr] = g if nr not in slot else slot[nr] + g
            return
        # a plain torch node (or a custom node outside the engine's walk that the prefix did not schedule): linear, so the
        # cotangent goes through it now
        # (a node with several forward outputs -- native_layer_norm's mean / rstd, a split -- gets this output's cotangent and
        # nothing for the others: by linearity two cotangents arriving at different outputs may go through one at a time)
        self.stats["pushed_through"] += 1
        n_out = max(self._n_outputs(fn), nr + 1)
        out = self._evaluate(fn, [g if i == nr else None for i in range(n_out)], fill=False)
        for (nxt, n2), g2 in zip(self._edges(fn, len(out)), out):
            self.deposit(nxt, n2, g2)

    # ---- called by the wrapped backward of a node the engine visits ------------------------------------------------------
    def node(self, ctx, bwd, g_cal):
        g_att = self.side.pop(id(ctx), None)
        pair = getattr(type(ctx)._forward_cls, "backward_pair", None)
        both = None
        if g_att and pair is not None:
            # one evaluation for both sets where the node can share the recomputation between them (None: not this time)
            both = pair(ctx, list(g_cal), [g_att.get(i) for i in range(len(g_cal))])
        if both is not None:
            res_cal, res_att = both
            self.stats["pair_nodes"] += 1
        else:
            res_cal = bwd(ctx, *g_cal)
            res_att = None
            if g_att:
                att = [g_att.get(i) for i in range(len(g_cal))]
                if getattr(ctx, "_acattn_materialize", True):
                    att = [a if a is not None else (torch.zeros_like(c) if c is not None else self._zeros_for(ctx, i))
                           for i, (a, c) in enumerate(zip(att, g_cal))]
                with torch.no_grad(), self._attack():
                    res_att = bwd(ctx, *att)
                self.stats["dual_nodes"] += 1
        if res_att is not None:
            if not isinstance(res_att, (tuple, list)):
                res_att = (res_att,)
            for (nxt, nr), g in zip(self._edges(ctx, len(res_att)), res_att):
                self.deposit(nxt, nr, g)
        return res_cal

    # ---- the part of the graph only the attacked loss reaches ------------------------------------------------------------
    def prefix(self, attacked_loss: Optional[torch.Tensor], calibrated_loss: torch.Tensor) -> None:
        self.keep = self._walk_graph(calibrated_loss.grad_fn)
        self.reach = {id(fn) for fn in self.keep}
        if attacked_loss is None or attacked_loss.grad_fn is None:
            return
        root = attacked_loss.grad_fn
        one = torch.ones_like(attacked_loss)
        if id(root) in self.reach:
            self.deposit(root, 0, one)
            return
        sub = [fn for fn in self._walk_graph_until(root)]
        ids = {id(fn) for fn in sub}
        deps = {id(fn): 0 for fn in sub}
        for fn in sub:
            for nxt, _ in fn.next_functions:
                if nxt is not None and id(nxt) in ids:
                    deps[id(nxt)] += 1
        buf = {id(root): {0: one}}
        ready = [root]
        while ready:
            fn = ready.pop()
            got = buf.pop(id(fn), {})
            n_out = max(self._n_outputs(fn), (max(got) + 1) if got else 1)
            out = self._evaluate(fn, [got.get(i) for i in range(n_out)]) if got else ()
            self.stats["prefix_nodes"] += 1
            edges = self._edges(fn, len(out)) if out else [(nxt, nr) for nxt, nr in fn.next_functions]
            outs = list(out) + [None] * (len(edges) - len(out))
            for (nxt, nr), g in zip(edges, outs):
                if nxt is None:
                    continue
                if id(nxt) in ids:
                    if g is not None:
                        slot = buf.setdefault(id(nxt), {})
                        slot[nr] = g if nr not in slot else slot[nr] + g
                    deps[id(nxt)] -= 1
                    if deps[id(nxt)] == 0:
                        ready.append(nxt)
                else:
                    self.deposit(nxt, nr, g)

    def _walk_graph_until(self, root):
        """Nodes reachable from `root` without passing through a node the engine will visit (leaves excluded)."""
        seen, order, stack = set(), [], [root]
        while stack:
            fn = stack.pop()
            if fn is None or id(fn) in seen or id(fn) in self.reach or _is_accumulate(fn):
                continue
            seen.add(id(fn))
            order.append(fn)
            for nxt, _ in fn.next_functions:
                stack.append(nxt)
        return order

    def finish(self) -> None:
        left = {k: list(v) for k, v in self.side.items()}
        self.side.clear()
        self.keep = []
        if left:
            raise RuntimeError(f"combined backward: att cotangents were deposited on nodes the walk never visited: {left}")


def instrument_package() -> None:
    """Every custom autograd node of the package (idempotent)."""
    from . import ce, fused_embed, fused_ln, linear, ops, tail
    instrument(ce._FullSortCE, ce._FullSortCEDir, ce._FullSortCEMean, ce._AttackedLoss, ce._AttackedLossRows,
               fused_embed._EmbedLayerNorm, fused_ln._DropoutAddLayerNorm, linear._SkinnyLinear, linear._Projections,
               linear._FusedProjections, linear._FullSortScores, linear._EmbeddingLookup, ops._CalibratedAttention,
               tail._FusedLayerTail, tail._LayerTail)
    # (ops._MaskPenalty and parallel._GlobalNorm hold no parameters and no pass-dependent state: the att cotangent is pushed
    # through them like through a plain torch node)
