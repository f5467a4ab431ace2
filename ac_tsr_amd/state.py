"""Per-model step state read by the custom autograd nodes.

The two-pass trainer (recbole/trainer/trainer.py:672-686) differentiates two losses over ONE graph: pass 1 with
the attack transforms frozen, pass 2 with only the attack transforms live.  A Python autograd.Function cannot see
which of its outputs the engine needs, so the nodes of this package ask "which pass is running?" to skip gradients
that would be computed and dropped.  That answer, the device-side step counter added to the kernels' random seeds
under hipGraph replay, and the dead-work switch of DESIGN.md section 5 are state of ONE model being trained by ONE
trainer: they live in a `StepState` object that the model attaches to its modules and every node captures at
forward time (`ctx.state`).  Two models / trainers in one process therefore cannot disturb each other, and a module
used on its own (no model around it) sees `DEFAULT`, which is frozen: every gradient is computed.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Optional

import torch


import os as _os
_CONFLICT = object()  # StepState.table_grad: two nodes published in one pass
_NO_HANDOVER = _os.environ.get("ACATTN_NO_HANDOVER") == "1"  # measurement / bisection hook
_NO_DEFER = _os.environ.get("ACATTN_NO_DEFER") == "1"  # measurement / bisection hook: every wgrad reduction where it is due


class StepState:
    __slots__ = ("pass_mode", "seed_tensor", "prune_dead_work", "seed_salt", "tick", "table_grad", "grad_home", "_frozen",
                 "_home_claimed", "affine_ws", "combined", "defer_reductions", "_deferred", "_flush_leaves")

    def __init__(self, frozen: bool = False):
        object.__setattr__(self, "_frozen", False)
        self.pass_mode: Optional[str] = None  # None | "calibrated" | "attack"
        self.seed_tensor: Optional[torch.Tensor] = None  # device int64[1], added to every counter-RNG seed
        # False = the reference's full schedule: every layer's attacked tail, the last layer's tails on all positions,
        # every input gradient (same results; bench.py measures both)
        self.prune_dead_work: bool = True
        # XORed into every kernel seed drawn for this model: data-parallel ranks seed torch's CPU generator alike, the
        # salt (set from the rank by the trainer) keeps row b of every shard from drawing the same noise and dropout
        self.seed_salt: int = 0
        # The item table receives two gradients in one backward walk: the full-sort cross-entropy's dense [N, H] one and
        # the embedding lookup's scattered rows.  Left to autograd they meet in a 25.6 MB add (plus the zero fill of the
        # scatter target).  Instead the cross-entropy node PUBLISHES its gradient tensor here and the embedding node,
        # which autograd runs later (it is upstream), scatters straight into it and returns nothing for the table.
        # `tick` orders forwards: the embedding node only takes a gradient published by a loss node built after it.
        # The hand-over is only sound when ONE node produces a table gradient in the walk: with two, autograd sums their
        # tensors as soon as the second arrives and nobody reads the published one any more -- the scattered rows would
        # be lost.  One producer is what the trainer's pass 1 guarantees (calibrated_loss.backward(inputs=...) walks one
        # loss: recbole/trainer/trainer.py:672-677), so nodes publish only inside `calibrated_pass()`, a second
        # publication in the same pass withdraws the first (the embedding node then returns its own gradient and
        # autograd adds), and any other walk -- sum(losses).backward(), torch.autograd.grad on a user loss -- takes the
        # plain path.
        self.tick: int = 0
        self.table_grad = None  # (tick of the publishing node's forward, the table, its gradient tensor) | _CONFLICT
        self._home_claimed = None  # data_ptrs whose flat-buffer home was handed out in this pass (see grad_buffer_for)
        # Data parallelism keeps every gradient in one flat buffer (parallel.GradSynchronizer).  A node that produces a
        # parameter's WHOLE gradient in one launch (the cross-entropy's dense table gradient: 99.9 % of the model's
        # bytes) can write it there directly instead of into a fresh tensor that is copied over afterwards:
        # {parameter data_ptr: its view of the flat buffer}, set by the trainer when it has a synchronizer.
        self.grad_home = None
        # The affine planes the projections launch hands to the attention launch behind it (linear._affine_workspace): one
        # zero-initialised buffer per (device, B, heads, L), owned HERE -- by the model's state -- and never evicted: a
        # captured hipGraph bakes the buffer's address into its projections and attention nodes, so the buffer has to live
        # at least as long as any graph captured from the model, and two models of the same shape must not share one.
        self.affine_ws = {}
        # combined.CombinedWalk while a single-pass combined backward runs (trainer, opt-in), else None
        self.combined = None
        # [r4] Deferred weight-gradient reductions (ops.linear_wgrad_grouped): inside a trainer's backward walk the nodes
        # launch only stage 1 of acattn_linear_wgrad (partial sums) and hand autograd an unwritten tensor; the trainer
        # calls flush_deferred() when the walk is over and ONE stage-2 launch writes every such tensor (six launches of
        # ~6 us per step become two).  Sound only because (a) nothing reads a parameter gradient before the walk ends,
        # (b) autograd ADOPTS the tensor it is handed for a leaf whose .grad is None (AccumulateGrad's use_count test:
        # the queue keeps the storage alive, not the tensor) -- flush_deferred verifies (b) for every job and raises
        # otherwise.  Set by the trainer (AttackSASRecTrainer), which opens its passes with the walk's leaves
        # (calibrated_pass(leaves) / attack_pass(leaves)): the pass context itself flushes at its end.  Off anywhere else.
        self.defer_reductions = False
        self._deferred = []
        self._flush_leaves = None
        object.__setattr__(self, "_frozen", frozen)

    def __setattr__(self, name, value):
        if getattr(self, "_frozen", False):  # (unpickling / copy.copy restore the slots before `_frozen` exists)
            raise AttributeError("state.DEFAULT is read-only: attach a StepState to the model (StepState().attach(model))")
        object.__setattr__(self, name, value)

    def __getstate__(self):
        # what survives pickling (torch.save(model), mp.spawn): the settings, not the per-walk hand-over state nor the
        # views into a synchronizer's flat buffer
        return {"pass_mode": self.pass_mode, "prune_dead_work": self.prune_dead_work, "seed_salt": self.seed_salt,
                "seed_tensor": self.seed_tensor, "tick": self.tick, "frozen": self._frozen}

    def __setstate__(self, st):
        object.__setattr__(self, "_frozen", False)
        self.pass_mode, self.prune_dead_work, self.seed_salt = st["pass_mode"], st["prune_dead_work"], st["seed_salt"]
        self.seed_tensor, self.tick = st["seed_tensor"], st["tick"]
        self.table_grad = self.grad_home = self._home_claimed = None
        self.affine_ws = {}
        self.combined = None
        self.defer_reductions = False
        self._deferred = []
        self._flush_leaves = None
        object.__setattr__(self, "_frozen", st["frozen"])

    def __copy__(self):
        new = StepState()
        new.__setstate__(self.__getstate__())
        return new

    def __deepcopy__(self, memo):
        new = StepState()
        new.pass_mode, new.prune_dead_work, new.seed_salt = self.pass_mode, self.prune_dead_work, self.seed_salt
        new.seed_tensor = None if self.seed_tensor is None else self.seed_tensor.clone()
        new.grad_home = None
        new.affine_ws = {}
        new.combined = None
        new.defer_reductions = False
        memo[id(self)] = new
        return new

    # pass 2 (attacked loss): only the attack transforms accumulate (trainer.py:678-684)
    @property
    def attack_pass_only(self) -> bool:
        return self.pass_mode == "attack"

    # pass 1 (calibrated loss): the attack transforms are frozen (trainer.py:672-677)
    @property
    def calibrated_pass_only(self) -> bool:
        return self.pass_mode == "calibrated"

    @contextmanager
    def _pass(self, mode, leaves=None):
        prev, prev_leaves = self.pass_mode, self._flush_leaves
        self.pass_mode = mode
        self.table_grad = None  # a new walk: nothing published, no flat-buffer home handed out yet
        self._home_claimed = None
        # reductions are deferred only inside a pass that was opened WITH the leaves of its walk: that pass flushes them when
        # it ends, so no caller can be left holding unwritten gradients
        self._flush_leaves = list(leaves) if (leaves is not None and self.defer_reductions) else None
        self._deferred = []
        ok = False
        try:
            yield self
            ok = True
        finally:
            try:
                if ok and self._flush_leaves is not None:
                    self.flush_deferred(self._flush_leaves)
            finally:
                self._deferred = []
                self.pass_mode, self._flush_leaves = prev, prev_leaves

    def calibrated_pass(self, leaves=None):
        """Inside: layers tagged `_acattn_attack = True` produce no parameter gradients.  `leaves`: the parameters the walk
        inside accumulates into (backward(inputs=leaves)) -- given, parameter-gradient reductions may be deferred to the
        end of the pass (defer_reductions)."""
        return self._pass("calibrated", leaves)

    def attack_pass(self, leaves=None):
        """Inside: only layers tagged `_acattn_attack = True` produce parameter gradients.  `leaves` as in calibrated_pass."""
        return self._pass("attack", leaves)

    def next_tick(self) -> int:
        """Forward-order stamp of an autograd node (-1 on the frozen default: no hand-over there)."""
        if self._frozen:
            return -1
        self.tick += 1
        return self.tick

    def publish_table_grad(self, tick: int, table: torch.Tensor, grad: Optional[torch.Tensor]) -> None:
        if self._frozen or tick < 0 or grad is None or self.pass_mode != "calibrated":
            return
        self.table_grad = (tick, table, grad) if self.table_grad is None else _CONFLICT

    def take_table_grad(self, tick: int, table: torch.Tensor) -> Optional[torch.Tensor]:
        """The gradient tensor THE loss node downstream of the caller published for `table` in this backward walk."""
        if self._frozen or self.table_grad is None or tick < 0 or _NO_HANDOVER:
            return None
        if self.table_grad is _CONFLICT:
            self.table_grad = None
            return None
        t, tab, grad = self.table_grad
        if t > tick and tab.data_ptr() == table.data_ptr() and grad.shape == table.shape:
            self.table_grad = None
            return grad
        return None

    def grad_buffer_for(self, param: torch.Tensor) -> Optional[torch.Tensor]:
        """The flat-buffer view a full gradient of `param` may be written into, or None (no synchronizer, or the parameter
        already holds a gradient that autograd would ADD this one to)."""
        if self._frozen or not self.grad_home or param.grad is not None or self.pass_mode != "calibrated":
            return None
        # once per pass and parameter: a second producer of the same walk would overwrite the first one's bytes (both see
        # param.grad is None until the walk ends)
        claimed = self._home_claimed if self._home_claimed is not None else set()
        if param.data_ptr() in claimed:
            return None
        claimed.add(param.data_ptr())
        self._home_claimed = claimed
        home = self.grad_home.get(param.data_ptr())
        # a fresh tensor object on the same memory: autograd keeps a gradient without copying it only if nobody else
        # holds the tensor object (AccumulateGrad's use_count test), and the synchronizer holds `home`
        return None if home is None else home.view_as(home)

    # ---- deferred weight-gradient reductions (see __init__) ------------------------------------------------------------
    def deferring(self) -> bool:
        """May a node of the walk that is running leave its stage-2 reduction to flush_deferred()?"""
        return (not self._frozen and self.defer_reductions and self.combined is None and self._flush_leaves is not None
                and self.pass_mode in ("calibrated", "attack") and not _NO_DEFER)

    def defer(self, job) -> None:
        self._deferred.append(job)

    def defer_sum(self, x: torch.Tensor, out: torch.Tensor) -> None:
        """out[c] = sum_r x[r, c], written by flush_deferred().  The caller hands autograd VIEWS of `out` and names them with
        watch() right behind: those are what has to be adopted."""
        self._deferred.append({"sum_x": x.data_ptr(), "sum_out": out.data_ptr(), "R": x.shape[0], "C": x.numel() // x.shape[0],
                               "watch": [], "keep": (x.untyped_storage(), out.untyped_storage())})

    def watch(self, out: torch.Tensor, *views) -> None:
        """The gradient tensors (views of `out`) that autograd must adopt -- if `out` is the last defer_sum's (ops.sum_rows0
        may have summed on the spot)."""
        if self._deferred and self._deferred[-1].get("sum_out") == out.data_ptr():
            self._deferred[-1]["watch"] += [v.data_ptr() for v in views if v is not None]

    def flush_deferred(self, params) -> None:
        """End of a backward walk over the leaves `params`: one stage-2 launch per ACATTN_WGRAD_MAX_REDUCE queued items.
        Every queued destination must by now BE the .grad of one of `params` (same memory): anything else means autograd
        copied the unwritten tensor instead of adopting it, and the step would train on garbage -- raise."""
        jobs, self._deferred = self._deferred, []
        if not jobs:
            return
        import ctypes as C
        from . import _lib
        owned = {p.grad.data_ptr() for p in params if p.grad is not None}
        for j in jobs:
            for k, ptr in enumerate((j["dw"], j["db"]) if "dw" in j else j["watch"]):
                if ptr is not None and ptr not in owned:
                    what = (f"weight gradient [{j['N']}, {j['K']}]" if "dw" in j else f"row sum [{j['R']}, {j['C']}]") + f", output {k}"
                    raise RuntimeError(f"a deferred parameter gradient ({what}) was not adopted by autograd as the parameter's "
                                       ".grad (ACATTN_NO_DEFER=1 switches the deferral off)")
        lib = _lib.load()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wj = [j for j in jobs if "dw" in j]
        sj = [j for j in jobs if "sum_x" in j]
        while wj or sj:  # one launch per (ACATTN_WGRAD_MAX_REDUCE weight gradients + ACATTN_SUMROWS_MAX_DEFER row sums)
            w, wj = wj[:_lib.WGRAD_MAX_REDUCE], wj[_lib.WGRAD_MAX_REDUCE:]
            r, sj = sj[:_lib.SUMROWS_MAX_DEFER], sj[_lib.SUMROWS_MAX_DEFER:]
            arr = lambda js, key: (C.c_void_p * max(1, len(js)))(*(j[key] for j in js))
            ints = lambda js, key: (C.c_int32 * max(1, len(js)))(*(j[key] for j in js))
            _lib.check(lib.acattn_linear_wgrad_reduce_many(arr(w, "part_w"), arr(w, "part_b"), ints(w, "K"), ints(w, "N"), ints(w, "P"),
                                                           arr(w, "dw"), arr(w, "db"), len(w), arr(r, "sum_x"), arr(r, "sum_out"),
                                                           ints(r, "R"), ints(r, "C"), len(r), stream), "linear_wgrad_reduce_many")

    def draw_seed(self) -> int:
        """One 63-bit seed for the library's counter RNG from torch's CPU generator (reproducible under
        torch.manual_seed, no device sync), salted per rank."""
        return (int(torch.empty((), dtype=torch.int64).random_().item()) ^ self.seed_salt) & 0x7FFFFFFFFFFFFFFF

    def attach(self, module: torch.nn.Module) -> "StepState":
        for m in module.modules():
            m.__dict__["_step_state"] = self
        return self


DEFAULT = StepState(frozen=True)


def state_of(module) -> StepState:
    """The StepState a module was attached to, or the frozen default."""
    return module.__dict__.get("_step_state", DEFAULT) if module is not None else DEFAULT
