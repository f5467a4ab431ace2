"""Batch data-parallelism: one process per MI355X, RCCL (torch.distributed "nccl" on ROCm) over xGMI.

The reference has no distributed code.  Sequences are independent through the whole forward, so the
batch dimension is split across ranks with replicated parameters and the gradients are averaged
between the backward passes and the optimizer step (recbole/trainer/trainer.py:684 -> :687).  All
parameter gradients live in one flat fp32 buffer (`.grad` tensors are views into it), so the exchange
is a few large collectives instead of ~60 small ones: on MI355X's point-to-point xGMI fabric few, large
messages are what keeps the 7 links busy; the item-embedding gradient ([n_items, H], dense because CE
scores every item) dominates it.

Overlap.  Under the two-pass protocol every parameter that is not an attack transform has its FINAL
gradient when pass 1 (calibrated loss) ends: pass 2 only reaches the attack transforms
(trainer.py:678-684).  `reduce_early()` therefore starts the all-reduce of that part (item table
included: 99.9 % of the bytes) right after pass 1, on the communication stream, and pass 2 runs under
it; `all_reduce()` after pass 2 sends the attack transforms' 66 KB and waits for both.

Not identical to one process on the global batch in ONE term, by default: the attacked loss carries
`torch.norm(1 - M)` over the whole (local) batch (acsasrec.py:131-137), and a norm is not additive.
Averaging per-rank gradients weights the mask penalty of the attack transforms by 1/sqrt(N) relative
to a single process seeing all N shards (the cross-entropy terms are exact means).  This is the
DDP-conventional "per-rank penalty" (SURVEY.md section 8e); every other gradient equals the
single-process one.  The exact alternative is `global_mask_penalty` below (model config key
`dp_mask_penalty: 'global'`): the per-layer sums of squares are all-reduced before the square root -- one
scalar per layer -- and N ranks then take exactly the step of one process on the concatenated batch.

Collective form.  `collective="all_reduce"` (default) is one all-reduce per bucket; `"reduce_scatter"`
is a reduce-scatter followed by an all-gather of the same bucket.  On MI355X's fully connected xGMI
(7 links per GPU) a ring all-reduce is bound by ONE link, while a reduce-scatter / all-gather pair
addresses all 7 peers at once; which one RCCL's own all-reduce picks is its
business, the flag makes the choice explicit and measurable (bench.py --dp-collective).

Works with backend "gloo" on CPU tensors as well (world_size-2 tests run without a GPU).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise from torchrun's environment.  Returns (rank, world_size, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(n: int, rank: int, world: int) -> slice:
    """Contiguous, near-equal slice of a global batch of n sequences owned by `rank`."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


class GradSynchronizer:
    """Flat gradient buffer + bucketed average all-reduce.

    `bucket_bytes` splits the flat buffer into a few large collectives so the first ones can start
    while later ones are still queued; the default keeps the (dominant) embedding gradient in buckets of
    32 MiB, far above the latency-bound regime of an xGMI ring step.

    Two ways to fill the buffer:
      * accumulate_in_place=False (default): `.grad` starts as None every step, autograd STORES each first gradient
        (no zero-fill of 27 MB, no read-modify-write per parameter) and `pack()` gathers them into the flat buffer
        with one multi-tensor copy; afterwards `.grad` of every parameter is a view of the (reduced) flat buffer,
        which is what the optimizer reads.  Inside a captured graph the trainer records `pack()` with the step.
      * accumulate_in_place=True: `.grad` are permanent views of the flat buffer and autograd accumulates into
        them (one small add kernel per parameter and step; supports gradient accumulation over micro-batches).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None,
                 accumulate_in_place: bool = False, late: Iterable[torch.nn.Parameter] = (),
                 collective: str = "all_reduce"):
        """`late`: the parameters whose gradient is only complete after the LAST backward pass (the attack transforms
        under the two-pass trainer).  They sit at the end of the flat buffer; everything in front of them can be
        reduced early (`reduce_early`).  `collective`: "all_reduce" or "reduce_scatter" (+ all-gather), see above."""
        assert collective in ("all_reduce", "reduce_scatter"), collective
        late_ids = {id(p) for p in late}
        every = [p for p in params]
        self.params: List[torch.nn.Parameter] = [p for p in every if id(p) not in late_ids] + \
                                                [p for p in every if id(p) in late_ids]
        self.n_early = sum(1 for p in every if id(p) not in late_ids) if late_ids else 0
        assert self.params, "no parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = collective
        # the early and the late section are each padded to a multiple of the world size: a reduce-scatter hands every
        # rank an equal share of a bucket (a few zeros at most; an all-reduce does not care)
        pad = lambda n: -(-n // self.world) * self.world
        early_raw = sum(p.numel() for p in self.params[:self.n_early])
        late_raw = sum(p.numel() for p in self.params[self.n_early:])
        self.early_numel = pad(early_raw)
        total = self.early_numel + pad(late_raw)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.views: List[torch.Tensor] = []
        off = 0
        for k, p in enumerate(self.params):
            if k == self.n_early:
                off = self.early_numel
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.in_place = accumulate_in_place
        if self.in_place:
            self.attach()
        per = pad(max(1, bucket_bytes // self.flat.element_size()))
        # buckets never straddle the early / late boundary
        self.early_buckets = [self.flat[s:min(s + per, self.early_numel)] for s in range(0, self.early_numel, per)]
        self.late_buckets = [self.flat[s:min(s + per, total)] for s in range(self.early_numel, total, per)]
        self.buckets = self.early_buckets + self.late_buckets
        self._early_works = None
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"  # RCCL averages in the collective

    @classmethod
    def for_two_pass_model(cls, model: torch.nn.Module, **kw) -> "GradSynchronizer":
        """Synchronizer for a model trained by AttackSASRecTrainer: the attack transforms (selected by name like the
        reference does, recbole/trainer/trainer.py:672-683) are the `late` parameters."""
        from .trainer import is_attack_param
        return cls(model.parameters(), late=[p for n, p in model.named_parameters() if is_attack_param(n)], **kw)

    def attach(self) -> None:
        """Make every parameter's .grad the corresponding view of the flat buffer."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero_grad(self) -> None:
        if self.in_place:
            self.flat.zero_()
            if any(p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(self.params, self.views)):
                self.attach()  # something replaced .grad (e.g. set_to_none)
        else:
            for p in self.params:
                p.grad = None

    def pack(self, part: str = "all") -> None:
        """Gather the parameters' gradients into the flat buffer (no-op for gradients that already live there).
        part: "all", "early" (everything but the `late` parameters) or "late"."""
        lo, hi = {"all": (0, len(self.params)), "early": (0, self.n_early), "late": (self.n_early, len(self.params))}[part]
        src, dst = [], []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)

    def _start(self, buckets):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.collective == "all_reduce":
            return [dist.all_reduce(b, op=op, group=self.group, async_op=True) for b in buckets]
        # reduce-scatter into this rank's share of the bucket (in place), then all-gather the shares back
        works = []
        rank = dist.get_rank(self.group)
        for b in buckets:
            n = b.numel() // self.world
            mine = b[rank * n:(rank + 1) * n]
            works.append(_reduce_scatter(mine, b, op, self.group, self.world, rank))
            works.append(dist.all_gather_into_tensor(b, mine, group=self.group, async_op=True))
        return works

    def reduce_early(self, packed: bool = False) -> None:
        """Call between the backward passes: starts the all-reduce of every gradient that is already final (all but
        the `late` parameters).  The collective is enqueued behind the work issued so far and runs on the
        communication stream while the caller keeps launching the last backward pass.  `packed=True`: the early part
        of the flat buffer was already filled (a captured graph ended with pack("early"))."""
        if self.n_early == 0:
            return
        if not self.in_place and not packed:
            self.pack("early")
        self._early_works = self._start(self.early_buckets) if self.world > 1 else []

    def all_reduce(self, packed: bool = False) -> None:
        """After this call every parameter's .grad is a view of the flat buffer holding the rank-averaged gradient.
        Reduces what `reduce_early` has not already sent and waits for both."""
        early_done = self._early_works is not None
        if not self.in_place:
            if not packed:
                self.pack("late" if early_done else "all")
            self.attach()
        works = self._early_works or []
        self._early_works = None
        if self.world > 1:
            works = works + self._start(self.late_buckets if early_done else self.buckets)
        for w in works:
            w.wait()
        if self.world > 1 and not self._avg:
            self.flat.div_(self.world)


class _Done:
    def wait(self):
        return True


def _reduce_scatter(out: torch.Tensor, whole: torch.Tensor, op, group, world: int, rank: int):
    """reduce_scatter_tensor where the backend has it (RCCL); over gloo, which has not, one reduce per share rooted at
    the share's owner -- the same result, used by the CPU tests."""
    if dist.get_backend(group) != "gloo":
        return dist.reduce_scatter_tensor(out, whole, op=op, group=group, async_op=True)
    n = whole.numel() // world
    for r in range(world):
        dist.reduce(whole[r * n:(r + 1) * n], dst=dist.get_global_rank(group, r) if group is not None else r, op=op, group=group)
    return _Done()


class _GlobalNorm(torch.autograd.Function):
    """|| 1 - M ||_2 over the GLOBAL batch (torch.norm(1 - attack_mask, p=2), acsasrec.py:135, with M the
    concatenation of every rank's mask): sum of squares locally, all-reduce of that scalar, square root.  Backward:
    d M = d * world * (M - 1) / norm -- the factor `world` because the synchronizer AVERAGES the ranks' gradients,
    and the single process' gradient is their SUM for this term (its shards' contributions add up under one root)."""

    @staticmethod
    def forward(ctx, m, group):
        s = ((1.0 - m) ** 2).sum()
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        norm = s.sqrt()
        ctx.save_for_backward(m, norm)
        ctx.world = dist.get_world_size(group)
        return norm

    @staticmethod
    def backward(ctx, d):
        m, norm = ctx.saved_tensors
        return d * ctx.world * (m - 1.0) / norm, None


def global_mask_penalty(attack_mask: torch.Tensor, group=None) -> torch.Tensor:
    """The mask penalty of one layer as ONE process on the concatenated batch would compute it; a plain local norm
    when no process group is up (or it has one member)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return torch.norm(1 - attack_mask, p=2)
    return _GlobalNorm.apply(attack_mask, group)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
