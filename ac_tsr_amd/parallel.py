"""Batch data-parallelism: one process per MI355X, RCCL (torch.distributed "nccl" on ROCm) over xGMI.

The reference has no distributed code.  Sequences are independent through the whole forward, so the
batch dimension is split across ranks with replicated parameters and ONE exchange step per
optimizer step: an average all-reduce of all gradients after the second backward pass
(recbole/trainer/trainer.py:684 -> :687).  All parameter gradients live in one flat fp32 buffer
(`.grad` tensors are views into it), so the exchange is a single large collective instead of ~60
small ones: on MI355X's point-to-point xGMI fabric few, large messages are what keeps the 7 links
busy; the item-embedding gradient ([n_items, H], dense because CE scores every item) dominates it.

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
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        assert self.params, "no parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)  # autograd accumulates in place into this view
            off += n
        per = max(1, bucket_bytes // self.flat.element_size())
        self.buckets = [self.flat[s:min(s + per, total)] for s in range(0, total, per)]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def zero_grad(self) -> None:
        self.flat.zero_()
        for p in self.params:  # re-attach views if something replaced .grad (e.g. set_to_none)
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                    p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size():
                self._reattach()
                break

    def _reattach(self) -> None:
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def all_reduce(self) -> None:
        if self.world <= 1:
            return
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
        for w in works:
            w.wait()
        self.flat.div_(self.world)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
