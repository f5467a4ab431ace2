"""CPU suite: the data-parallel path (ac_tsr_amd/parallel.py) with world_size 2 over gloo.

The HIP kernels cannot run here, so the replicas train a small torch module; what is under test is the
part that is new relative to the (single-device) reference: batch sharding, the flat gradient buffer,
the bucketed average all-reduce and parameter broadcast -- i.e. that N ranks on B/N sequences each take
the same optimizer step as one process on B sequences."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ac_tsr_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Embedding(50, 8), torch.nn.Flatten(), torch.nn.Linear(8 * 5, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 50))


def _batch():
    g = torch.Generator().manual_seed(9)
    return torch.randint(0, 50, (12, 5), generator=g), torch.randint(0, 50, (12,), generator=g)


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    if rank == 1:  # replicas must not depend on identical seeding: perturb, then broadcast from rank 0
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    parallel.broadcast_parameters(model, src=0)
    sync = parallel.GradSynchronizer(model.parameters(), bucket_bytes=1024)  # several buckets
    assert len(sync.buckets) > 1
    x, y = _batch()
    sl = parallel.shard_batch(x.shape[0], rank, world)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        sync.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x[sl]), y[sl])
        loss.backward()
        sync.all_reduce()
        opt.step()
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_one_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = torch.load(tmp_path / "rank0.pt")
    s1 = torch.load(tmp_path / "rank1.pt")
    # single-process reference on the full batch (equal shards -> mean of shard means == full mean)
    model = _model()
    x, y = _batch()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(model(x), y).backward()
        opt.step()
    ref = model.state_dict()
    for k in ref:
        assert torch.equal(s0[k], s1[k]), k  # replicas stay bit-identical
        assert (s0[k] - ref[k]).abs().max() <= 1e-5, k


def test_shard_batch_covers_everything():
    for n in (1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                sl = parallel.shard_batch(n, r, world)
                seen += list(range(n))[sl]
            assert seen == list(range(n))


def test_pack_mode_stores_gradients_and_hands_the_optimizer_flat_views():
    model = _model()
    sync = parallel.GradSynchronizer(model.parameters())  # default: accumulate_in_place=False
    x, y = _batch()
    ref = _model()
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    for _ in range(2):  # twice: a stale flat buffer must not leak into the next step
        sync.zero_grad()
        assert all(p.grad is None for p in model.parameters())
        torch.nn.functional.cross_entropy(model(x), y).backward()
        sync.all_reduce()
        lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + sync.flat.numel() * 4
        for p, q in zip(model.parameters(), ref.parameters()):
            assert lo <= p.grad.data_ptr() < hi and torch.equal(p.grad, q.grad)
    # a parameter that received no gradient reads as zeros
    sync.zero_grad()
    first = next(model.parameters())
    sync.flat.fill_(7.0)
    sync.all_reduce()
    assert first.grad.abs().sum() == 0


def test_grad_views_survive_zero_grad_and_accumulate_in_place():
    model = _model()
    sync = parallel.GradSynchronizer(model.parameters(), accumulate_in_place=True)
    x, y = _batch()
    sync.zero_grad()
    torch.nn.functional.cross_entropy(model(x), y).backward()
    ptrs = [p.grad.data_ptr() for p in model.parameters()]
    lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + sync.flat.numel() * 4
    assert all(lo <= q < hi for q in ptrs)
    assert sync.flat.abs().sum() > 0
    g1 = sync.flat.clone()
    torch.nn.functional.cross_entropy(model(x), y).backward()  # second backward accumulates into the same buffer
    assert torch.allclose(sync.flat, 2 * g1, atol=1e-6)
    sync.zero_grad()
    assert sync.flat.abs().sum() == 0 and [p.grad.data_ptr() for p in model.parameters()] == ptrs


def test_two_pass_trainer_protocol_on_cpu_module():
    """The trainer's requires_grad toggling (trainer.py:672-686) routed through AttackSASRecTrainer.train_step:
    attack parameters receive only the attacked loss' gradient, all others only the calibrated loss'."""
    import ac_tsr_amd as A

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.body = torch.nn.Linear(4, 4)
            self.attack_query_transform = torch.nn.Linear(4, 4)

        def calculate_loss(self, x):
            h = self.body(x)
            a = self.attack_query_transform(h)
            return -(a ** 2).mean() + h.mean(), ((h + a.detach() * 0 + a) ** 2).mean()

    m = Toy()
    x = torch.randn(6, 4)
    att, cal = m.calculate_loss(x)
    g_att = torch.autograd.grad(att, list(m.attack_query_transform.parameters()), retain_graph=True)
    g_cal = torch.autograd.grad(cal, list(m.body.parameters()))
    tr = A.AttackSASRecTrainer(A.DictConfig(learner="sgd", learning_rate=0.0), m)
    tr.train_step(x)
    for p, g in zip(m.attack_query_transform.parameters(), g_att):
        assert torch.allclose(p.grad, g, atol=1e-6)
    for p, g in zip(m.body.parameters(), g_cal):
        assert torch.allclose(p.grad, g, atol=1e-6)


class _TwoPassToy(torch.nn.Module):
    """A CPU stand-in with the trainer-facing surface of the models: `calculate_loss` returns (attacked, calibrated),
    and the attack transforms are found by name (recbole/trainer/trainer.py:672-683)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.item_embedding = torch.nn.Embedding(40, 8)
        self.body = torch.nn.Linear(8, 8)
        self.attack_query_transform = torch.nn.Linear(8, 8)
        self.attack_key_transform = torch.nn.Linear(8, 8)

    def calculate_loss(self, batch):
        x, y = batch
        h = torch.tanh(self.body(self.item_embedding(x).mean(1)))
        a = self.attack_query_transform(h) * torch.sigmoid(self.attack_key_transform(h))
        logits = (h + 0.1 * a) @ self.item_embedding.weight.t()
        calibrated = torch.nn.functional.cross_entropy(logits, y)
        attacked = -torch.nn.functional.cross_entropy((h + a) @ self.item_embedding.weight.t(), y)
        return attacked, calibrated


def _toy_batch():
    g = torch.Generator().manual_seed(4)
    return torch.randint(0, 40, (16, 6), generator=g), torch.randint(0, 40, (16,), generator=g)


def _trainer_worker(rank, world, port, out):
    import ac_tsr_amd as A
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    model = _TwoPassToy()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(0.5)
    parallel.broadcast_parameters(model, src=0)
    sync = parallel.GradSynchronizer.for_two_pass_model(model, bucket_bytes=512)
    assert sync.n_early == 3 and len(sync.early_buckets) > 1 and len(sync.late_buckets) >= 1
    # the attack transforms sit at the end of the flat buffer, everything else in front
    assert all(any(p is q for q in model.attack_query_transform.parameters()) or
               any(p is q for q in model.attack_key_transform.parameters()) for p in sync.params[sync.n_early:])
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="adam", learning_rate=1e-2), model, grad_sync=sync)
    assert (trainer.state.seed_salt != 0) == (rank != 0)  # ranks draw different in-kernel seeds
    x, y = _toy_batch()
    sl = parallel.shard_batch(x.shape[0], rank, world)
    for _ in range(3):
        trainer.train_step((x[sl], y[sl]))
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out, f"trainer_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_pass_trainer_with_synchronizer_on_two_ranks(tmp_path):
    """AttackSASRecTrainer + GradSynchronizer together over gloo, world size 2: early reduce between the passes, late
    reduce after them, one Adam step -- replicas stay identical and equal one process on the whole batch (both
    losses are means over equal shards, so the averaged gradients are the full-batch gradients)."""
    import ac_tsr_amd as A
    port = _free_port()
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = torch.load(tmp_path / "trainer_rank0.pt")
    s1 = torch.load(tmp_path / "trainer_rank1.pt")
    model = _TwoPassToy()
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="adam", learning_rate=1e-2), model)
    for _ in range(3):
        trainer.train_step(_toy_batch())
    ref = model.state_dict()
    for k in ref:
        assert torch.equal(s0[k], s1[k]), k
        assert (s0[k] - ref[k]).abs().max() <= 1e-5, k



class _PenaltyToy(_TwoPassToy):
    """_TwoPassToy whose attacked loss also carries the mask penalty of the models: a 2-norm over a tensor with one
    slice per sequence of the batch (torch.norm(1 - attack_mask), acsasrec.py:131-137) -- the one term of the step that
    is not a mean over sequences."""
    penalty = "local"

    def calculate_loss(self, batch):
        x, y = batch
        h = torch.tanh(self.body(self.item_embedding(x).mean(1)))
        a = self.attack_query_transform(h) * torch.sigmoid(self.attack_key_transform(h))
        m = torch.softmax(a, dim=-1)  # an "attack mask" [B, 8]
        logits = (h + 0.1 * a) @ self.item_embedding.weight.t()
        calibrated = torch.nn.functional.cross_entropy(logits, y)
        attacked = -torch.nn.functional.cross_entropy((h + a) @ self.item_embedding.weight.t(), y)
        pen = parallel.global_mask_penalty(m) if self.penalty == "global" else torch.norm(1 - m, p=2)
        return attacked + 0.3 * pen, calibrated


def _penalty_worker(rank, world, port, out, penalty, collective):
    import ac_tsr_amd as A
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    model = _PenaltyToy()
    model.penalty = penalty
    parallel.broadcast_parameters(model, src=0)
    sync = parallel.GradSynchronizer.for_two_pass_model(model, bucket_bytes=512, collective=collective)
    assert sync.flat.numel() % world == 0 and all(b.numel() % world == 0 for b in sync.buckets)
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="adam", learning_rate=1e-2), model, grad_sync=sync)
    x, y = _toy_batch()
    sl = parallel.shard_batch(x.shape[0], rank, world)
    for _ in range(3):
        trainer.train_step((x[sl], y[sl]))
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out, f"pen_{penalty}_{collective}_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["all_reduce", "reduce_scatter"])
def test_global_mask_penalty_makes_two_ranks_equal_one_process(tmp_path, collective):
    """The mask penalty is a norm over the WHOLE batch (acsasrec.py:131-137).  With the per-layer sum of squares
    all-reduced before the square root (parallel.global_mask_penalty, config `dp_mask_penalty: 'global'`) two ranks on
    half the batch each take exactly the step of one process on the concatenated batch; with the per-rank norm (the
    default) they do not -- and the reduce-scatter + all-gather form of the exchange moves the same numbers."""
    import ac_tsr_amd as A
    model = _PenaltyToy()
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner="adam", learning_rate=1e-2), model)
    for _ in range(3):
        trainer.train_step(_toy_batch())
    ref = model.state_dict()
    diffs = {}
    for penalty in ("global", "local"):
        port = _free_port()
        mp.spawn(_penalty_worker, args=(2, port, str(tmp_path), penalty, collective), nprocs=2, join=True)
        s0 = torch.load(tmp_path / f"pen_{penalty}_{collective}_rank0.pt")
        s1 = torch.load(tmp_path / f"pen_{penalty}_{collective}_rank1.pt")
        for k in ref:
            assert torch.equal(s0[k], s1[k]), (penalty, k)  # replicas stay bit-identical either way
        diffs[penalty] = max((s0[k] - ref[k]).abs().max().item() for k in ref)
    assert diffs["global"] <= 1e-5, diffs
    assert diffs["local"] > 10 * max(diffs["global"], 1e-7), diffs  # the per-rank norm is a different objective


def test_step_state_survives_pickling_and_copy():
    """Every module of a model carries the StepState: torch.save(model) / mp.spawn must not trip over it."""
    import copy
    import io
    import pickle
    from ac_tsr_amd.state import DEFAULT, StepState
    st = StepState()
    st.seed_salt, st.prune_dead_work = 12345, False
    lin = torch.nn.Linear(3, 3)
    st.attach(lin)
    for clone in (pickle.loads(pickle.dumps(st)), copy.copy(st), copy.deepcopy(st)):
        assert (clone.seed_salt, clone.prune_dead_work, clone.table_grad, clone.grad_home) == (12345, False, None, None)
        clone.seed_salt = 1  # not frozen
    buf = io.BytesIO()
    torch.save(lin, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert back.__dict__["_step_state"].seed_salt == 12345
    frozen = pickle.loads(pickle.dumps(DEFAULT))
    with pytest.raises(AttributeError):
        frozen.seed_salt = 3
