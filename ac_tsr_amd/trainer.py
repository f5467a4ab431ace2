"""The adversarial two-pass training step of AC-SASRec, plus batch data-parallelism.

Mirrors AttackSASRecTrainer._train_epoch (recbole/trainer/trainer.py:631-693): one forward,
`calibrated_loss.backward(retain_graph=True)` with the attack transforms frozen, then
`attacked_loss.backward()` with ONLY the attack transforms live, then one optimizer step over all
parameters.  The reference is single-device (recbole/config/configurator.py:344-348); with
`parallel.GradSynchronizer` the same step runs as one process per GPU with the batch split across
ranks and one RCCL all-reduce of the flat gradient buffer between the second backward and the
optimizer step.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
from torch import optim

from .state import StepState


def is_attack_param(name: str) -> bool:
    """The trainer selects the attack parameters by substring (trainer.py:672-683): names are load-bearing."""
    return 'attack_key_transform' in name or 'attack_query_transform' in name


class AttackSASRecTrainer:
    """Minimal trainer: optimizer construction (trainer.py:590-615: Adam by default) and the epoch loop."""

    def __init__(self, config, model, grad_sync=None, combined_backward: Optional[bool] = None):
        """`combined_backward` (or config key of that name; default False = the reference's protocol): ONE walk of the
        autograd graph carries the cotangents of both losses (ac_tsr_amd/combined.py) instead of two walks of the same
        graph (trainer.py:672-686).  Same losses, same gradients on every parameter, same update."""
        self.config = config
        self.model = model
        if combined_backward is None:
            try:
                combined_backward = bool(config['combined_backward']) if config is not None else False
            except KeyError:
                combined_backward = False
        self.combined_backward = bool(combined_backward)
        if self.combined_backward:
            from .combined import instrument_package
            instrument_package()  # before any forward whose graph will be walked
        self.learner = (config['learner'] or 'adam') if config is not None else 'adam'
        self.learning_rate = (config['learning_rate'] or 1e-3) if config is not None else 1e-3
        self.weight_decay = (config['weight_decay'] or 0.0) if config is not None else 0.0
        self.device = next(model.parameters()).device
        self.grad_sync = grad_sync
        self._graph2 = None
        self._graph_opt = None
        # the step state the model's autograd nodes read (which backward pass is running, replay seed counter);
        # owned by the model so that two trainers / models in one process stay independent
        self.state = getattr(model, 'step_state', None) or StepState().attach(model)
        # weight-gradient reductions of a walk in one launch at its end (state.py): needs every leaf's .grad to START the walk
        # as None, i.e. not the synchronizer's accumulate-in-place mode
        self.state.defer_reductions = grad_sync is None or not grad_sync.in_place
        self.optimizer = self._build_optimizer()
        self._graph = None
        for name, module in model.named_modules():  # lets the linear layers skip discarded gradients in pass 2
            if isinstance(module, torch.nn.Linear):
                module._acattn_attack = is_attack_param(name)
        self._attack = [p for n, p in model.named_parameters() if is_attack_param(n)]
        self._others = [p for n, p in model.named_parameters() if not is_attack_param(n)]
        if grad_sync is not None:
            # whole-parameter gradients may be written straight into the flat buffer (StepState.grad_home)
            self.state.grad_home = {p.data_ptr(): v for p, v in zip(grad_sync.params, grad_sync.views)}
            import torch.distributed as dist
            if dist.is_initialized():  # ranks seed torch's generator alike: keep their in-kernel draws apart
                self.state.seed_salt = (dist.get_rank(grad_sync.group) * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF

    def _build_optimizer(self):
        params = self.model.parameters()
        learner = self.learner.lower()
        if learner == 'adam':
            # capturable: the step counter lives on the device, so the whole step can sit inside a hipGraph.
            # fused: one multi-tensor kernel for all ~60 parameters; the foreach implementation spends ~100 tiny
            # per-parameter kernels per step on the bias corrections of its device-side step counters
            # (0.55 ms of a 3.9 ms step, profiles/).  Same update rule (torch.optim.Adam, trainer.py:590-615).
            on_gpu = self.device.type == 'cuda'
            if on_gpu:
                # torch.optim.Adam whose update is ONE launch for all parameters (ac_tsr_amd/optim.py, acattn_adam_step:
                # torch's fused implementation takes three, 74 us of the step); same state, same arithmetic
                from .optim import Adam
                return Adam(params, lr=self.learning_rate, weight_decay=self.weight_decay, capturable=True, fused=True)
            return optim.Adam(params, lr=self.learning_rate, weight_decay=self.weight_decay)
        if learner == 'sgd':
            return optim.SGD(params, lr=self.learning_rate, weight_decay=self.weight_decay)
        if learner == 'adagrad':
            return optim.Adagrad(params, lr=self.learning_rate, weight_decay=self.weight_decay)
        if learner == 'rmsprop':
            return optim.RMSprop(params, lr=self.learning_rate, weight_decay=self.weight_decay)
        return optim.Adam(params, lr=self.learning_rate)

    def _check_nan(self, loss):
        if torch.isnan(loss):
            raise ValueError('Training loss is nan')  # trainer.py:763-765

    def _pass_one(self, interaction, check_nan: bool = False):
        """Forward + backward pass 1 of one batch (trainer.py:660-677).  Returns the two losses."""
        if self._seed_t is not None and not self._fused_inputs:
            self._seed_t += 1  # fresh in-kernel randomness on every (replayed) step
        if self.grad_sync is not None:
            self.grad_sync.zero_grad()
        else:
            # None instead of zeros: the first gradient that reaches a leaf is then stored, not added to a zero
            # buffer (one fill + one add kernel less per parameter and step)
            self.optimizer.zero_grad(set_to_none=True)
        attacked_loss, calibrated_loss = self.model.calculate_loss(interaction)
        if check_nan:
            if attacked_loss is not None:
                self._check_nan(attacked_loss)
            self._check_nan(calibrated_loss)
        # The reference freezes the attack transforms for pass 1 and everything else for pass 2 by toggling
        # requires_grad, then walks the WHOLE graph twice (frozen leaves just drop what reaches them).
        # `backward(inputs=...)` accumulates into exactly the same leaves with the same values, and lets
        # autograd skip the branches that only feed frozen leaves (weight-gradient GEMMs, embedding scatter).
        with self.state.calibrated_pass(self._others):  # (the pass ends with the walk's parameter-gradient reductions, one launch: state.py)
            calibrated_loss.backward(self._root(calibrated_loss), retain_graph=attacked_loss is not None, inputs=self._others)
        return attacked_loss, calibrated_loss

    def _pass_two(self, attacked_loss):
        """Backward pass 2 (trainer.py:678-684): only the attack transforms accumulate."""
        if attacked_loss is not None:
            with self.state.attack_pass(self._attack):
                attacked_loss.backward(self._root(attacked_loss), inputs=self._attack)

    def _forward(self, interaction, check_nan: bool = False):
        if self._seed_t is not None and not self._fused_inputs:
            self._seed_t += 1
        if self.grad_sync is not None:
            self.grad_sync.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        attacked_loss, calibrated_loss = self.model.calculate_loss(interaction)
        if check_nan:
            if attacked_loss is not None:
                self._check_nan(attacked_loss)
            self._check_nan(calibrated_loss)
        return attacked_loss, calibrated_loss

    def _combined_pass(self, interaction, check_nan: bool = False):
        """Forward + ONE backward walk for both losses (combined.py).  The walk is `calibrated_loss.backward` over the
        non-attack leaves -- the engine's own gradients are the calibrated set -- with the attacked set travelling beside
        it; the attack transforms receive the attacked set's gradients directly."""
        attacked_loss, calibrated_loss = self._forward(interaction, check_nan)
        self.combined_backward_walk(attacked_loss, calibrated_loss)
        return attacked_loss, calibrated_loss

    def combined_backward_walk(self, attacked_loss, calibrated_loss):
        """The single walk itself, for losses of a forward that ran AFTER this trainer was built with
        combined_backward=True (the nodes are instrumented then).  Accumulates into .grad like the two walks do."""
        from .combined import CombinedWalk
        if not self.combined_backward:
            raise RuntimeError("build the trainer with combined_backward=True before the forward whose graph is walked")
        walk = CombinedWalk(self.state, self._attack)
        with self.state.calibrated_pass():
            self.state.combined = walk
            try:
                walk.prefix(attacked_loss, calibrated_loss)
                calibrated_loss.backward(self._root(calibrated_loss), inputs=self._others)
                walk.finish()
            finally:
                self.state.combined = None
        self.last_walk_stats = walk.stats

    def _eager_step(self, interaction, check_nan: bool = False):
        if self.combined_backward:
            attacked_loss, calibrated_loss = self._combined_pass(interaction, check_nan)
            if self.grad_sync is not None:
                self.grad_sync.all_reduce()  # one walk: every gradient is final at the same time, one exchange
            self.optimizer.step()
            return attacked_loss, calibrated_loss
        return self._two_pass_step(interaction, check_nan)

    def _two_pass_step(self, interaction, check_nan: bool = False):
        """One batch of trainer.py:660-687.  With a gradient synchronizer the all-reduce of everything but the attack
        transforms (the item table: 99.9 % of the bytes) starts between the passes and pass 2 runs under it."""
        attacked_loss, calibrated_loss = self._pass_one(interaction, check_nan)
        if self.grad_sync is not None:
            self.grad_sync.reduce_early()
        self._pass_two(attacked_loss)
        if self.grad_sync is not None:
            self.grad_sync.all_reduce()
        self.optimizer.step()
        return attacked_loss, calibrated_loss

    _seed_t = None
    _fused_inputs = False  # graph mode: the replay counter and the batch copies are ONE launch in front of the replay
    _one = None            # the root cotangent of both backward walks (a constant: no fill launch per walk)

    def _root(self, loss):
        if self._one is None or self._one.device != loss.device:
            self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        return self._one

    def _stage_inputs(self, interaction) -> bool:
        """Graph mode: the batch into the static buffers, the replay counter + 1 and the read positions
        (item_length - 1) in one launch (acattn_step_inputs).  False = this batch cannot go that way (the caller copies)."""
        import ctypes as C
        from . import _lib
        keys = [k for k in self._static_in if not k.startswith("_acattn_")]
        if len(keys) > 6 or any(not (interaction[k].is_cuda and interaction[k].is_contiguous() and interaction[k].dtype == self._static_in[k].dtype)
                                for k in keys):
            return False
        n = len(keys)
        src = (C.c_void_p * n)(*(interaction[k].data_ptr() for k in keys))
        dst = (C.c_void_p * n)(*(self._static_in[k].data_ptr() for k in keys))
        nbytes = (C.c_int64 * n)(*(self._static_in[k].numel() * self._static_in[k].element_size() for k in keys))
        last = self._static_in.get("_acattn_last_row")
        length = interaction[self.model.ITEM_SEQ_LEN] if last is not None else None
        _lib.check(_lib.load().acattn_step_inputs(src, dst, nbytes, n, C.c_void_p(self._seed_t.data_ptr()),
                                                  None if length is None else C.c_void_p(length.data_ptr()),
                                                  None if length is None else C.c_void_p(last.data_ptr()),
                                                  0 if length is None else length.numel(),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "step_inputs")
        return True

    def enable_graph(self, example_interaction, warmup: int = 3, debug_dump: Optional[str] = None):
        """Capture the training step into hipGraphs (about 350 short kernels per step: eager launches are host-bound).
        Inputs are copied into static buffers before each replay; the in-kernel RNG adds a device-side step counter to
        its seeds so every replay draws fresh noise / dropout.  Without a gradient synchronizer the whole step incl.
        the optimizer is ONE graph.  With one, the step is TWO graphs sharing a memory pool -- forward + pass 1 +
        packing of the early gradients, then pass 2 + packing of the attack transforms' -- and the collectives and
        the optimizer step are issued eagerly: the early all-reduce between the two replays (it overlaps pass 2), the
        rest after.  `debug_dump`: path of a .dot file that receives the captured graph (hipGraphDebugDotPrint)."""
        assert self.device.type == 'cuda'
        assert warmup >= 1, "at least one eager step must precede the capture (it creates the optimizer's state)"
        for m in self.model.modules():
            # 'annealing' advances a host-side step counter on every forward (recbole/model/layers.py:890-891) and the
            # rate travels to the kernel by value: a replayed graph would keep the rate of the capture forever
            if getattr(m, 'combine_option', None) == 'annealing':
                raise ValueError("combine_option='annealing' cannot run from a captured graph: the anneal rate is "
                                 "recomputed on the host for every forward (layers.py:890-891); train it eagerly")
        if any(isinstance(g.get('lr'), float) and g.get('initial_lr') is not None for g in self.optimizer.param_groups):
            # an LR scheduler registers `initial_lr` in every group; the float rate is frozen into the captured launch
            raise ValueError("an LR scheduler is attached to the optimizer: a captured step replays the learning rate of "
                             "the capture (ac_tsr_amd/optim.py); train eagerly or give the optimizer a tensor lr")
        if getattr(self.model, 'dp_mask_penalty', 'local') == 'global' and self.grad_sync is not None and self.grad_sync.world > 1:
            raise ValueError("dp_mask_penalty='global' all-reduces a scalar per layer INSIDE the forward (parallel.py); "
                             "that collective is not captured here: train it eagerly")
        self._static_in = {k: v.clone() for k, v in example_interaction.items()}
        self._seed_t = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.state.seed_tensor = self._seed_t
        len_key = getattr(self.model, "ITEM_SEQ_LEN", None)
        if (len_key in self._static_in and self._static_in[len_key].dtype == torch.int64
                and type(self.model).__name__ == "ACSASRec" and all(v.is_cuda for v in self._static_in.values())):
            # the position the model reads (item_length - 1, abstract_recommender.py:130-134), formed by the input launch
            self._static_in["_acattn_last_row"] = self._static_in[len_key] - 1
        self._fused_inputs = all(v.is_cuda and v.is_contiguous() for v in self._static_in.values()) and len(self._static_in) <= 6
        self.model.train()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if self._fused_inputs:
                    self._seed_t += 1
                self._eager_step(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph(keep_graph=True) if debug_dump else torch.cuda.CUDAGraph()
        if debug_dump:
            graph.enable_debug_mode()  # keeps the captured hipGraph so that debug_dump can print it (diagnosis only)
        self._graph2 = None
        if self.combined_backward:
            with torch.cuda.graph(graph):
                outs = self._combined_pass(self._static_in)
                if self.grad_sync is None:
                    self.optimizer.step()
                else:
                    self.grad_sync.pack("all")
            if self.grad_sync is not None:
                self.grad_sync.attach()
        elif self.grad_sync is None:
            with torch.cuda.graph(graph):
                outs = self._pass_one(self._static_in)
                self._pass_two(outs[0])
                self.optimizer.step()
        else:
            if self.grad_sync.n_early:
                with torch.cuda.graph(graph):
                    outs = self._pass_one(self._static_in)
                    self.grad_sync.pack("early")
                self._graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph2, pool=graph.pool()):  # pass 2 walks the autograd graph of pass 1
                    self._pass_two(outs[0])
                    self.grad_sync.pack("late")
            else:  # a synchronizer that was not told which parameters finish late: one graph, one exchange at the end
                with torch.cuda.graph(graph):
                    outs = self._pass_one(self._static_in)
                    self._pass_two(outs[0])
                    self.grad_sync.pack("all")
            self.grad_sync.attach()  # from now on .grad are the flat views the captured pack() fills on every replay
        self._graph_opt = None
        if self.grad_sync is not None and type(self.optimizer).__module__.endswith("ac_tsr_amd.optim"):
            # [r4] under data parallelism the collectives sit between the captured backward and the optimizer: the optimizer's
            # ONE launch (optim.Adam over the flat views) is captured on its own and replayed behind the exchange instead of
            # being issued eagerly (host time per step: the only eager launch left in the data-parallel step)
            self._graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_opt, pool=graph.pool()):
                self.optimizer.step()
        if debug_dump:
            self._raw_graph = graph.raw_cuda_graph() if hasattr(graph, "raw_cuda_graph") else None  # hipGraph_t (diagnosis)
            graph.debug_dump(debug_dump)  # hipGraphDebugDotPrint: nodes and dependency edges as a .dot file
        self._graph, self._static_out = graph, outs
        return self

    def train_step(self, interaction, check_nan: bool = False):
        """One batch.  Returns (attacked_loss, calibrated_loss) as 0-d tensors (no .item(): the caller decides when
        to synchronise).  In graph mode the returned tensors are static buffers overwritten by the next step."""
        if self._graph is None:
            return self._eager_step(interaction, check_nan)
        if any(interaction[k].shape != buf.shape for k, buf in self._static_in.items() if not k.startswith("_acattn_")):
            # e.g. the shorter last batch of an epoch: the captured graph is shape-specific, run this one eagerly
            if self._fused_inputs:
                self._seed_t += 1  # (the eager step does not advance the replay counter itself in this mode)
            return self._eager_step(interaction, check_nan)
        if not (self._fused_inputs and self._stage_inputs(interaction)):
            for k, buf in self._static_in.items():
                if k.startswith("_acattn_"):
                    continue
                src = interaction[k]
                if src.data_ptr() != buf.data_ptr():
                    buf.copy_(src, non_blocking=True)
            if self._fused_inputs:  # (a batch the input launch cannot take: the pieces one by one)
                self._seed_t += 1
                if "_acattn_last_row" in self._static_in:
                    torch.sub(self._static_in[self.model.ITEM_SEQ_LEN], 1, out=self._static_in["_acattn_last_row"])
        self._graph.replay()
        if self.grad_sync is not None:
            if self._graph2 is not None:
                self.grad_sync.reduce_early(packed=True)  # on the communication stream, under the replay of pass 2
                self._graph2.replay()
            self.grad_sync.all_reduce(packed=True)
            if self._graph_opt is not None:
                self._graph_opt.replay()
            else:
                self.optimizer.step()
        if check_nan:
            for t in self._static_out:
                if t is not None:
                    self._check_nan(t)
        return self._static_out

    def _train_epoch(self, train_data: Iterable, epoch_idx: int = 0, attack: bool = True, calibrate: bool = True):
        assert attack or calibrate
        self.model.train()
        # owned accumulators: in graph mode train_step returns static buffers that the next replay overwrites, so the
        # running sums must never alias them (the reference sums .item() values, trainer.py:664-669; one sync per
        # epoch here instead of two per batch)
        total_att = torch.zeros((), device=self.device)
        total_cal = torch.zeros((), device=self.device)
        saw_att = False
        for interaction in train_data:
            interaction = {k: v.to(self.device) for k, v in interaction.items()} if isinstance(interaction, dict) \
                else interaction.to(self.device)
            att, cal = self.train_step(interaction, check_nan=True)
            if att is not None:
                total_att += att.detach()
                saw_att = True
            total_cal += cal.detach()
        return (total_att.item() if saw_att else 0), total_cal.item()

    @torch.no_grad()
    def evaluate_scores(self, interaction):
        """_full_sort_batch_eval (trainer.py:926-945): full-sort logits with the padding item masked out."""
        self.model.eval()
        _, scores = self.model.full_sort_predict(interaction)
        scores = scores.view(-1, self.model.n_items)
        scores[:, 0] = -float('inf')
        return scores


class ACSASRecTrainer(AttackSASRecTrainer):
    """Alias kept by the reference (trainer.py:1042-1044)."""
