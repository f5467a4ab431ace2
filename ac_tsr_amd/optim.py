"""torch.optim.Adam with its update done by ONE launch for all parameters (include/acattn.h: acattn_adam_step).

The reference builds `optim.Adam(params, lr, weight_decay)` (recbole/trainer/trainer.py:590-615); the optimizer step is
part of the training step bench.py measures.  `Adam` below IS torch.optim.Adam -- same constructor, same `state` layout
(`step`, `exp_avg`, `exp_avg_sq` per parameter, `capturable`: the step counters live on the device), same state_dict --
with `step()` routed to the library when the situation is the plain one (fp32 parameters and dense gradients on one HIP
device, no amsgrad / maximize / differentiable, float learning rate).  Anything else, and the very first step (which
creates the state), goes through torch's own implementation.  The learning rate travels to the launch BY VALUE: a
captured graph replays the rate of the capture (trainer.enable_graph refuses a model whose optimizer has an LR scheduler
attached; use a tensor `lr`, which takes torch's path, to change it under replay).  The arithmetic reproduces ATen's fused kernel operation by
operation (csrc/acattn_adam.hip; tests/test_hip_adam.py compares the two).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .ops import _stream


class Adam(torch.optim.Adam):
    def _plain(self, group) -> bool:
        return not (group.get("amsgrad") or group.get("maximize") or group.get("differentiable")) and group.get("capturable") \
            and isinstance(group["lr"], float) and group.get("decoupled_weight_decay", False) is False

    @torch.no_grad()
    def step(self, closure=None):
        todo = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            ok = self._plain(group) and all(
                p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == p.device
                and len(self.state[p]) != 0 and torch.is_tensor(self.state[p]["step"]) and self.state[p]["step"].is_cuda
                and self.state[p]["step"].dtype == torch.float32 for p in ps)
            if not ok or not ps:
                # torch's own implementation, UNDECORATED: both this step() and torch.optim.Adam.step are wrapped by
                # Optimizer.profile_hook_step, and going through the wrapper again would run the step pre / post hooks
                # (and an LR scheduler's step counting) twice
                inner = getattr(torch.optim.Adam.step, "__wrapped__", None)
                return inner(self, closure) if inner is not None else super().step(closure)
            todo.append((group, ps))
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group, ps in todo:
            done = self.__dict__.get("_acattn_done")
            if done is None or done.device != ps[0].device:
                done = self.__dict__["_acattn_done"] = torch.zeros(1, dtype=torch.int32, device=ps[0].device)
            beta1, beta2 = group["betas"]
            for i in range(0, len(ps), _lib.ADAM_MAX_TENSORS):
                chunk = ps[i:i + _lib.ADAM_MAX_TENSORS]
                g = _lib.AdamGroup()
                g.n_tensors = len(chunk)
                for k, p in enumerate(chunk):
                    st = self.state[p]
                    g.param[k], g.grad[k] = p.data_ptr(), p.grad.data_ptr()
                    g.exp_avg[k], g.exp_avg_sq[k], g.step[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()
                    g.numel[k] = p.numel()
                _lib.check(lib.acattn_adam_step(C.byref(g), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                                float(group["weight_decay"]), done.data_ptr(), _stream()), "adam_step")
        return loss
