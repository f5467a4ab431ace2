"""GPU suite (-m gpu): the one-launch Adam update (acattn_adam_step, csrc/acattn_adam.hip, ac_tsr_amd/optim.py) against
torch.optim.Adam(fused=True, capturable=True), the implementation the trainer used before: same parameters, same
gradients, several steps.  The arithmetic is reproduced operation by operation; what is allowed to differ is the
contraction of a multiply-add in double (<= 1 ulp of the float result)."""
import pytest
import torch

from ac_tsr_amd.optim import Adam

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(100001, 64), (64, 64), (64,), (256, 64), (50, 64), (50,), (1,), (3, 7), (4099,), (16384, 5)]


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_one_launch_adam_equals_torch_fused_adam(weight_decay):
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(*s, generator=g) * 0.1 for s in SHAPES]
    grads = [[torch.randn(*s, generator=g) * (0.01 if i % 2 else 1.0) for s in SHAPES] for i in range(6)]

    def run(cls, **kw):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        opt = cls(ps, lr=1e-3, weight_decay=weight_decay, capturable=True, **kw)
        for step_grads in grads:
            for p, gr in zip(ps, step_grads):
                p.grad = gr.to(DEV).clone()
            opt.step()
        return ps, opt

    ref, ropt = run(torch.optim.Adam, fused=True)
    got, gopt = run(Adam, fused=True)
    for s, a, b in zip(SHAPES, got, ref):
        assert torch.isfinite(a).all()
        diff = (a - b).abs().max().item()
        assert diff <= 2e-7 * b.abs().max().item() + 1e-9, (s, diff)
        for key in ("exp_avg", "exp_avg_sq", "step"):
            x, y = gopt.state[a][key], ropt.state[b][key]
            assert (x - y).abs().max().item() <= 2e-7 * y.abs().max().item() + 1e-12, (s, key)
    assert float(gopt.state[got[0]]["step"]) == len(grads)


def test_state_dict_round_trip_and_fallbacks():
    p = torch.nn.Parameter(torch.randn(300, 64, device=DEV))
    q = torch.nn.Parameter(torch.randn(64, device=DEV))
    opt = Adam([p, q], lr=1e-3, capturable=True, fused=True)
    for _ in range(3):
        p.grad, q.grad = torch.randn_like(p), None  # a parameter without a gradient is skipped, like torch does
        opt.step()
    assert float(opt.state[p]["step"]) == 3 and len(opt.state[q]) == 0
    sd = opt.state_dict()
    opt2 = Adam([p, q], lr=1e-3, capturable=True, fused=True)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.state[p]["exp_avg"], opt.state[p]["exp_avg"])
    # amsgrad is not covered by the launch: torch's implementation runs
    r = torch.nn.Parameter(torch.randn(10, device=DEV))
    ams = Adam([r], lr=1e-3, amsgrad=True, capturable=True)
    r.grad = torch.randn_like(r)
    ams.step()
    ams.step()
    assert "max_exp_avg_sq" in ams.state[r]
