"""GPU suite (-m gpu): the fused projections (acattn_projections_fwd / _bwd, csrc/acattn_proj.hip)

    mq, mk, mv = query(x), key(x), value(x)                               recbole/model/layers.py:687-689
    qa, ka     = attack_query_transform(mq), attack_key_transform(mk)     recbole/model/layers.py:658-659
    gate       = gate(mq)                                                 recbole/model/layers.py:887

against the same six nn.Linear in fp64 on the CPU: outputs, and under the trainer's two backward passes
(recbole/trainer/trainer.py:672-684) every gradient that pass keeps.

Tolerances: outputs 2e-5 absolute (O(1) values, fp32 products); gradients 2e-4 of the tensor's largest magnitude."""
import pytest
import torch
import torch.nn.functional as F

from ac_tsr_amd import linear
from ac_tsr_amd.state import StepState

pytestmark = pytest.mark.gpu
DEV = "cuda"
W = ("wq", "bq", "wk", "bk", "wv", "bv", "waq", "baq", "wak", "bak", "wg", "bg")
OUT = ("mq", "mk", "mv", "qa", "ka", "gate")


def _inputs(rows, H, G, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    t = dict(x=r(rows, H))
    for n in ("q", "k", "v", "aq", "ak"):
        t["w" + n], t["b" + n] = 0.2 * (64 / H) ** 0.5 * r(H, H), 0.1 * r(H)
    if G:
        t["wg"], t["bg"] = 0.2 * (64 / H) ** 0.5 * r(G, H), 0.1 * r(G)
    cot = {k: r(rows, G if k == "gate" else H) for k in OUT if (k != "gate" or G)}
    return t, cot


def _reference(t, G):
    d = {k: v.double().requires_grad_(True) for k, v in t.items()}
    mq, mk, mv = F.linear(d["x"], d["wq"], d["bq"]), F.linear(d["x"], d["wk"], d["bk"]), F.linear(d["x"], d["wv"], d["bv"])
    out = dict(mq=mq, mk=mk, mv=mv, qa=F.linear(mq, d["waq"], d["baq"]), ka=F.linear(mk, d["wak"], d["bak"]))
    if G:
        out["gate"] = F.linear(mq, d["wg"], d["bg"])
    return d, out


# hidden 128 / 256 [round 3]: the streamed-weight kernels (BASELINE configs[3], [4]); 32 resp. 16 rows per wave, so the
# ragged row counts end inside a wave's second row block, and G = 200 / 130 / 37 end inside a gate tile
WIDE = [(512, 50, 128), (37, 50, 128), (16400, 130, 128), (100, 0, 128), (200, 200, 128), (48, 37, 128), (20480 + 17, 100, 128),
        (512, 50, 256), (37, 0, 256), (4117, 200, 256), (48, 37, 256)]


@pytest.mark.parametrize("rows,G,H", [(512, 50, 64), (37, 50, 64), (16384 + 21, 50, 64), (100, 0, 64), (64, 64, 64), (48, 37, 64),
                                      (200, 200, 64), (16400, 130, 64)] + WIDE)
def test_fused_projections_match_fp64_linears(rows, G, H):
    t, cot = _inputs(rows, H, G, seed=rows + G)
    d, ref = _reference(t, G)
    loss = sum((ref[k] * cot[k].double()).sum() for k in cot)
    names = ["x"] + [n for n in W if n in t]
    want = dict(zip(names, torch.autograd.grad(loss, [d[n] for n in names])))
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    args = [dev.get(n) for n in W]
    attack_names = ("waq", "baq", "wak", "bak")

    def run(state, attack_upstream=True):
        outs = linear._FusedProjections.apply(dev["x"], *args, attack_upstream, state)
        got = dict(zip(OUT, outs))
        return got, sum((got[k] * cot[k].to(DEV)).sum() for k in cot)

    st = StepState()
    got, dloss = run(st)
    for k in cot:
        assert (got[k].detach().cpu() - ref[k].detach().float()).abs().max() <= 2e-5, k

    def check(gr, keys):
        for k in keys:
            err = (gr[k].cpu() - want[k].float()).abs().max().item()
            assert err <= 2e-4 * want[k].abs().max().item() + 1e-6, (k, err)

    # no pass restriction (a module used on its own): every gradient
    check(dict(zip(names, torch.autograd.grad(dloss, [dev[n] for n in names]))), names)
    # pass 1: attack transforms frozen;  pass 2: only the attack transforms (+ the input when something upstream holds one)
    others = [n for n in names if n not in attack_names]
    got, dloss = run(st)
    with st.calibrated_pass():
        check(dict(zip(others, torch.autograd.grad(dloss, [dev[n] for n in others], retain_graph=True))), others)
    keep = ["x"] + list(attack_names)
    with st.attack_pass():
        check(dict(zip(keep, torch.autograd.grad(dloss, [dev[n] for n in keep]))), keep)
    # first layer in pass 2: no input gradient is owed, the node returns only the attack transforms' gradients
    got, dloss = run(st, attack_upstream=False)
    with st.attack_pass():
        gr = torch.autograd.grad(dloss, [dev[n] for n in attack_names])
    check(dict(zip(attack_names, gr)), attack_names)


@pytest.mark.parametrize("H", [64, 128, 256])
def test_fused_projections_equal_library_node(H):
    rows, G = 25600, 50
    t, cot = _inputs(rows, H, G, seed=3)
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    names = ["x"] + list(W)
    res = []
    for node in (linear._FusedProjections, linear._Projections):
        outs = node.apply(dev["x"], *(dev[n] for n in W), True, StepState())
        loss = sum((o * cot[k].to(DEV)).sum() for k, o in zip(OUT, outs))
        res.append((outs, torch.autograd.grad(loss, [dev[n] for n in names])))
    for a, b in zip(res[0][0], res[1][0]):
        assert (a - b).abs().max() <= 2e-5
    for n, a, b in zip(names, res[0][1], res[1][1]):
        assert (a - b).abs().max() <= 2e-4 * b.abs().max() + 1e-6, n


@pytest.mark.parametrize("rows,H", [(100, 64), (25600, 64), (100, 128), (25600, 128), (1000, 256)])
def test_residual_gradient_enters_the_backward_launch(rows, H):
    """The node's seventh output is x for the layer's residual connections: a cotangent on it is the start value of dx
    in the backward launch (acattn_proj_bwd_io.dx_init), i.e. dx = dx(projections) + d_res, and alone it passes through."""
    G = 50
    t, cot = _inputs(rows, H, G, seed=5)
    dev = {k: v.to(DEV).requires_grad_(True) for k, v in t.items()}
    res_cot = torch.randn(rows, H, generator=torch.Generator().manual_seed(9)).to(DEV)
    outs = linear._FusedProjections.apply(dev["x"], *(dev[n] for n in W), True, StepState())
    assert len(outs) == 8 and outs[6].data_ptr() == dev["x"].data_ptr()
    loss = sum((o * cot[k].to(DEV)).sum() for k, o in zip(OUT, outs))
    base, = torch.autograd.grad(loss, [dev["x"]], retain_graph=True)
    both, = torch.autograd.grad(loss + (outs[6] * res_cot).sum(), [dev["x"]], retain_graph=True)
    assert (both - (base + res_cot)).abs().max() <= 1e-5 * (base.abs().max() + 1)
    alone, = torch.autograd.grad((outs[6] * res_cot).sum(), [dev["x"]])
    assert torch.equal(alone, res_cot)
