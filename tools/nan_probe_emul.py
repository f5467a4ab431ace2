"""Graph-mode seeding emulated eagerly (frozen host seeds + the device step counter), with finiteness checks after every
walk: finds the step, the walk and the parameters where a non-finite gradient first appears, then re-runs that step's
forward with hooks on every module output.   python tools/nan_probe_emul.py --config cfg4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
import bench
import ac_tsr_amd as A

a = bench.parse()
device = torch.device("cuda:0")
torch.manual_seed(42)
model = getattr(A, a.model)(A.DictConfig(bench.model_config(a)), A.ItemCount(a.items)).to(device)
trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model)
model.train()
gen = torch.Generator().manual_seed(1000)
pool = [bench.synthetic_batch(a.batch, a.seq_len, a.items, gen, device) for _ in range(8)]
st = trainer.state
seed_t = torch.zeros(1, dtype=torch.int64, device=device)
trainer._seed_t = seed_t
st.seed_tensor = seed_t
draws = {"i": 0}
base = int(os.environ.get("PROBE_SEED", "12345"))
st_cls = type(st)
def frozen_draw(self):
    draws["i"] += 1
    return (base * 7919 + draws["i"] * 104729) & 0x7FFFFFFFFFFFFFFF
st_cls.draw_seed = frozen_draw

def finite(ts):
    return [n for n, t in ts if t is not None and not torch.isfinite(t).all()]

for i in range(int(os.environ.get("PROBE_STEPS", "300"))):
    draws["i"] = 0
    seed_t += 1
    batch = pool[i % 8]
    trainer.optimizer.zero_grad(set_to_none=True)
    att, cal = model.calculate_loss(batch)
    with st.calibrated_pass():
        cal.backward(retain_graph=True, inputs=trainer._others)
    bad1 = finite([(n, p.grad) for n, p in model.named_parameters()])
    with st.attack_pass():
        att.backward(inputs=trainer._attack)
    bad2 = finite([(n, p.grad) for n, p in model.named_parameters()])
    if i % 50 == 0 or bad1 or bad2:
        print(i, float(att.detach()), float(cal.detach()), "pass1:", bad1[:8], "pass2:", [n for n in bad2 if n not in bad1][:8], flush=True)
    if bad1 or bad2:
        # re-run this step's forward with the same draws and look at every module's output
        draws["i"] = 0
        bad_out = []
        def hook(name):
            def f(mod, inp, out):
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for k, o in enumerate(outs):
                    if torch.is_tensor(o) and o.is_floating_point() and not torch.isfinite(o).all():
                        bad_out.append((name, k, tuple(o.shape), int((~torch.isfinite(o)).sum())))
            return f
        hs = [m.register_forward_hook(hook(n)) for n, m in model.named_modules()]
        with torch.no_grad():
            model.calculate_loss(batch)
        for h in hs:
            h.remove()
        print("forward outputs with non-finite values:", bad_out[:12])
        sys.exit(3)
    trainer.optimizer.step()
print("clean")
