"""bench.py's graph-mode training loop with the losses printed every step:  python tools/nan_probe_graph.py --config cfg4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ac_tsr_amd as A

a = bench.parse()
device = torch.device("cuda:0")
torch.manual_seed(42)
model = getattr(A, a.model)(A.DictConfig(bench.model_config(a)), A.ItemCount(a.items)).to(device)
if a.model == "AcBERT4Rec":
    model.cloze_on_device = True  # as bench.py does: the cloze batch built with tensor ops, so that the step captures
trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model, combined_backward=a.combined_backward)
model.train()
gen = torch.Generator().manual_seed(1000)
pool = [bench.synthetic_batch(a.batch, a.seq_len, a.items, gen, device) for _ in range(8)]
if os.environ.get("PROBE_GRAPH", "1") == "1":
    trainer.enable_graph(pool[0])
n_steps = int(os.environ.get("PROBE_STEPS", "24"))
for i in range(n_steps):
    att, cal = trainer.train_step(pool[i % 8])
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    fa, fc = float(att.detach()), float(cal.detach())
    if i % 50 == 0 or bad or fa != fa or fc != fc:
        print(i, fa, fc, bad[:6], flush=True)
    if bad or fa != fa or fc != fc:
        sys.exit(3)
print("clean", n_steps)
