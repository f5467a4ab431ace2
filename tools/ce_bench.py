import sys, time, torch
sys.path.insert(0, ".")
from ac_tsr_amd import ce
dev = "cuda"
B, N, H = 512, 100000, 64
g = torch.Generator().manual_seed(0)
out = (0.5 * torch.randn(B, H, generator=g)).to(dev).requires_grad_(True)
tab = (0.5 * torch.randn(N, H, generator=g)).to(dev).requires_grad_(True)
tgt = torch.randint(0, N, (B,), generator=g).to(dev)
def run(table_grad=True):
    loss = ce.full_sort_cross_entropy(out, tab, tgt)
    torch.autograd.grad(loss, [out, tab] if table_grad else [out])
for tg in (True, False):
    for _ in range(3): run(tg)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run(tg)
    torch.cuda.synchronize(); print("fused CE fwd+bwd table_grad=%s: %.1f us" % (tg, (time.perf_counter() - t) / 20 * 1e6))
def ref():
    l = torch.nn.functional.cross_entropy(out @ tab.t(), tgt); torch.autograd.grad(l, [out, tab])
for _ in range(3): ref()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): ref()
torch.cuda.synchronize(); print("torch matmul + cross_entropy fwd+bwd: %.1f us" % ((time.perf_counter() - t) / 20 * 1e6))
