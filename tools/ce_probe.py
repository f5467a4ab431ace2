import torch, time
dev = "cuda"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
B, N, H = 512, 100000, 64
g = torch.randn(B, N, device=dev); E = torch.randn(N, H, device=dev); h = torch.randn(B, H, device=dev)
print("g @ E            %.1f us" % bench(lambda: g @ E))
for s in (10, 25, 50, 100, 125):
    print("split-K bmm s=%d  %.1f us" % (s, bench(lambda: torch.bmm(g.view(B, s, N // s).transpose(0, 1), E.view(s, N // s, H)).sum(0))))
print("g.t() @ h        %.1f us" % bench(lambda: g.t() @ h))
print("h @ E.t()        %.1f us" % bench(lambda: h @ E.t()))
idx = torch.randint(1, N, (512, 50), device=dev); gg = torch.randn(512, 50, H, device=dev); W = torch.randn(N, H, device=dev, requires_grad=True)
def dense():
    out = torch.nn.functional.embedding(idx, W); out.backward(gg); W.grad = None
def atomic():
    gw = gg.new_zeros(N, H); gw.index_add_(0, idx.reshape(-1), gg.reshape(-1, H)); return gw
print("embedding fwd+dense bwd %.1f us" % bench(dense))
print("zeros + index_add_      %.1f us" % bench(atomic))
lg = torch.randn(B, N, device=dev, requires_grad=True); pos = torch.randint(0, N, (B,), device=dev)
def ce():
    l = torch.nn.functional.cross_entropy(lg, pos); l.backward(); lg.grad = None
print("cross_entropy fwd+bwd on [512,100k]  %.1f us" % bench(ce))
