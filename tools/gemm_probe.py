import torch, time
dev = "cuda"
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
shapes = [(25600, 64, 64), (25600, 64, 256), (25600, 256, 64), (25600, 64, 50), (25600, 64, 192)]
for lib in ("cublaslt", "cublas", "ck"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        g = torch.randn(M, N, device=dev)
        try:
            t1 = bench(lambda: torch.nn.functional.linear(x, w, b))
            t2 = bench(lambda: g.t() @ x)      # weight grad
            t3 = bench(lambda: g @ w)          # input grad
            print(f"{lib:9s} M={M} K={K} N={N}: fwd {t1:7.1f} us  dW {t2:7.1f} us  dX {t3:7.1f} us")
        except Exception as e:
            print(lib, (M, K, N), "failed", str(e)[:80])
