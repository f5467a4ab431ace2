import sys, torch
sys.path.insert(0, ".")
import ac_tsr_amd as A
DEV = "cuda"
B, L, H, nh = 96, 50, 64, 2
g = torch.Generator().manual_seed(7)
q, k, v, qa, ka = (torch.randn(B, L, H, generator=g).to(DEV) for _ in range(5))
gl = torch.randn(B, L, L, generator=g).to(DEV)
lens = torch.randint(1, L + 1, (B,), generator=g)
kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8)
kv[0] = 1 - kv[0]
kv = kv.to(DEV)
w = lambda *s: (0.3 * torch.randn(*s, generator=g)).to(DEV)
dh = H // nh
kw = dict(w_order=w(1, 2 * dh), b_order=w(1), w_dist=w(1, 2 * dh), b_dist=w(1), scalar=w(1))
cfg = A.AttentionConfig(n_heads=nh, combine_option="gate")
for causal in (True, False):
    mask = A.StructuredMask(kv, causal=causal)
    seed = 991
    fast = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=0.0, seed=seed, **kw)
    rnd = A.materialize_randomness(B, nh, L, seed, 0.0, DEV)
    ref = A.calibrated_attention(q, k, v, qa, ka, gl, mask, cfg, p_drop=0.0, rnd=A.ExplicitRandomness(noise=rnd.noise), **kw)
    for n, a, b in zip(("ctx_a", "ctx_c", "M"), fast[:3], ref[:3]):
        d = (a - b).abs()
        idx = torch.nonzero(d == d.max())[0].tolist()
        print(causal, n, "max diff", d.max().item(), "at", idx, a[tuple(idx)].item(), b[tuple(idx)].item(), "len", lens[idx[0]].item())
