"""Diagnostic: per-wave phase timeline of the fast forward kernel (stamp build of the library)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, ".")
from ac_tsr_amd import _lib
_lib.LIB_PATH = os.path.join(_lib.CSRC, "libacattn_stamps.so")
lib = _lib.load()
dev = "cuda"
B, L, H, nh = 512, 50, 64, 2
g = torch.Generator().manual_seed(0)
def mk():
    q, k, v, qa, ka = (torch.randn(B, L, H, generator=g).to(dev) for _ in range(5))
    gl = torch.randn(B, L, L, generator=g).to(dev)
    return q, k, v, qa, ka, gl
sets = [mk() for _ in range(6)]
lens = torch.randint(1, L + 1, (B,), generator=g)
kv = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).to(dev)
w = lambda *s: (0.02 * torch.randn(*s, generator=g)).to(dev)
wo, bo, wd, bd, sc = w(64), w(1), w(64), w(1), w(1)
ctx_a, ctx_c = torch.empty(B, L, H, device=dev), torch.empty(B, L, H, device=dev)
M = torch.empty(B, nh, L, L, device=dev); st = torch.empty(B, nh, L, 8, device=dev)
stamps = torch.zeros(B * nh * 4 * 8, dtype=torch.int64, device=dev)
def run(i):
    q, k, v, qa, ka, gl = sets[i % 6]
    p = _lib.Problem(); p.B, p.L, p.H, p.n_heads = B, L, H, nh
    p.q, p.k, p.v, p.qa, p.ka, p.gate_logits = (t.data_ptr() for t in (q, k, v, qa, ka, gl))
    p.mask_mode, p.causal, p.key_valid = 0, 1, kv.data_ptr()
    p.w_order, p.b_order, p.w_dist, p.b_dist, p.scalar = (t.data_ptr() for t in (wo, bo, wd, bd, sc))
    p.adversarial, p.combine_option, p.two_level, p.rng_mode, p.p_drop, p.seed = 1, 1, 1, 1, 0.5, 5
    o = _lib.FwdOut(); o.ctx_attacked, o.ctx_calibrated, o.attack_mask, o.row_stats = ctx_a.data_ptr(), ctx_c.data_ptr(), M.data_ptr(), st.data_ptr()
    o.after_spatial = stamps.data_ptr()
    rc = lib.acattn_calibrated_attention_fwd(C.byref(p), C.byref(o), None); assert rc == 0, rc
for i in range(8): run(i)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(B * nh, 4, 8).astype(np.int64)
t0 = s[..., 0].min()
rel = (s[..., :7] - t0).astype(np.float64)
print("kernel span (cycles of s_memtime):", rel[..., 6].max())
names = ["start", "kv_staged", "g_staged", "km_lt", "barrier", "pass1", "end"]
for k, n in enumerate(names):
    print(f"{n:14s} mean {rel[..., k].mean():9.0f}  p10 {np.percentile(rel[..., k], 10):9.0f}  p90 {np.percentile(rel[..., k], 90):9.0f} max {rel[..., k].max():9.0f}")
d = np.diff(rel, axis=-1)
for qb in range(4):
    print("qb", qb, "mean nt", s[:, qb, 7].mean(), "phase durations:", {names[k + 1]: int(d[:, qb, k].mean()) for k in range(6)})
print("---- per-XCD (blockIdx % 8) timing, cycles ----")
blk = np.arange(B * nh)
for x in range(8):
    sel = s[blk % 8 == x]
    st_, en_ = sel[..., 0], sel[..., 6]
    base = st_.min()
    print(f"xcd-group {x}: start p0 {0} p50 {np.percentile(st_ - base, 50):.0f} p100 {(st_ - base).max():.0f} | end p50 {np.percentile(en_ - base, 50):.0f} max {(en_ - base).max():.0f} | n={sel.shape[0]}")
sel = s[blk % 8 == 0]
order = np.argsort(sel[:, 0, 0])
base = sel[..., 0].min()
print("xcd-group 0 blocks sorted by start: (block-in-group, start, end) every 8th")
for k in order[::8]:
    print(int(k), int(sel[k, 0, 0] - base), int(sel[k, :, 6].max() - base))
print("---- clusters by absolute counter (waves whose start is within 200k cycles of the median of their cluster) ----")
allst = s[:, 0, 0]
rem = np.ones(len(allst), bool)
for it in range(10):
    if not rem.any(): break
    med = np.median(allst[rem])
    cl = np.abs(allst - med) < 200000
    cl &= rem
    if cl.sum() < 8:
        rem &= ~cl
        if cl.sum() == 0:
            rem[np.argmax(rem)] = False
        continue
    st_ = s[cl][..., 0]; en_ = s[cl][..., 6]
    base = st_.min()
    q = np.percentile(st_[:, 0] - base, [0, 10, 25, 50, 75, 90, 100])
    print(f"cluster n={cl.sum()} start quantiles {q.astype(int).tolist()} end max {int(en_.max() - base)} end p50 {int(np.percentile(en_ - base, 50))}")
    rem &= ~cl
