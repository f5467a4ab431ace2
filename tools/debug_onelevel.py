import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ac_tsr_amd as A
from oracle import ac_tsr_ref as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_hip_onehop import _problem, _oracle_mask
DEV = "cuda"
B, L, H, nh, causal, rich = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 200, 64, 2, False, "fixed"
t, kv, lens, g = _problem(B, L, H, nh, seed=L + nh, causal=causal)
print("lens", lens)
seed, p_drop = 99, 0.5
rnd = A.materialize_randomness(B, nh, L, seed, p_drop, DEV)
names = ["q", "k", "v", "qa", "ka", "gl", "w_order", "b_order", "w_dist", "b_dist", "scalar"]
cpu = {k: t[k].clone().requires_grad_(True) for k in names}
ocfg = O.EncoderCfg(n_layers=1, n_heads=nh, hidden_size=H, inner_size=4 * H, combine_option="gate", seq_length=L,
                    attn_dropout_prob=p_drop, two_level=False, rich_calibrated_combine=rich)
ref = O.core_from_projected(cpu["q"], cpu["k"], cpu["v"], cpu["qa"], cpu["ka"], cpu["gl"], _oracle_mask(kv, causal),
                            cpu["w_order"], cpu["b_order"], cpu["w_dist"], cpu["b_dist"], cpu["scalar"], ocfg,
                            rnd.noise.cpu(), keep_after=rnd.keep_after.cpu().float(), keep_mask=rnd.keep_mask.cpu().float(),
                            keep_before=rnd.keep_before.cpu().float())
cot = {k: torch.randn(ref[k].shape, generator=g) for k in ("ctx_attacked", "ctx_calibrated", "M")}
want = dict(zip(names, torch.autograd.grad(sum((ref[k] * cot[k]).sum() for k in cot), [cpu[k] for k in names])))
dev = {k: t[k].to(DEV).requires_grad_(True) for k in names}
cfg = A.AttentionConfig(n_heads=nh, combine_option="gate", two_level=False, rich_calibrated_combine=rich)
mask = A.StructuredMask(kv.to(DEV), causal=causal)
ctx_a, ctx_c, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], dev["qa"], dev["ka"], dev["gl"], mask, cfg,
                                            p_drop=p_drop, seed=seed, **{k: dev[k] for k in names[6:]})
loss = sum((o * cot[k].to(DEV)).sum() for o, k in ((ctx_a, "ctx_attacked"), (ctx_c, "ctx_calibrated"), (M, "M")))
got = dict(zip(names, torch.autograd.grad(loss, [dev[k] for k in names])))
for k in names:
    e = (got[k].cpu() - want[k]).abs()
    print(k, "err", e.max().item(), "scale", want[k].abs().max().item())
e = (got["gl"].cpu() - want["gl"]).abs()
flat = e.flatten().topk(20)
for v, i in zip(flat.values, flat.indices):
    b, r = divmod(i.item(), L * L)
    i_, j_ = divmod(r, L)
    print(f"b {b} i {i_} j {j_} err {v.item():.3e} got {got['gl'][b, i_, j_].item():.4e} want {want['gl'][b, i_, j_].item():.4e}")
print("rows with err > 1e-5:", sorted(set((e > 1e-5).nonzero()[:, 1].tolist()))[:50])
print("cols with err > 1e-5:", sorted(set((e > 1e-5).nonzero()[:, 2].tolist()))[:50])
