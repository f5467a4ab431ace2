import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ac_tsr_amd as A
from ac_tsr_amd import _lib
from oracle import ac_tsr_ref as O
from tests.test_hip_onehop import _problem, _oracle_mask, _affine_planes
DEV = "cuda"
B, L, H, nh = 512, 50, 64, 2
for causal in (True, False):
  for p_drop in (0.5, 0.0):
    for left_pad in (True, False):
        t, kv, lens, g = _problem(B, L, H, nh, seed=404, causal=causal, left_pad=left_pad)
        dev = {k: v.to(DEV) for k, v in t.items()}
        cfg = A.AttentionConfig(n_heads=nh, adversarial=False)
        mask = A.StructuredMask(kv.to(DEV), causal=causal)
        kw = {k: dev[k] for k in ("w_order", "b_order", "w_dist", "b_dist", "scalar")}
        seed = 31337
        lib = _lib.load()
        ocfg = O.EncoderCfg(n_layers=1, n_heads=nh, hidden_size=H, inner_size=4 * H, combine_option="gate", seq_length=L, attn_dropout_prob=p_drop)
        keep_after = None
        if p_drop > 0:
            keep_after = A.materialize_randomness(B, nh, L, seed, p_drop, DEV).keep_after.cpu().float()
        with torch.no_grad():
            zeros = torch.zeros(B, nh, L, L)
            after = O.core_from_projected(t["q"], t["k"], t["v"], t["q"], t["k"], t["gl"], _oracle_mask(kv, causal), t["w_order"],
                                          t["b_order"], t["w_dist"], t["b_dist"], t["scalar"], ocfg, zeros, keep_after=keep_after, materialize=False)["after"]
            v = O._heads(t["v"], nh).permute(0, 2, 1, 3)
            expect = O.context_only(after, v)
        for which in (_lib.FWD_STREAM, 3):
            lib.acattn_select_forward_kernel(which)
            for pre in (True, False):
                extra = dict(affine=_affine_planes(t, nh).to(DEV)) if pre else {}
                _, ctx, M, _ = A.calibrated_attention(dev["q"], dev["k"], dev["v"], None, None, None, mask, cfg, p_drop=p_drop, seed=seed, **extra, **kw)
                d = (ctx.cpu() - expect).abs()
                bad = (d > 1e-4).nonzero()
                print(f"causal={causal} p_drop={p_drop} left_pad={left_pad} kernel={which} pre={pre}: max {d.max().item():.3e}  n_bad {len(bad)}", end="")
                if len(bad):
                    bs = sorted(set(bad[:, 0].tolist()))
                    print("  seqs", bs[:8], "lens", [int(lens[b]) for b in bs[:8]], "rows", sorted(set(bad[:, 1].tolist()))[:12], "heads", sorted(set((bad[:, 2] // 32).tolist())), end="")
                    b0, i0, c0 = bad[0].tolist()
                    print(f"  first ({b0},{i0},{c0}) got {ctx[b0,i0,c0].item():.6f} want {expect[b0,i0,c0].item():.6f}", end="")
                print()
        lib.acattn_select_forward_kernel(_lib.FWD_AUTO)
