#!/usr/bin/env python3
"""bench.py -- AC-SASRec training throughput on MI355X + roofline of the fused calibrated-attention kernel.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL) -- or without a launcher, in which
case this script starts the N ranks itself as child processes; fewer than N visible devices is an error, never a
1-GPU number.  Rank 0 prints ONE JSON line (`n_gpus`, `rccl_ranks` = size of the process group the steps ran in).

  step      = one pass of the hot path over one synthetic batch: ACSASRec.calculate_loss (embedding -> LN ->
              2 calibrated encoder layers, each ONE fused HIP attention launch -> two CE losses + mask penalty),
              the trainer's two backward passes, gradient all-reduce (N > 1) and one Adam step.
  value     = user-sequences/sec over all ranks (weak scaling: B = 512 sequences per GPU), inputs resident in HBM.
  roofline  = fused calibrated-attention FORWARD kernel (contract A): algorithmic bytes per launch /
              average launch duration, measured here with HIP events over back-to-back launches on rotating
              buffer sets (> 256 MiB, so the Infinity Cache cannot serve them), against the 8 TB/s HBM3E peak.
  cpu_baseline = the oracle (CPU restatement of the reference algorithm, incl. its [B,h,L,L,2dh] concatenation)
              running the same full step on the host cores; rank 0, N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def _ce_products_mode(a):
    """How the full-catalogue cross-entropy evaluates its products in this run (include/acattn.h ABI 28)."""
    from ac_tsr_amd._lib import load
    mode = load().acattn_full_sort_ce_products(-1)
    if mode == 0:
        return "fp32 MFMA (v_mfma_f32_16x16x4_f32)"
    split = "fp32 operands split exactly into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 accumulation (fp32 accuracy: tests/test_hip_ce.py)"
    applies = mode == 2 or (getattr(a, "hidden", 0) == 64 and getattr(a, "items", 0) > 65536)
    return split if applies else "fp32 MFMA (the split sweeps cover hidden 64 with more than 65,536 items)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default=None,
                    help="BASELINE.json configs[1..4] presets: cfg2/cfg3 = the default L=50 d=64 workload (the metric's "
                         "configuration; cfg2 is its spatial-only kernel line), cfg4 = L=200 d=128 4 heads AC-SASRec, "
                         "cfg5 = L=200 d=256 4 heads AcBERT4Rec (bidirectional mask)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=50)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--inner", type=int, default=256)
    ap.add_argument("--items", type=int, default=None, help="catalogue size (default 100000; 20000 for --config cfg5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short measurements of the other named shapes (L=200 d=64, configs[3], configs[4])")
    ap.add_argument("--kernel-only", action="store_true", help="only the kernel roofline measurement")
    ap.add_argument("--kernel-iters", type=int, default=300)
    ap.add_argument("--kernel-kinds", default="ragged,full,spatial",
                    help="--kernel-only: which forward-kernel measurements to run (counter passes want one at a time)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--no-full-schedule", action="store_true",
                    help="skip the extra timing of the step with the reference's full (unpruned) schedule")
    ap.add_argument("--fwd-kernel", choices=["auto", "stream", "staged", "general"], default="auto",
                    help="pin the attention forward kernel (acattn_select_forward_kernel); measurements only")
    ap.add_argument("--bwd-kernel", choices=["auto", "stream", "row"], default="auto",
                    help="pin the attention backward kernel (acattn_select_backward_kernel); measurements only")
    ap.add_argument("--tail", choices=["fused", "unfused"], default="fused",
                    help="layer tail: the fused launch (acattn_layer_tail_*) or the unfused node; measurements only")
    ap.add_argument("--projections", choices=["fused", "library"], default="fused",
                    help="the six projections of a layer: one launch (acattn_projections_*) or hipBLASLt GEMMs; measurements only")
    ap.add_argument("--dp-collective", choices=["all_reduce", "reduce_scatter"], default="all_reduce",
                    help="gradient exchange under data parallelism: one all-reduce per bucket, or reduce-scatter + "
                         "all-gather (parallel.GradSynchronizer)")
    ap.add_argument("--combined-backward", action="store_true",
                    help="opt-in trainer mode: ONE backward walk carries the cotangents of both losses (ac_tsr_amd/combined.py) "
                         "instead of the reference protocol's two walks; same gradients")
    ap.add_argument("--force-grad-sync", action="store_true",
                    help="use the data-parallel gradient path (flat buffer, graph without optimizer) even on one GPU")
    a = ap.parse_args()
    a.model = "ACSASRec"
    if a.config == "cfg4":  # "Yelp-scale synthetic, L=200 d=128 4 heads"; the catalogue size is ours (stated in the line)
        a.seq_len, a.hidden, a.heads, a.inner = 200, 128, 4, 512
    elif a.config == "cfg5":  # "AC-BERT4Rec variant (bidirectional mask) L=200 d=256"
        # heads: BASELINE leaves them open; 4 -> head size 64 (the reference default of 2 is head size 128: since round 3
        # the streaming kernels cover it, one wave per SIMD).  20,000 items by default: the masked-slot CE at d=256 goes through
        # materialised [rows, N] logits (measured faster than the fused kernels at this width, model.py) and ~20 k rows x
        # 100 k items are 8 GB per logits tensor; `--items 100000` runs that too.
        heads = a.heads if "--heads" in sys.argv else 4  # `--config cfg5 --heads 2`: the reference default (head size 128)
        a.seq_len, a.hidden, a.heads, a.inner, a.model = 200, 256, heads, 1024, "AcBERT4Rec"
        a.items = a.items or 20000
    a.items = a.items or 100000
    return a


def synthetic_batch(B, L, n_items, gen, device):
    """item_length ~ U{1..L}, right-padded ids ~ U{1..N-1}, target ~ U{1..N-1}."""
    lens = torch.randint(1, L + 1, (B,), generator=gen)
    ids = torch.randint(1, n_items, (B, L), generator=gen)
    ids = ids * (torch.arange(L)[None, :] < lens[:, None])
    return {"item_id_list": ids.to(device), "item_length": lens.to(device),
            "item_id": torch.randint(1, n_items, (B,), generator=gen).to(device)}


def model_config(a):
    extra = dict(mask_ratio=0.2) if a.model == "AcBERT4Rec" else {}
    return dict(**extra, n_layers=a.layers, n_heads=a.heads, hidden_size=a.hidden, inner_size=a.inner, hidden_dropout_prob=0.5,
                attn_dropout_prob=0.5, hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE',
                combine_option='gate', two_level=True, use_order=True, use_distance=True, rich_calibrated_combine='none',
                use_position_embedding=False, trainable_mask_loss_weight=False, mask_loss_weight=0.03,
                MAX_ITEM_LIST_LENGTH=a.seq_len, gate_seq_length=a.seq_len)


# --------------------------------------------------------------------------------------------------
# kernel roofline: direct C-ABI launches on torch's current stream, HIP events around the timed region
# --------------------------------------------------------------------------------------------------
def kernel_roofline(a, device, adversarial=True, iters=300, nsets=None, full_length=False):
    """The forward kernel as the training step launches it: q, k, v, qa, ka, the gate as the projections launch hands
    it over (probabilities, acattn_problem.gate_is_prob) and the affine planes it writes (acattn_problem.affine; here
    produced by acattn_spatial_affines).  `full_length`: every sequence has all L items (SURVEY 8d's worst case) instead
    of item_length ~ U{1..L}."""
    nsets = nsets or int(os.environ.get("ACTSR_BENCH_NSETS", "6"))  # 6 sets = 372 MB > 256 MiB Infinity Cache
    from ac_tsr_amd import _lib, ops
    lib = _lib.load()
    B, L, H, nh = a.batch, a.seq_len, a.hidden, a.heads
    gen = torch.Generator().manual_seed(42)
    sets = []
    keep = []
    w = lambda *s: (0.02 * torch.randn(*s, generator=gen)).to(device)
    w_order, b_order, w_dist, b_dist = w(2 * H // nh), w(1), w(2 * H // nh), w(1)
    scalar = torch.randn(1, generator=gen).to(device)
    for s in range(nsets):
        # unit-variance rows like LN(embedding) pushed through N(0, 0.02) projections would be tiny; use
        # N(0,1) activations so every exp/log sees non-trivial arguments (zero-ish data flatters kernels)
        q, k, v, qa, ka = (torch.randn(B, L, H, generator=gen).to(device) for _ in range(5))
        gl = torch.randn(B, L, L, generator=gen).to(device)
        lens = torch.randint(1, L + 1, (B,), generator=gen)
        if full_length:
            lens = torch.full((B,), L)
        kv = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(device)
        ctx_a, ctx_c = torch.empty_like(q), torch.empty_like(q)
        M = torch.empty(B, nh, L, L, device=device)
        stats = torch.empty(B, nh, L, _lib.NSTAT, device=device)
        p = _lib.Problem()
        p.B, p.L, p.H, p.n_heads = B, L, H, nh
        p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
        p.mask_mode, p.causal, p.key_valid = _lib.MASK_STRUCTURED, int(a.model != "AcBERT4Rec"), kv.data_ptr()
        p.w_order, p.b_order, p.w_dist, p.b_dist, p.scalar = (t.data_ptr() for t in (w_order, b_order, w_dist, b_dist, scalar))
        p.adversarial = int(adversarial)
        p.two_level = 1
        p.rng_mode, p.p_drop, p.seed = _lib.RNG_COUNTER, 0.5, 1234 + s
        o = _lib.FwdOut()
        o.ctx_calibrated = ctx_c.data_ptr()
        if adversarial:
            p.qa, p.ka, p.gate_logits = qa.data_ptr(), ka.data_ptr(), gl.data_ptr()
            p.combine_option = _lib.COMBINE["gate"]
            o.ctx_attacked, o.attack_mask, o.row_stats = ctx_a.data_ptr(), M.data_ptr(), stats.data_ptr()
            if L > 64:
                # as in the training step: the long form of the kernel also writes the mask penalty's row sums
                # (acattn_fwd_out.penalty_part; at L <= 64 they are a separate launch behind the kernel, not part of it)
                pen = torch.empty(B, nh, (L + 15) // 16, device=device)
                o.penalty_part = pen.data_ptr()
                keep.append(pen)
        if not os.environ.get("ACTSR_BENCH_NO_EXTRAS"):
            affine = torch.empty(B, nh, 4, 16 * ((L + 15) // 16), device=device)
            _lib.check(lib.acattn_spatial_affines(C.byref(p), affine.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "spatial_affines")
            p.affine = affine.data_ptr()
            if adversarial:
                gl = torch.sigmoid(gl)
                p.gate_logits, p.gate_is_prob = gl.data_ptr(), 1
            keep.append(affine)
        sets.append((p, o))
        keep.append((q, k, v, qa, ka, gl, kv, ctx_a, ctx_c, M, stats))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fwd = lib.acattn_calibrated_attention_fwd

    def launch(n):
        for i in range(n):
            p, o = sets[i % nsets]
            rc = fwd(C.byref(p), C.byref(o), stream)
            if rc:
                _lib.check(rc, "fwd")

    launch(min(20, iters))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rounds = []
    for _ in range(3):
        e0.record()
        launch(iters)
        e1.record()
        torch.cuda.synchronize()
        rounds.append(e0.elapsed_time(e1) * 1e3 / iters)
    best = sum(rounds) / len(rounds)  # average launch duration over all timed launches
    dh = H // nh
    flavour = "true" if adversarial else "false"
    # the automatic choice of acattn_calibrated_attention_fwd for this configuration (csrc/acattn_fwd.hip)
    # template arguments: head size, key tiles per row, adversarial calibrator, p_drop == 0.5, producer extras
    kernel_name = "acattn_fwd_stream_kernel<%d,%d,%s,true,%s>" % (
        dh, 4 if L <= 64 else 13, flavour, "false" if os.environ.get("ACTSR_BENCH_NO_EXTRAS") else "true")
    if a.fwd_kernel == "staged" and L <= 64:
        kernel_name = ("acattn_fwd_dma_kernel<%d,%s>" if L > 48 else "acattn_fwd_fast_kernel<%d,%s>") % (dh, flavour)
    elif a.fwd_kernel == "general":
        kernel_name = "acattn_fwd_kernel<%d,%d,...>" % (dh, 4 if L <= 64 else (8 if L <= 128 else 13))
    alg = ops.fwd_algorithmic_bytes(B, L, H, nh, adversarial, "gate")
    achieved = alg / (best * 1e-6) / 1e9
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r04_fwd_pmc_full_length.json" if full_length else "fwd_pmc_latest.json")
    if os.path.exists(pmc) and (B, L, H, nh) == (512, 50, 64, 2):
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same kernel and shape
        # (FETCH_SIZE / WRITE_SIZE collected in separate passes, gfx950 FETCH_SIZE x2 correction applied)
        k = json.load(open(pmc))["kernels"].get(kernel_name)
        if k and "hbm_bytes_per_launch_corrected" in k:
            traffic, traffic_src = k["hbm_bytes_per_launch_corrected"], "profiles/" + os.path.basename(pmc)

    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kernel_name,
            "contract": "A" if adversarial else "A'", "algorithmic_bytes_per_launch": alg,
            "item_length": "all L" if full_length else "U{1..L}",
            "avg_launch_us": round(best, 2), "launches_timed": 3 * iters, "buffer_sets": nsets}


def ce_roofline(a, device, iters=40, nsets=3):
    """The three sweeps of the full-catalogue cross-entropy (the largest block of the step: acsasrec.py:117-120) as the
    training step launches them -- forward, forward with direction, backward with table gradient -- each entry point timed
    with events over `iters` calls on `nsets` rotating (out, table) sets.  Matrix bound: `achieved` = the products'
    ALGORITHMIC fp32 FLOPs (2 B N H per product; 1 / 2 / 3 products) over the average call, `peak` = the fp32 matrix
    peak of MI355X_MICROARCH.md (157.3 TFLOP/s: what the exact-fp32 instruction could reach); `bf16_mfma_tflops` = the
    bf16 matrix work actually issued (6 MFMAs per product at hidden 64 in the split-product mode) against 2,500."""
    import ctypes as C
    from ac_tsr_amd import _lib
    lib = _lib.load()
    B, N, H = a.batch, a.items, a.hidden
    gen = torch.Generator().manual_seed(7)
    sets = []
    for s in range(nsets):
        out = torch.randn(B, H, generator=gen).to(device)
        table = (0.05 * torch.randn(N, H, generator=gen)).to(device)
        target = torch.randint(1, N, (B,), generator=gen).to(device)
        p = _lib.CeProblem()
        p.B, p.N, p.H = B, N, H
        p.out, p.table, p.target = out.data_ptr(), table.data_ptr(), target.data_ptr()
        nbytes = lib.acattn_full_sort_ce_workspace_bytes(C.byref(p))
        if nbytes < 0:
            return None
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        lse, row_loss = torch.empty(B, device=device), torch.empty(B, device=device)
        direction, d_out, d_table = torch.empty_like(out), torch.empty_like(out), torch.empty_like(table)
        coef = torch.full((1,), 1.0, device=device)
        pb = _lib.CeProblem()
        pb.B, pb.N, pb.H, pb.out, pb.table, pb.target = B, N, H, p.out, p.table, p.target
        pb.coef_is_scalar, pb.coef_scale = 1, 1.0 / B
        sets.append((p, pb, ws, lse, row_loss, direction, d_out, d_table, coef, out, table, target))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    calls = {
        "forward (acattn_full_sort_ce_fwd)": (1, lambda S: lib.acattn_full_sort_ce_fwd(C.byref(S[0]), ptr(S[2]), ptr(S[3]), ptr(S[4]), stream)),
        "forward with direction (acattn_full_sort_ce_fwd_dir)": (2, lambda S: lib.acattn_full_sort_ce_fwd_dir(C.byref(S[0]), ptr(S[2]), ptr(S[3]), ptr(S[4]), ptr(S[5]), stream)),
        "backward with table gradient (acattn_full_sort_ce_bwd)": (3, lambda S: lib.acattn_full_sort_ce_bwd(C.byref(S[1]), ptr(S[3]), ptr(S[8]), ptr(S[2]), ptr(S[6]), ptr(S[7]), stream)),
    }
    split = "split" in _ce_products_mode(a)
    res = []
    for name, (n_prod, fn) in calls.items():
        for i in range(3):
            _lib.check(fn(sets[i % nsets]), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            rc = fn(sets[i % nsets])
            if rc:
                _lib.check(rc, name)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        flops = 2.0 * B * N * H * n_prod
        tf = flops / (us * 1e-6) / 1e12
        res.append({"call": name, "avg_call_us": round(us, 1), "products": n_prod, "algorithmic_flops_per_call": int(flops),
                    "bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 3),
                    "bf16_mfma_tflops": round(6 * tf, 1) if split else None,
                    "bf16_mfma_frac_of_2500": round(6 * tf / 2500.0, 3) if split else None})
    return {"arithmetic": _ce_products_mode(a), "calls_timed": iters, "buffer_sets": nsets,
            "note": "each call = its sweep + the small launches behind it (operand split of the batch rows, slab / partial folds)",
            "sweeps": res}


# --------------------------------------------------------------------------------------------------
# the other shapes north_star / BASELINE.json name, a handful of steps each, appended to the default line
# --------------------------------------------------------------------------------------------------
OTHER_CONFIGS = [
    # (label, overrides)
    ("reference's shipped Amazon-Beauty hyper-parameters (config/amazon-beauty.yaml:33-36): AC-SASRec B=512 L=50 d=64 4 heads 3 layers inner 128",
     dict(heads=4, layers=3, inner=128)),
    ("north_star second shape: AC-SASRec B=512 L=200 d=64 2 heads", dict(seq_len=200)),
    ("BASELINE configs[3]: AC-SASRec B=512 L=200 d=128 4 heads", dict(seq_len=200, hidden=128, heads=4, inner=512)),
    ("BASELINE configs[4]: AC-BERT4Rec (bidirectional mask) B=512 L=200 d=256 4 heads, 20000 items",
     dict(seq_len=200, hidden=256, heads=4, inner=1024, items=20000, model="AcBERT4Rec")),
]


def other_configs(a, device, steps=8, warmup=3):
    """ms/step, sequences/s and the forward kernel's roofline fraction of the other named shapes (one GPU, hipGraph replay,
    same synthetic data recipe).  Short on purpose: the default run must stay within a couple of minutes."""
    import copy
    import gc
    import ac_tsr_amd as A
    out = []
    for label, over in OTHER_CONFIGS:
        b = copy.copy(a)
        for k, v in over.items():
            setattr(b, k, v)
        torch.manual_seed(42)
        rec = {"workload": label}
        model = trainer = pool = None
        try:
            model = getattr(A, b.model)(A.DictConfig(model_config(b)), A.ItemCount(b.items)).to(device)
            if b.model == "AcBERT4Rec":
                model.cloze_on_device = True
            trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model)
            model.train()
            gen = torch.Generator().manual_seed(2000)
            pool = [synthetic_batch(b.batch, b.seq_len, b.items, gen, device) for _ in range(2)]
            trainer.enable_graph(pool[0])
            for i in range(warmup):
                trainer.train_step(pool[i % 2])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = None
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            marks[0].record()
            for i in range(steps):
                last = trainer.train_step(pool[i % 2])
                marks[i + 1].record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            # few steps on purpose, so one slow replay (a first touch, a host hiccup) would be a fifth of a mean: the
            # figure reported is the MEDIAN step; the mean over the wall clock is kept beside it
            per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
            dt = per[len(per) // 2] * 1e-3
            losses = [float(x.detach()) for x in last]
            rec.update(ms_per_step=round(dt * 1e3, 3), ms_per_step_wall_mean=round(wall * 1e3, 3), ms_per_step_max=round(per[-1], 3),
                       value=round(b.batch / dt, 1), unit="user-sequences/sec", steps=steps,
                       final_losses=[round(x, 4) for x in losses], finite=all(x == x for x in losses))
            # the same step in the opt-in trainer mode that walks the autograd graph ONCE for both losses (SURVEY 8 f4,
            # ac_tsr_amd/combined.py): same gradients; beyond L = 64 the first layer's attention backward then evaluates both
            # cotangent sets in one launch pair
            del trainer
            trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model, combined_backward=True)
            trainer.enable_graph(pool[0])
            for i in range(warmup):
                trainer.train_step(pool[i % 2])
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            marks[0].record()
            for i in range(steps):
                last = trainer.train_step(pool[i % 2])
                marks[i + 1].record()
            torch.cuda.synchronize()
            per1 = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
            dt1 = per1[len(per1) // 2] * 1e-3
            rec["combined_backward"] = {"ms_per_step": round(dt1 * 1e3, 3), "value": round(b.batch / dt1, 1),
                                        "pair_nodes": trainer.last_walk_stats.get("pair_nodes"),
                                        "finite": all(float(x.detach()) == float(x.detach()) for x in last)}
        except Exception as e:  # a shape that does not run must show up in the line, not end the run
            rec["error"] = f"{type(e).__name__}: {e}"[:300]
        del trainer, model, pool
        gc.collect()
        torch.cuda.empty_cache()
        try:
            # >= 200 launches over rotating buffer sets larger than the Infinity Cache (3 sets are 0.9-3 GB here)
            r = kernel_roofline(b, device, True, iters=70, nsets=3)
            rec["roofline"] = {k: r[k] for k in ("frac", "achieved", "avg_launch_us", "kernel", "algorithmic_bytes_per_launch",
                                                 "launches_timed", "buffer_sets", "item_length")}
        except Exception as e:
            rec["roofline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        gc.collect()
        torch.cuda.empty_cache()
        out.append(rec)
    return out


# --------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's full step (reference algorithm on the host cores)
# --------------------------------------------------------------------------------------------------
def cpu_baseline(a, state_dict, steps):
    from oracle import ac_tsr_ref as O
    # the 1-GPU box exposes every host CPU but grants a 16-CPU share: more threads only oversubscribe
    cores = min(int(os.environ.get("ACTSR_CPU_THREADS", "16")), os.cpu_count() or 1)
    torch.set_num_threads(cores)
    ecfg = O.EncoderCfg(n_layers=a.layers, n_heads=a.heads, hidden_size=a.hidden, inner_size=a.inner,
                        combine_option="gate", rich_calibrated_combine="none", seq_length=a.seq_len)
    mcfg = O.ModelCfg(enc=ecfg, n_items=a.items, max_seq_length=a.seq_len, mask_loss_weight=0.03)
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state_dict.items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-4)
    gen = torch.Generator().manual_seed(7)

    def step(B):
        batch = synthetic_batch(B, a.seq_len, a.items, gen, "cpu")
        opt.zero_grad()
        att, cal = O.calculate_loss(batch, P, mcfg, train=True)
        names = list(P)
        g_cal = torch.autograd.grad(cal, [P[n] for n in names], retain_graph=True, allow_unused=True)
        g_att = torch.autograd.grad(att, [P[n] for n in names], allow_unused=True)
        for n, gc, ga in zip(names, g_cal, g_att):
            g = ga if O.is_attack_param(n) else gc
            P[n].grad = g if g is not None else torch.zeros_like(P[n])
        opt.step()

    # bounded sample: the reference algorithm materialises [B, h, L, L, 2 dh] (21 GB at B=512, L=200, d=128), so long
    # sequences are timed on 16 sequences per step
    cpu_batch = a.batch if a.seq_len <= 64 else 16
    step(min(32, cpu_batch))  # thread-pool / allocator warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        step(cpu_batch)
    dt = time.perf_counter() - t0
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(steps * cpu_batch / dt, 1), "unit": "user-sequences/sec", "cores": torch.get_num_threads(),
            "kind": "port", "cpu_model": cpu_model, "os_cpu_count": os.cpu_count(),
            "sample": f"{steps} full steps (fwd + two-pass bwd + Adam) of B={cpu_batch} L={a.seq_len} d={a.hidden} "
                      f"h={a.heads} {a.layers} layers N={a.items}, oracle/ac_tsr_ref.py (materialised q||k concat), "
                      f"{dt:.1f} s wall"}


def main():
    # a process that dies in native code (a fault at teardown, an abort inside a library) leaves its Python stack on
    # stderr; together with the exit marker at the very end this turns "empty stdout" into a record of where it ended
    import faulthandler
    faulthandler.enable(file=sys.stderr, all_threads=True)
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    rehearsal = os.environ.get("ACATTN_BENCH_REHEARSAL") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher around it: measure N GPUs or fail -- never a silent 1-GPU number.
        # The ranks are CHILD processes (torch.distributed.run); this process has not touched the GPU (device_count() does
        # not initialise it) and only passes their output and exit code on.
        import socket
        import subprocess
        have = torch.cuda.device_count()
        if have < a.gpus and not rehearsal:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but only {have} HIP device(s) are visible "
                             f"(ACATTN_BENCH_REHEARSAL=1 rehearses the {a.gpus}-rank path on one GPU over gloo)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"bench.py: --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
        raise SystemExit(subprocess.run(cmd).returncode)
    # A process that starts while the previous GPU process of the same box is still being torn down has (rarely: 2 of
    # ~40 back-to-back launches) found no device; wait for the device before doing anything, then fail loudly.
    for attempt in range(10):
        if torch.cuda.is_available():
            break
        # never silent: if this ever happens again the record says when and how long
        print(f"bench.py: no HIP device visible yet (attempt {attempt + 1}/10, pid {os.getpid()}, t={time.time():.3f})",
              file=sys.stderr, flush=True)
        time.sleep(1.0)
    else:
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks (the 8-GPU run is the driver's): ACATTN_BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and uses gloo for the collectives.  Exercises the launch / barrier / two-graph / early-reduce path;
    # its throughput means nothing.
    if rehearsal:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local} but {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    import ac_tsr_amd as A
    from ac_tsr_amd import _lib, parallel
    _lib.load().acattn_select_forward_kernel(["auto", "stream", "staged", "general"].index(a.fwd_kernel))
    _lib.load().acattn_select_backward_kernel(["auto", "stream", "row"].index(a.bwd_kernel))
    from ac_tsr_amd import tail as _tail
    _tail.FUSED_KERNEL = a.tail == "fused"
    from ac_tsr_amd import linear as _linear
    _linear.FUSED_PROJECTIONS = a.projections == "fused"

    if world > 1:
        if rehearsal:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            parallel.init_distributed("nccl")
    import torch.distributed as dist
    backend = dist.get_backend() if world > 1 else None
    ranks = dist.get_world_size() if world > 1 else 1

    if a.kernel_only:
        kinds = a.kernel_kinds.split(",")
        out = {}
        if "ragged" in kinds:
            out["roofline"] = kernel_roofline(a, device, True, a.kernel_iters)
        if "full" in kinds:
            out["roofline_full_length"] = kernel_roofline(a, device, True, a.kernel_iters, full_length=True)
        if "spatial" in kinds:
            out["roofline_spatial_only"] = kernel_roofline(a, device, False, a.kernel_iters)
        print(json.dumps(out), flush=True)
        return

    torch.manual_seed(42)  # config/*.yaml seed: 42 -- identical initial parameters on every rank
    model = getattr(A, a.model)(A.DictConfig(model_config(a)), A.ItemCount(a.items)).to(device)
    if a.model == "AcBERT4Rec":
        model.cloze_on_device = True  # the cloze batch is built with tensor ops: the whole step replays as a graph
    parallel.broadcast_parameters(model)
    sync = parallel.GradSynchronizer.for_two_pass_model(model, collective=a.dp_collective) if (world > 1 or a.force_grad_sync) else None
    trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model, grad_sync=sync,
                                    combined_backward=a.combined_backward)
    model.train()
    gen = torch.Generator().manual_seed(1000 + rank)
    pool = [synthetic_batch(a.batch, a.seq_len, a.items, gen, device) for _ in range(8)]
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if rank == 0 else None

    if not a.no_graph:
        trainer.enable_graph(pool[0])  # capture happens before the warm-up steps; every timed step replays it
    for i in range(a.warmup):
        trainer.train_step(pool[i % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    # one event per step boundary (recorded inside the timed region: microseconds of host time per step) for the
    # median / min / max of the step time; `value` and `ms_per_step` stay the wall clock over all K steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    marks[0].record()
    for i in range(a.steps):
        last = trainer.train_step(pool[i % len(pool)])
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    att, cal = (float(x.detach()) for x in last)
    if not (att == att and cal == cal):
        raise SystemExit("Training loss is nan")

    # The same step with the reference's full schedule (nothing that DESIGN.md section 5 lists as provably dead is
    # skipped): reported next to `value` so that both can be judged.  Single GPU only, short, after the timed region.
    full_ms = None
    if world == 1 and not a.no_full_schedule:
        model.step_state.prune_dead_work = False
        try:
            full_trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model)
            if not a.no_graph:
                full_trainer.enable_graph(pool[0])
            for i in range(3):
                full_trainer.train_step(pool[i % len(pool)])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_full = max(5, a.steps // 3)
            for i in range(n_full):
                full_trainer.train_step(pool[i % len(pool)])
            torch.cuda.synchronize()
            full_ms = (time.perf_counter() - t1) / n_full * 1e3
        finally:
            model.step_state.prune_dead_work = True

    if rank == 0:
        res = {
            "metric": "user-sequences/sec fwd+bwd, AC-SASRec L=50 d=64, 1/2/4/8 MI355X",
            "value": round(world * a.batch * a.steps / dt, 1), "unit": "user-sequences/sec", "n_gpus": world,
            # ranks of the process group the timed steps exchanged gradients over, and its backend ("nccl" is RCCL on ROCm)
            "rccl_ranks": ranks if backend == "nccl" else (None if world == 1 else 0), "collective_backend": backend,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
            "ms_per_step_max": round(per_step[-1], 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"{a.model} full training step (spatial + adversarial calibrators, BASELINE "
                             f"configs[{ {None: 2, 'cfg2': 2, 'cfg3': 2, 'cfg4': 3, 'cfg5': 4}[a.config] }]"
                             + ("; configs[1] spatial-only kernel under roofline_spatial_only" if a.config in (None, "cfg2", "cfg3") else "")
                             + f"), synthetic {a.items}-item catalogue, B={a.batch}/GPU L={a.seq_len} d={a.hidden} "
                             f"h={a.heads} {a.layers} layers inner={a.inner}, CE loss, two-pass backward + Adam"),
                "global_batch": world * a.batch, "seq_len": a.seq_len, "hidden": a.hidden, "heads": a.heads,
                "parallelism": f"dp{world}" + (" (rehearsal: all ranks on one GPU, gloo)" if rehearsal else ""), "launch": "eager" if a.no_graph else "hipGraph replay per step",
                "backward": "one combined walk (opt-in)" if a.combined_backward else "two walks (reference protocol, trainer.py:672-686)",
                "final_losses": [round(att, 4), round(cal, 4)],
                "ce_products": _ce_products_mode(a),
                "reference_schedule": None if full_ms is None else {
                    "ms_per_step": round(full_ms, 3), "value": round(a.batch / full_ms * 1e3, 1),
                    "note": "same step computing also the provably dead work the reference computes (DESIGN.md 5)"}},
        }
        res["roofline"] = kernel_roofline(a, device, True, a.kernel_iters)
        # the same kernel when every sequence has all L items ("also report the all-L worst case"): the
        # algorithmic bytes are the same, no key tile past the last item can be skipped
        res["roofline_full_length"] = kernel_roofline(a, device, True, a.kernel_iters, full_length=True)
        res["roofline_spatial_only"] = kernel_roofline(a, device, False, a.kernel_iters)
        if a.model != "AcBERT4Rec":
            try:
                res["roofline_ce"] = ce_roofline(a, device)
            except Exception as e:  # (an extra of the line: never the reason a bench run fails)
                res["roofline_ce"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and a.config is None and not a.no_other_configs and (a.batch, a.seq_len, a.hidden, a.heads) == (512, 50, 64, 2):
            res["other_configs"] = other_configs(a, device)
        if world == 1 and not a.no_cpu_baseline and a.model == "ACSASRec":
            res["cpu_baseline"] = cpu_baseline(a, init_state, a.cpu_steps)
        else:
            res["cpu_baseline"] = None
        # flushed at once: stdout is a pipe or a file here (block-buffered), and a line still sitting in the buffer is
        # lost if anything goes wrong while the process is being torn down (3 of ~60 runs in this round ended with an
        # empty stdout; none of the 50 runs whose stderr was kept did)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # release the captured graphs and everything they hold before the interpreter starts tearing modules down
    try:
        del full_trainer
    except NameError:
        pass
    del trainer, sync, model, pool
    import gc
    gc.collect()
    torch.cuda.synchronize()
    # last action of the process' own code: everything after this line on stderr is interpreter / runtime teardown
    print(f"bench.py: exit marker rank={rank} pid={os.getpid()} status=ok", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
