"""CPU suite: the C-ABI library builds/loads, exports every declared symbol, and its ctypes mirror
matches the C layout.  No compute is launched (there is no GPU here); only the argument-validation
paths, which return before any HIP call, are exercised."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from ac_tsr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(acattn_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert "acattn_calibrated_attention_fwd" in names and "acattn_calibrated_attention_bwd" in names
    for n in names:
        assert hasattr(lib, n), f"libacattn.so does not export {n}"
        assert n in _lib.SYMBOLS, f"ctypes binding lacks {n}"
    assert sorted(_lib.SYMBOLS) == names


def test_ce_products_mode_query_needs_no_gpu(lib):
    # (acattn_full_sort_ce_products: 0 exact fp32 MFMA, 1 default, 2 split sweeps everywhere; other values only query)
    old = lib.acattn_full_sort_ce_products(-1)
    assert old in (0, 1, 2)
    assert lib.acattn_full_sort_ce_products(2) == old and lib.acattn_full_sort_ce_products(old) == 2


def test_abi_version(lib):
    assert lib.acattn_abi_version() == _lib.ABI_VERSION


def test_ctypes_layout_matches_c(tmp_path):
    """Compile a tiny C program against include/acattn.h and compare sizeof/offsetof with ctypes."""
    fields = {"acattn_problem": _lib.Problem, "acattn_fwd_out": _lib.FwdOut, "acattn_bwd_io": _lib.BwdIO,
              "acattn_ce_problem": _lib.CeProblem, "acattn_ln_problem": _lib.LnProblem,
              "acattn_embed_problem": _lib.EmbedProblem, "acattn_proj_problem": _lib.ProjProblem,
              "acattn_proj_out": _lib.ProjOut, "acattn_proj_bwd_io": _lib.ProjBwdIO,
              "acattn_tail_problem": _lib.TailProblem, "acattn_tail_saved": _lib.TailSaved,
              "acattn_tail_bwd_io": _lib.TailBwdIO, "acattn_adam_group": _lib.AdamGroup}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "acattn.h"', 'int main(void){']
    for cname, cls in fields.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(l.split() for l in out.strip().splitlines())
    for cname, cls in fields.items():
        assert int(got[cname]) == C.sizeof(cls)
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_algorithmic_bytes_contracts(lib):
    from ac_tsr_amd import ops
    # contract A = 7*T_LH + (1+h)*T_LL, contract A' = 4*T_LH
    assert ops.fwd_algorithmic_bytes(1, 50, 64, 2) == 119600
    assert ops.fwd_algorithmic_bytes(512, 50, 64, 2) == 512 * 119600
    assert ops.fwd_algorithmic_bytes(1, 200, 128, 4) == 1516800
    assert ops.fwd_algorithmic_bytes(1, 50, 64, 2, adversarial=False) == 51200


def test_validation_errors_without_gpu(lib):
    p, o = _lib.Problem(), _lib.FwdOut()
    p.B, p.L, p.H, p.n_heads = 2, 50, 64, 3
    assert lib.acattn_calibrated_attention_fwd(C.byref(p), C.byref(o), None) < 0
    assert b"not a multiple" in lib.acattn_last_error()
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    p.n_heads = 2
    assert lib.acattn_calibrated_attention_fwd(C.byref(p), C.byref(o), None) < 0
    assert b"non-NULL" in lib.acattn_last_error()
    p.L = 500
    assert lib.acattn_calibrated_attention_fwd(C.byref(p), C.byref(o), None) < 0
    assert b"sequence length" in lib.acattn_last_error()
    # the layer-level launches validate before touching the device as well
    tp, ts = _lib.TailProblem(), _lib.TailSaved()
    tp.rows, tp.H, tp.I = 16, 96, 256
    assert lib.acattn_layer_tail_fwd(C.byref(tp), C.byref(ts), None) < 0 and b"hidden_size" in lib.acattn_last_error()
    assert lib.acattn_layer_tail_supported(64, 256) == 1 and lib.acattn_layer_tail_supported(128, 512) == 1
    assert lib.acattn_layer_tail_supported(256, 1024) == 1 and lib.acattn_layer_tail_supported(96, 256) == 0
    assert lib.acattn_layer_tail_bwd_workspace_bytes(128, 512) == 4 * (2 * 128 * 512 + 128 * 128)
    assert lib.acattn_layer_tail_bwd_workspace_bytes(64, 256) == 4 * (2 * 64 * 256 + 64 * 64)
    assert lib.acattn_layer_tail_bwd_partial_rows_for(100, 128) == 7 and lib.acattn_layer_tail_bwd_partial_rows_for(102400, 64) == 6400
    # the row-sum form of the mask penalty (round 3) validates before touching the device too
    assert lib.acattn_mask_penalty_rows(None, 2, 2, 50, None, None) < 0 and b"non-NULL" in lib.acattn_last_error()
    assert lib.acattn_attacked_loss_finish_rows(None, 4, None, 2, 16, 0.03, None, None, 0, None) < 0
    assert lib.acattn_mask_penalty_drows(None, None, 0.5, 16, None, 2, None) < 0
    pp, po = _lib.ProjProblem(), _lib.ProjOut()
    pp.rows, pp.H, pp.G = 16, 64, 300
    assert lib.acattn_projections_fwd(C.byref(pp), C.byref(po), None) < 0 and b"gate" in lib.acattn_last_error()
    assert lib.acattn_projections_supported(64, 50) == 1 and lib.acattn_projections_supported(128, 50) == 1
    assert lib.acattn_projections_supported(256, 200) == 1 and lib.acattn_projections_supported(96, 50) == 0
    pp.H, pp.G = 128, 100
    assert lib.acattn_projections_bwd_workspace_bytes(C.byref(pp)) == 4 * (5 * 128 * 128 + 128 * 112)
    pp.H, pp.G = 64, 300


def test_cpu_tensors_fail_loudly(lib):
    import torch

    from ac_tsr_amd import AttentionConfig, StructuredMask, calibrated_attention
    x = torch.zeros(1, 50, 64)
    with pytest.raises(_lib.AcattnError):
        calibrated_attention(x, x, x, x, x, torch.zeros(1, 50, 50), StructuredMask(torch.ones(1, 50, dtype=torch.uint8)),
                             AttentionConfig(n_heads=2))


def test_module_surface_matches_reference_state_dict_keys():
    """State-dict keys and shapes of SURVEY.md section 8b (H=64, h=2, inner=256, L=50)."""
    from tests._golden import Case
    import ac_tsr_amd as A
    c = Case("model_eval")
    cfg = c.model_cfg()
    m = A.ACSASRec(A.DictConfig(n_layers=cfg.enc.n_layers, n_heads=cfg.enc.n_heads, hidden_size=cfg.enc.hidden_size,
                                inner_size=cfg.enc.inner_size, hidden_dropout_prob=0.5, attn_dropout_prob=0.5,
                                hidden_act='gelu', layer_norm_eps=1e-12, initializer_range=0.02, loss_type='CE',
                                combine_option='gate', two_level=True, use_order=True, use_distance=True,
                                rich_calibrated_combine='none', mask_loss_weight=0.03), A.ItemCount(cfg.n_items))
    ref = c.params()
    sd = m.state_dict()
    assert sorted(sd) == sorted(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict(ref)  # a reference checkpoint loads as is
    attack = [n for n, _ in m.named_parameters() if A.is_attack_param(n)]
    assert len(attack) == 4 * cfg.enc.n_layers


def test_dispatcher_operators_are_registered_with_shape_functions():
    """ac_tsr_amd/dispatch.py registers the two attention entry points with torch's dispatcher; without a GPU the Meta
    implementations (shapes only) are what can run, and a CPU tensor must be refused, not computed."""
    import torch
    from ac_tsr_amd import dispatch  # noqa: F401
    B, L, H, nh = 3, 50, 64, 2
    q = torch.empty(B, L, H, device="meta")
    kv = torch.empty(B, L, dtype=torch.uint8, device="meta")
    w, b1 = torch.empty(2 * H // nh, device="meta"), torch.empty(1, device="meta")
    gate = torch.empty(B, L, L, device="meta")
    ctx_a, ctx_c, M, stats, pen = torch.ops.acattn.calibrated_attention_fwd(q, q, q, q, q, gate, kv, True, w, b1, w, b1, b1, nh, 0.5, 1,
                                                                        None, False, None, True)
    assert ctx_a.shape == ctx_c.shape == (B, L, H) and M.shape == (B, nh, L, L) and stats.shape == (B, nh, L, _lib.NSTAT)
    assert pen.shape == (B, nh, (L + 15) // 16)
    outs = torch.ops.acattn.calibrated_attention_bwd(q, q, q, q, q, gate, kv, True, w, b1, w, b1, b1, nh, 0.5, 1, None, False, M,
                                                     stats, q, q, M, None, None, False)
    assert [tuple(t.shape) for t in outs] == [(B, L, H)] * 5 + [(B, nh, L, L), (B * nh, 4 * (H // nh) + 4)]
    spatial = torch.ops.acattn.calibrated_attention_fwd(q, q, q, None, None, None, kv, True, w, b1, w, b1, b1, nh, 0.0, 1, None,
                                                        False, None, False)
    assert spatial[1].shape == (B, L, H) and spatial[0].numel() == 0
    cpu = torch.zeros(B, L, H)
    with pytest.raises(Exception):  # no CPU kernel is registered: the dispatcher refuses
        torch.ops.acattn.calibrated_attention_fwd(cpu, cpu, cpu, cpu, cpu, torch.zeros(B, L, L), torch.ones(B, L, dtype=torch.uint8),
                                                  True, torch.zeros(64), torch.zeros(1), torch.zeros(64), torch.zeros(1),
                                                  torch.zeros(1), nh, 0.5, 1, None, False, None, True)



def test_spilling_streaming_forward_instantiations_are_the_ones_the_gpu_suite_covers():
    """Round 3 recorded wrong tiles from two work-in-progress builds of the streaming forward whose registers spilled;
    round 4 found every committed source right when forced to spill -- but a compiler / flag / source change
    that makes ANOTHER instantiation spill should not pass unnoticed: the set of spilling instantiations in the built
    objects must be the set that tests/test_hip_onehop.py runs against the oracle at >= 2 waves per SIMD."""
    import glob
    import os
    import shutil
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources as KR
    if not (shutil.which("objcopy") and os.path.exists(os.path.join(KR.LLVM, "clang-offload-bundler"))):
        pytest.skip("no binutils / ROCm LLVM tools here")
    objs = sorted(glob.glob(os.path.join(KR.ROOT, "ac_tsr_amd", "csrc", "acattn_fwd_stream_dh*.o")))
    if not objs:
        pytest.skip("objects not built in-tree (library built elsewhere)")
    # <64, 13, .., false>: test_hip_onehop.py id dh64_L200_spilling (3,328 waves); <32, 4, true, false, true>: the general-rate
    # form of the benchmark kernel, run at B = 512 with p_drop = 0 by test_hip_forward.py::test_fast_training_kernel_equals_general_kernel
    covered = {"<64, 13, true, true, false>", "<64, 13, true, false, false>", "<32, 4, true, false, true>"}
    spilling = set()
    n = 0
    for o in objs:
        ks = KR.code_object_kernels(o)
        names = KR.demangle([k["name"] for k in ks])
        for k in ks:
            n += 1
            if k.get("vgpr_spill_count", 0) or k.get("sgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0):
                spilling.add("<" + names[k["name"]].split("<", 1)[1].split(">")[0] + ">")
    assert n >= 48
    assert spilling <= covered, f"streaming-forward instantiations that spill without a >= 2 waves/SIMD oracle test: {spilling - covered}"


def test_deferred_reductions_are_only_armed_inside_a_pass_that_will_flush_them():
    """StepState.deferring() (state.py): on only when the trainer enabled it AND the running pass was opened with the leaves
    of its walk (that pass flushes at its end); never on the frozen default, in the one-walk mode, or in a pass without
    leaves -- otherwise a caller could be left holding unwritten gradient tensors.  Host logic only."""
    import torch
    from ac_tsr_amd.state import DEFAULT, StepState
    st = StepState()
    leaves = [torch.nn.Parameter(torch.zeros(3))]
    assert not st.deferring()
    with st.calibrated_pass(leaves):
        assert not st.deferring()  # the trainer has not enabled it
    st.defer_reductions = True
    assert not st.deferring()  # no pass
    with st.calibrated_pass():
        assert not st.deferring()  # nobody would flush
    with st.attack_pass(leaves):
        assert st.deferring()
        st.combined = object()
        assert not st.deferring()  # the one-walk mode deposits gradients by hand
        st.combined = None
        with st.calibrated_pass():  # a nested pass without leaves does not inherit the outer one's
            assert not st.deferring()
        assert st.deferring()
    assert not st.deferring() and st._deferred == [] and st._flush_leaves is None
    # an exception inside the pass drops the queue instead of flushing it
    try:
        with st.calibrated_pass(leaves):
            st._deferred.append({"dw": 1, "db": None})
            raise KeyError("boom")
    except KeyError:
        pass
    assert st._deferred == [] and st.pass_mode is None
    # watch() ignores tensors that are not the last deferred sum's output
    with st.calibrated_pass(leaves):
        st.watch(torch.zeros(2), torch.zeros(1))
    assert not DEFAULT.deferring()
