"""ctypes binding of libacattn.so -- the C ABI declared in include/acattn.h.

The structures below mirror the header field for field.  There is NO fallback: if the shared
library has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C ac_tsr_amd/csrc``) importing any compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("ACATTN_LIB") or os.path.join(CSRC, "libacattn.so")  # ACATTN_LIB: experiments only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "acattn.h")

ABI_VERSION = 29
MAX_MASKS = 8  # ACATTN_MAX_MASKS
NSTAT = 8
MASK_STRUCTURED, MASK_DENSE_LL, MASK_DENSE_L = 0, 1, 2
COMBINE = {"fixed": 0, "gate": 1, "annealing": 2}
RICH = {"none": 0, "fixed": 1, "trainable": 2}
RNG_EXPLICIT, RNG_COUNTER = 0, 1
FWD_AUTO, FWD_STREAM, FWD_STAGED, FWD_GENERAL = 0, 1, 2, 3
BWD_AUTO, BWD_STREAM, BWD_ROW = 0, 1, 2

_f = C.c_void_p  # every device pointer travels as an integer address


class Problem(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("n_heads", C.c_int32),
        ("q", _f), ("k", _f), ("v", _f), ("qa", _f), ("ka", _f), ("gate_logits", _f),
        ("mask_mode", C.c_int32), ("causal", C.c_int32), ("key_valid", _f), ("mask", _f),
        ("w_order", _f), ("b_order", _f), ("w_dist", _f), ("b_dist", _f), ("scalar", _f),
        ("adversarial", C.c_int32), ("combine_option", C.c_int32), ("anneal_rate", C.c_float),
        ("two_level", C.c_int32), ("rich_combine", C.c_int32), ("rich_ratio", _f),
        ("rng_mode", C.c_int32), ("p_drop", C.c_float), ("noise", _f), ("keep_after", _f), ("keep_before", _f),
        ("keep_mask", _f), ("seed", C.c_uint64), ("seed_device", _f), ("affine", _f), ("gate_is_prob", C.c_int32),
    ]


class FwdOut(C.Structure):
    _fields_ = [
        ("ctx_attacked", _f), ("ctx_calibrated", _f), ("attack_mask", _f), ("row_stats", _f),
        ("after_spatial", _f), ("before_spatial", _f), ("perturbed_attention", _f), ("calibrated_attention", _f),
        ("penalty_part", _f),
    ]


class BwdIO(C.Structure):
    _fields_ = [
        ("attack_mask", _f), ("row_stats", _f),
        ("d_ctx_attacked", _f), ("d_ctx_calibrated", _f), ("d_attack_mask", _f),
        ("dq", _f), ("dk", _f), ("dv", _f), ("dqa", _f), ("dka", _f), ("dgate_logits", _f),
        ("dw_order_part", _f), ("dw_dist_part", _f), ("dsmall_part", _f), ("part_stride", C.c_int32),
        ("active_qblocks", _f), ("attack_only", C.c_int32), ("workspace", _f), ("read_rows", _f),
        ("n_read_rows", C.c_int32), ("d_penalty_part", _f), ("dgate_summed", C.c_int32),
        ("d_ctx_calibrated2", _f), ("d_penalty_part2", _f), ("dqa2", _f), ("dka2", _f),
    ]


class CeProblem(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("out", _f), ("table", _f), ("target", _f),
                ("coef_is_scalar", C.c_int32), ("coef_scale", C.c_float)]


class LnProblem(C.Structure):
    _fields_ = [("rows", C.c_int32), ("H", C.c_int32), ("residual_rows", C.c_int32), ("z", _f), ("residual", _f),
                ("gamma", _f), ("beta", _f), ("eps", C.c_float), ("p_drop", C.c_float), ("keep", _f),
                ("seed", C.c_uint64), ("seed_device", _f)]


LN_BWD_GRID = 512


class ProjProblem(C.Structure):
    _fields_ = [("rows", C.c_int32), ("H", C.c_int32), ("G", C.c_int32), ("x", _f), ("wq", _f), ("bq", _f), ("wk", _f),
                ("bk", _f), ("wv", _f), ("bv", _f), ("waq", _f), ("baq", _f), ("wak", _f), ("bak", _f), ("wg", _f),
                ("bg", _f), ("w_order", _f), ("b_order", _f), ("w_dist", _f), ("b_dist", _f), ("n_heads", C.c_int32),
                ("L", C.c_int32)]


class ProjOut(C.Structure):
    _fields_ = [("mq", _f), ("mk", _f), ("mv", _f), ("qa", _f), ("ka", _f), ("gate", _f), ("affine", _f),
                ("gate_prob", C.c_int32)]


class ProjBwdIO(C.Structure):
    _fields_ = [("dmq", _f), ("dmk", _f), ("dmv", _f), ("dqa", _f), ("dka", _f), ("dgate", _f), ("dmq_total", _f),
                ("dmk_total", _f), ("dx", _f), ("dx_init", _f), ("workspace", _f)]


ADAM_MAX_TENSORS = 64


class AdamGroup(C.Structure):
    _fields_ = [("n_tensors", C.c_int32), ("param", _f * ADAM_MAX_TENSORS), ("grad", _f * ADAM_MAX_TENSORS),
                ("exp_avg", _f * ADAM_MAX_TENSORS), ("exp_avg_sq", _f * ADAM_MAX_TENSORS), ("step", _f * ADAM_MAX_TENSORS),
                ("numel", C.c_int64 * ADAM_MAX_TENSORS)]


class TailProblem(C.Structure):
    _fields_ = [("rows", C.c_int32), ("H", C.c_int32), ("I", C.c_int32), ("ctx", _f), ("x", _f), ("wd", _f), ("bd", _f),
                ("g1", _f), ("b1", _f), ("w1", _f), ("bb1", _f), ("w2", _f), ("bb2", _f), ("g2", _f), ("b2", _f),
                ("eps1", C.c_float), ("eps2", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("keep1", _f),
                ("keep2", _f), ("seed1", C.c_uint64), ("seed2", C.c_uint64), ("seed_device", _f), ("src_index", _f),
                ("src_R", C.c_int32), ("src_L", C.c_int32)]


class TailSaved(C.Structure):
    _fields_ = [("h1", _f), ("st1", _f), ("a", _f), ("act", _f), ("h3", _f), ("st2", _f), ("out", _f),
                ("gelu_grad", _f)]


class TailBwdIO(C.Structure):
    _fields_ = [("d_out", _f), ("d_ctx", _f), ("d_x", _f), ("d_h1", _f), ("d_h2", _f), ("d_h3", _f), ("dgb_part", _f),
                ("workspace", _f)]


class EmbedProblem(C.Structure):
    _fields_ = [("rows", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("n_table_rows", C.c_int64), ("idx", _f),
                ("table", _f), ("pos", _f), ("gamma", _f), ("beta", _f), ("eps", C.c_float), ("p_drop", C.c_float),
                ("keep", _f), ("seed", C.c_uint64), ("seed_device", _f), ("nonzero_out", _f), ("hot_id_plus1", C.c_int64)]


EMBED_BWD_CHUNKS = 8
PENALTY_WS_FLOATS = 1024
WGRAD_MAX_GROUP = 8
WGRAD_MAX_REDUCE = 32
SUMROWS_MAX_DEFER = 8

# name -> (restype, argtypes); must list every symbol include/acattn.h declares (tests check this)
SYMBOLS = {
    "acattn_abi_version": (C.c_int, []),
    "acattn_calibrated_attention_bwd_gate_summed": (C.c_int, [C.POINTER(Problem), C.POINTER(BwdIO)]),
    "acattn_calibrated_attention_bwd_pair_supported": (C.c_int, [C.POINTER(Problem), C.POINTER(BwdIO)]),
    "acattn_last_error": (C.c_char_p, []),
    "acattn_fwd_algorithmic_bytes": (C.c_int64, [C.POINTER(Problem)]),
    "acattn_calibrated_attention_fwd": (C.c_int, [C.POINTER(Problem), C.POINTER(FwdOut), C.c_void_p]),
    "acattn_calibrated_attention_bwd": (C.c_int, [C.POINTER(Problem), C.POINTER(BwdIO), C.c_void_p]),
    "acattn_spatial_affines": (C.c_int, [C.POINTER(Problem), _f, C.c_void_p]),
    "acattn_full_sort_ce_workspace_bytes": (C.c_int64, [C.POINTER(CeProblem)]),
    "acattn_full_sort_ce_fwd": (C.c_int, [C.POINTER(CeProblem), _f, _f, _f, C.c_void_p]),
    "acattn_full_sort_ce_fwd_dir": (C.c_int, [C.POINTER(CeProblem), _f, _f, _f, _f, C.c_void_p]),
    "acattn_full_sort_ce_bwd": (C.c_int, [C.POINTER(CeProblem), _f, _f, _f, _f, _f, C.c_void_p]),
    "acattn_full_sort_ce_products": (C.c_int, [C.c_int]),
    "acattn_dropout_add_layernorm_fwd": (C.c_int, [C.POINTER(LnProblem), _f, _f, C.c_void_p]),
    "acattn_dropout_add_layernorm_bwd": (C.c_int, [C.POINTER(LnProblem), _f, _f, _f, _f, _f, C.c_void_p]),
    "acattn_embed_layernorm_fwd": (C.c_int, [C.POINTER(EmbedProblem), _f, _f, C.c_void_p]),
    "acattn_embed_layernorm_bwd": (C.c_int, [C.POINTER(EmbedProblem), _f, _f, C.c_int64, _f, _f, _f, C.c_void_p]),
    "acattn_projections_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "acattn_projections_bwd_workspace_bytes": (C.c_int64, [C.POINTER(ProjProblem)]),
    "acattn_projections_fwd": (C.c_int, [C.POINTER(ProjProblem), C.POINTER(ProjOut), C.c_void_p]),
    "acattn_projections_bwd": (C.c_int, [C.POINTER(ProjProblem), C.POINTER(ProjBwdIO), C.c_void_p]),
    "acattn_layer_tail_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "acattn_layer_tail_fwd": (C.c_int, [C.POINTER(TailProblem), C.POINTER(TailSaved), C.c_void_p]),
    "acattn_layer_tail_bwd": (C.c_int, [C.POINTER(TailProblem), C.POINTER(TailSaved), C.POINTER(TailBwdIO), C.c_void_p]),
    "acattn_layer_tail_bwd_partial_rows": (C.c_int32, [C.c_int32]),
    "acattn_layer_tail_bwd_partial_rows_for": (C.c_int32, [C.c_int32, C.c_int32]),
    "acattn_layer_tail_bwd_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "acattn_select_layer_tail_blocks": (C.c_int, [C.c_int]),
    "acattn_adam_step": (C.c_int, [C.POINTER(AdamGroup), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _f,
                                   C.c_void_p]),
    "acattn_dense_ce_fwd": (C.c_int, [_f, C.c_int64, C.c_int64, _f, _f, _f, C.c_void_p]),
    "acattn_dense_ce_bwd": (C.c_int, [_f, _f, _f, _f, C.c_int64, C.c_int64, _f, C.c_void_p]),
    "acattn_step_inputs": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "acattn_sum_rows": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "acattn_sum_rows_pair": (C.c_int, [_f, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "acattn_mask_penalty_fwd": (C.c_int, [_f, C.c_int64, _f, _f, C.c_void_p]),
    "acattn_mask_penalty_partial": (C.c_int, [_f, C.c_int64, _f, C.c_void_p]),
    "acattn_mask_penalty_partial_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, _f, C.c_void_p]),
    "acattn_mask_penalty_bwd_scaled_multi": (C.c_int, [C.c_void_p, _f, _f, C.c_float, C.c_int64, C.c_void_p, C.c_int32,
                                                       C.c_void_p]),
    "acattn_attacked_loss_finish": (C.c_int, [_f, C.c_int32, _f, C.c_int32, C.c_int64, C.c_float, _f, _f, C.c_int32,
                                              C.c_void_p]),
    "acattn_mask_penalty_rows": (C.c_int, [_f, C.c_int32, C.c_int32, C.c_int32, _f, C.c_void_p]),
    "acattn_attacked_loss_finish_rows": (C.c_int, [_f, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, _f, _f, C.c_int32,
                                                  C.c_void_p]),
    "acattn_mask_penalty_drows": (C.c_int, [_f, _f, C.c_float, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "acattn_mask_penalty_drows_dir": (C.c_int, [_f, _f, C.c_float, C.c_int32, C.c_void_p, C.c_int32, _f, _f, C.c_int32, C.c_void_p]),
    "acattn_mask_penalty_bwd_scaled": (C.c_int, [_f, _f, _f, C.c_float, C.c_int64, _f, C.c_void_p]),
    "acattn_mask_penalty_bwd": (C.c_int, [_f, _f, _f, C.c_int64, _f, C.c_void_p]),
    "acattn_linear_wgrad_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "acattn_linear_wgrad_grouped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int32, C.c_int64, _f, C.c_void_p]),
    "acattn_linear_wgrad": (C.c_int, [_f, _f, C.c_int64, C.c_int32, C.c_int32, _f, _f, _f, C.c_void_p]),
    "acattn_linear_wgrad_grouped_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "acattn_linear_wgrad_reduce_many": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_int32, C.c_void_p]),
    "acattn_select_forward_kernel": (C.c_int, [C.c_int]),
    "acattn_select_backward_kernel": (C.c_int, [C.c_int]),
    "acattn_calibrated_attention_bwd_workspace_bytes": (C.c_int64, [C.POINTER(Problem)]),
    "acattn_rng_materialize": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_float, _f, _f, _f, _f,
                                         C.c_void_p]),
}

_lib = None


class AcattnError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libacattn.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    res = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise AcattnError("building libacattn.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AcattnError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`make -C ac_tsr_amd/csrc` (or __graft_entry__.build()). There is no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.acattn_abi_version()
    if got != ABI_VERSION:
        raise AcattnError(f"libacattn.so ABI {got} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    msg = load().acattn_last_error().decode()
    if rc < 0 and "not a multiple of the number of attention" in msg:
        raise ValueError(msg)  # same exception type as recbole/model/layers.py:618-622
    if rc < 0 and ("unknown combine_option" in msg or "unknown rich_calibrated_combine" in msg):
        raise KeyError(msg)  # recbole/model/layers.py:894-895, 935-936
    raise AcattnError(f"{what} failed (rc={rc}): {msg}")
