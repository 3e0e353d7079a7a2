"""ctypes binding of libgf_amd.so (C ABI declared in include/gf_amd.h).

The product path has NO fallback: if the shared library is missing or a launcher returns
an error, a RuntimeError is raised.  ``build()`` compiles the library in-tree with hipcc
(cross-compiles for gfx950 without a GPU)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgf_amd.so")
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 17      # == GF_AMD_ABI_VERSION in include/gf_amd.h (bumped with every signature / workspace change)

_c = ctypes
_P, _I, _F, _L, _D = _c.c_void_p, _c.c_int, _c.c_float, _c.c_int64, _c.c_double
_S = _c.POINTER(_c.c_int64)

# name -> argument types; mirrors include/gf_amd.h one to one
SIGNATURES = {
    "gf_abi_version": [],
    "gf_attn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _S, _S, _S, _S, _F, _I, _P],
    "gf_attn_fwd_ex": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _S, _S, _S, _S, _F, _I, _I, _P, _P],
    "gf_attn_cross_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _S, _S, _S, _S, _S, _S, _F, _I, _P],
    "gf_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                    _S, _S, _S, _S, _S, _S, _S, _S, _F, _I, _P],
    "gf_attn_bwd_acc": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I,
                        _S, _S, _S, _S, _S, _S, _S, _S, _F, _I, _I, _P],
    "gf_rows_lse": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_rows_argmax": [_P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_assign_write": [_P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_dual_softmax_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _F, _P, _I, _I, _I, _I, _I, _P],
    "gf_head_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_filter_matches": [_P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P],
    "gf_sinkhorn_plan": [_I, _I, _I, _I, _I, _I, _P],
    "gf_sinkhorn_ws_bytes": [_I, _I, _I, _I],
    "gf_sinkhorn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_sinkhorn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_line_csr": [_P, _P, _P, _I, _I, _I, _P],
    "gf_line_gather": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_line_segsum": [_P, _L, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gf_line_expand": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gf_rows_gather": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_line_pair_scores": [_P, _P, _P, _I, _I, _I, _I, _P],
    "gf_dense_rowcol": [_P, _L, _L, _P, _P, _I, _I, _I, _I, _P],
    "gf_dense_assign": [_P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _P],
    "gf_dense_assign_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gf_bgemm": [_P, _P, _P, _I, _I, _I, _I, _S, _S, _S, _F, _I, _P],
    "gf_multi_cast_transpose": [_P, _I, _I, _I, _P],
    "gf_cast_entry_bytes": [],
    "gf_fold_entry_bytes": [],
    "gf_fold_linear_fwd": [_P, _I, _I, _P],
    "gf_fold_linear_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_adam_entry_bytes": [],
    "gf_multi_adam": [_P, _I, _P, _P, _P, _P, _D, _D, _F, _F, _P],
    "gf_weight_grad_map": [_P, _P, _P, _P, _P, _F, _I, _I, _P],
    "gf_colsum_f32": [_P, _P, _P, _I, _I, _I, _P],
    "gf_colsum_ws_floats": [_I, _I],
    "gf_small_dw": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gf_small_dw_ws_floats": [_I, _I],
    "gf_small_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "gf_gemm": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _L, _L, _I, _P],
    "gf_gemm_res2": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _I, _P],
    "gf_linear_dw_ws_bytes": [_I, _I, _I],
    "gf_linear_dw": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gf_linear_dw2": [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "gf_bn_nblk": [_I],
    "gf_bn_stats": [_P, _P, _I, _I, _I, _P],
    "gf_bn_act_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gf_bn_bwd_stats": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gf_bn_bwd_dx": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gf_bn_finalize_fwd": [_P, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    "gf_bn_replay_running": [_P, _I, _I, _F, _F, _P, _P, _P, _P],
    "gf_bn_finalize_bwd": [_P, _I, _I, _F, _P, _P, _P, _P, _P],
    "gf_bn_pack_sums": [_P, _I, _I, _I, _F, _P, _P, _P],
    "gf_bn_finalize_sets_fwd": [_P, _I, _I, _F, _F, _P, _P, _P, _P],
    "gf_bn_finalize_sets_bwd": [_P, _P, _I, _I, _P, _P],
    "gf_bn_replay_running_n": [_P, _P, _I, _I, _F, _P, _P, _P, _P],
    "gf_gt_nn": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gf_bias_act_bn_nhwc": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "gf_conv1_bias_act_bn": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gf_conv3x3_c64": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gf_conv3x3_c64_ld": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "gf_nms_scores": [_P, _P, _I, _I, _I, _I, _I, _P],
    "gf_nms_candidates": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_nms_candidates_cap": [_I, _I, _I],
    "gf_topk_candidates": [_P, _P, _P, _P, _I, _I, _I, _P],
    "gf_detector_scores": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_sample_descriptors": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "gf_rowdot_nblk": [_I],
    "gf_rowdot_fwd": [_P, _P, _F, _P, _P, _I, _I, _I, _P],
    "gf_rowdot_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gf_rowdot2_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gf_rowdot2_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gf_rows_lse_argmax": [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_lg_loss_fwd": [_P] * 9 + [_L] + [_P] * 13 + [_I, _I, _I, _I, _I, _P],
    "gf_lg_loss_bwd_tokens": [_P] * 11 + [_L] + [_P] * 7 + [_I, _I, _I, _P],
    "gf_lg_loss_bwd_rows": [_P] * 5 + [_L] + [_P] * 3 + [_I, _I, _I, _I, _I, _P],
    "gf_rotary_qk": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gf_rotary_qk_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "gf_ln_gelu_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "gf_ln_gelu_nblk": [_I],
    "gf_ln_gelu_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
}
_RESTYPE = {"gf_sinkhorn_ws_bytes": _c.c_int64, "gf_linear_dw_ws_bytes": _c.c_int64}

_lib = None


def build(force=False):
    """Compile csrc/*.hip into libgf_amd.so with hipcc for gfx950 (in-tree)."""
    jobs = str(min(8, os.cpu_count() or 1))
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    res = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libgf_amd.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def load():
    """dlopen libgf_amd.so, declare the prototypes, check the ABI version. Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "glue_factory_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> missing symbol, loud by design
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, _c.c_int)
    if lib.gf_abi_version() != ABI_VERSION:
        raise RuntimeError("libgf_amd.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


_ERRORS = {-1: "unsupported configuration (e.g. head_dim != 64, D not in {64,128,256})",
           -2: "bad shape", -3: "misaligned pointer/stride (rows must stay 16-byte aligned)",
           -4: "unsupported dtype"}


def check(code, what):
    if code != 0:
        msg = _ERRORS.get(code, f"hipError_t {code}")
        raise RuntimeError(f"{what} failed: {msg}")


def strides(*vals):
    return (ctypes.c_int64 * len(vals))(*vals)
