"""ctypes binding of libmmgnn.so (C ABI: include/mmgnn.h).

The library is REQUIRED: there is no CPU or eager-PyTorch fallback anywhere in this
package.  If it is missing, importing the ops raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmgnn.so")

MMG_MAX_REL = 4


class MmgError(RuntimeError):
    pass


class RelT(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("rowscale", C.c_void_p),
                ("colscale", C.c_void_p), ("table", C.c_void_p), ("out", C.c_void_p),
                ("n_cols", C.c_int32), ("flags", C.c_uint32), ("mask_t", C.c_void_p),
                ("mask_r", C.c_void_p)]


class PrologueT(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int), ("drop_p", C.c_float),
                ("seed", C.c_uint64), ("site", C.c_uint32), ("row_offset", C.c_int64), ("seed_ptr", C.c_void_p)]


class HeadT(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("W3", C.c_void_p), ("b3", C.c_void_p)]


class HeadGradT(C.Structure):
    _fields_ = [("dA", C.c_void_p), ("dB", C.c_void_p), ("dW2", C.c_void_p), ("db2", C.c_void_p),
                ("dW3", C.c_void_p), ("db3", C.c_void_p)]


class SmallFwdT(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("X2", C.c_void_p), ("W2", C.c_void_p), ("bias", C.c_void_p),
                ("Y", C.c_void_p), ("M", C.c_int64), ("flags", C.c_int)]


class SmallWgradT(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p), ("M", C.c_int64),
                ("accumulate", C.c_int)]


class SmallBnT(C.Structure):
    _fields_ = [("Y", C.c_void_p), ("out", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("stats_out", C.c_void_p), ("M", C.c_int64),
                ("training", C.c_int), ("act", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64), ("site", C.c_uint32),
                ("row_offset", C.c_int64), ("seed_ptr", C.c_void_p)]


class SmallBnBwdT(C.Structure):
    _fields_ = [("G", C.c_void_p), ("Y", C.c_void_p), ("dY", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dbeta", C.c_void_p), ("dgamma", C.c_void_p), ("M", C.c_int64),
                ("training", C.c_int), ("act", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64), ("site", C.c_uint32),
                ("row_offset", C.c_int64), ("seed_ptr", C.c_void_p)]


class BnFinT(C.Structure):
    _fields_ = [("count", C.c_int64), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("n_updates", C.c_int), ("momentum", C.c_float), ("eps", C.c_float),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p)]


class NextBnT(C.Structure):
    _fields_ = [("y", C.c_void_p), ("pro", C.POINTER(PrologueT)), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("sums", C.c_void_p), ("accumulate", C.c_int), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


class PairSavedT(C.Structure):
    _fields_ = [("h1_bits", C.c_void_p), ("h2", C.c_void_p), ("by_position", C.c_int), ("n_entries", C.c_int64)]


class WgradReduceT(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("n4", C.c_int64), ("n_split", C.c_int), ("dW", C.c_void_p), ("dbias", C.c_void_p),
                ("nk4", C.c_int64), ("accumulate", C.c_int)]


class SumJobT(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p * 4), ("n_src", C.c_int), ("len", C.c_int),
                ("cols", C.c_int), ("ld_dst", C.c_int), ("ld_src", C.c_int * 4)]


_vp, _i64, _i32, _f32, _sz, _u64, _u32 = (C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t,
                                          C.c_uint64, C.c_uint32)
_P = C.POINTER

# name -> (restype, argtypes): every symbol include/mmgnn.h declares
SIGNATURES = {
    "mmg_version": (C.c_int, []),
    "mmg_last_error": (C.c_char_p, []),
    "mmg_stream_create": (C.c_int, [_P(_vp)]),
    "mmg_stream_destroy": (C.c_int, [_vp]),
    "mmg_csr_build_ws_bytes": (_sz, [_i64, _i64]),
    "mmg_csr_build": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mmg_row_degree": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "mmg_col_degree": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "mmg_rel_mask_words": (_sz, [_i64, C.c_int32]),
    "mmg_rel_mask_build": (C.c_int, [_vp, _vp, _i64, C.c_int32, _vp, _vp, _vp]),
    "mmg_gather_rows": (C.c_int, [_P(RelT), _i32, _i64, _i32, _vp, _i32, _vp]),
    "mmg_gather_rows_stats_ws_bytes": (_sz, [_i64, _i32]),
    "mmg_gather_rows_stats": (C.c_int, [_P(RelT), _i32, _i64, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "mmg_gather_rows_stats_bn": (C.c_int, [_P(RelT), _i32, _i64, _i32, _vp, _i32, _vp, _vp, _sz, _P(BnFinT), _vp]),
    "mmg_scatter_rows_ws_bytes": (_sz, [_P(RelT), _i32, _i64, _i32]),
    "mmg_scatter_rows": (C.c_int, [_P(RelT), _i32, _i64, _i32, _vp, _vp, _sz, _vp]),
    "mmg_linear_fwd": (C.c_int, [_vp, _P(PrologueT), _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "mmg_linear_fwd_stats_ws_bytes": (_sz, [_i64, _i32]),
    "mmg_linear_fwd_stats": (C.c_int, [_vp, _P(PrologueT), _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "mmg_linear_fwd_stats_bn": (C.c_int, [_vp, _P(PrologueT), _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _P(BnFinT), _vp]),
    "mmg_linear_wgrad_ws_bytes": (_sz, [_i64, _i32, _i32]),
    "mmg_linear_wgrad": (C.c_int, [_vp, _vp, _P(PrologueT), _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "mmg_linear_wgrad_deferred": (C.c_int, [_vp, _vp, _P(PrologueT), _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp,
                                            _P(WgradReduceT)]),
    "mmg_wgrad_reduce_group": (C.c_int, [_P(WgradReduceT), _i32, _vp]),
    "mmg_linear_wgrad_is_direct": (C.c_int, [_i64, _i32, _i32]),
    "mmg_col_reduce2_ws_bytes": (_sz, [_i64, _i32]),
    "mmg_col_reduce2": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _sz, _vp]),
    "mmg_bn_finalize": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp,
                                  _i32, _vp]),
    "mmg_affine_act_drop": (C.c_int, [_vp, _P(PrologueT), _vp, _i64, _i32, _vp]),
    "mmg_affine_act_drop_rows": (C.c_int, [_vp, _P(PrologueT), _vp, _i64, _vp, _i32, _vp]),
    "mmg_probe_arm": (C.c_int, [_i32]),
    "mmg_probe_read": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32]),
    "mmg_bn_bwd_stats2": (C.c_int, [_vp, _vp, _vp, _P(PrologueT), _P(PrologueT), _vp, _vp, _vp, _i64, _i32, _vp, C.c_size_t, _vp]),
    "mmg_bn_bwd_apply2": (C.c_int, [_vp, _vp, _vp, _P(PrologueT), _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _i64,
                                    _i32, _vp]),
    "mmg_bn_bwd_stats_rows_ws_bytes": (C.c_size_t, [_i32]),
    "mmg_bn_bwd_stats_rows": (C.c_int, [_vp, _vp, _vp, _i64, _P(PrologueT), _vp, _vp, _vp, _i32, _vp, C.c_size_t, _vp]),
    "mmg_bn_bwd_apply_rows": (C.c_int, [_vp, _vp, _vp, _i64, _P(PrologueT), _vp, _i32, _vp]),
    "mmg_bn_bwd_stats": (C.c_int, [_vp, _vp, _P(PrologueT), _vp, _vp, _vp, _i64, _i32, _vp, _sz, _vp]),
    "mmg_bn_bwd_apply": (C.c_int, [_vp, _vp, _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "mmg_linear_fwd_l2norm_supported": (C.c_int, [_i64, _i32, _i32]),
    "mmg_linear_fwd_l2norm": (C.c_int, [_vp, _P(PrologueT), _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "mmg_linear_bnbwd_supported": (C.c_int, [_i64, _i32, _i32]),
    "mmg_linear_l2bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "mmg_next_bn_ws_bytes": (_sz, [_i64, _i32]),
    "mmg_linear_fwd_next_bn": (C.c_int, [_vp, _P(PrologueT), _vp, _vp, _vp, _i64, _i32, _i32, _i32, _P(NextBnT), _vp]),
    "mmg_linear_l2bwd_next_bn": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _P(NextBnT), _vp]),
    "mmg_linear_bnbwd_next_bn": (C.c_int, [_vp, _vp, _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp,
                                           _i64, _i32, _i32, _P(NextBnT), _vp]),
    "mmg_linear_bnbwd_rows_next_bn": (C.c_int, [_vp, _vp, _i64, _vp, _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp,
                                                _vp, _vp, _i64, _i32, _i32, _P(NextBnT), _vp]),
    "mmg_gather_rows_next_bn": (C.c_int, [_P(RelT), _i32, _i64, _i32, _vp, _i32, _P(NextBnT), _vp]),
    "mmg_linear_bnbwd_rows": (C.c_int, [_vp, _vp, _i64, _vp, _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp,
                                        _i64, _i32, _i32, _vp]),
    "mmg_linear_bnbwd2": (C.c_int, [_vp, _vp, _vp, _P(PrologueT), _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp,
                                    _vp, _vp, _i64, _i32, _i32, _vp]),
    "mmg_linear_bnbwd": (C.c_int, [_vp, _vp, _P(PrologueT), _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _i64, _i32,
                                   _i32, _vp]),
    "mmg_l2norm_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "mmg_l2norm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "mmg_pair_loss_ws_bytes": (_sz, [_i64]),
    "mmg_pair_loss": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_double, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "mmg_sup_mask_ws_bytes": (_sz, [_i64]),
    "mmg_sup_mask_draw": (C.c_int, [_vp, _u64, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mmg_dropout_mask": (C.c_int, [_u64, _vp, _u32, _i64, _i64, _f32, _vp, _vp]),
    "mmg_pair_head_fwd": (C.c_int, [_P(HeadT), _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _f32, _u64, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp]),
    "mmg_pair_head_fwd_save": (C.c_int, [_P(HeadT), _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _f32, _u64, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _P(PairSavedT), _vp]),
    "mmg_pair_head_bwd_saved": (C.c_int, [_P(HeadT), _P(HeadGradT), _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _f32,
                                          _u64, _vp, _vp, _vp, _vp, _vp, _vp, _P(PairSavedT), _vp, _sz, _vp]),
    "mmg_pair_head_bwd_ws_bytes": (_sz, [_i64, _i32]),
    "mmg_pair_head_bwd": (C.c_int, [_P(HeadT), _P(HeadGradT), _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _f32,
                                    _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mmg_small_fwd_group": (C.c_int, [_P(SmallFwdT), _i32, _i32, _i32, _vp]),
    "mmg_small_wgrad_group": (C.c_int, [_P(SmallWgradT), _i32, _i32, _i32, _vp]),
    "mmg_small_bn_act_group": (C.c_int, [_P(SmallBnT), _i32, _i32, _f32, _f32, _vp]),
    "mmg_small_bn_bwd_group": (C.c_int, [_P(SmallBnBwdT), _i32, _i32, _vp]),
    "mmg_adam_step": (C.c_int, [_vp, _vp, _vp, _P(C.c_void_p), _P(C.c_int32), _i32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp]),
    "mmg_adam_step_dev": (C.c_int, [_vp, _vp, _vp, _P(C.c_void_p), _P(C.c_int32), _i32, _vp, _vp, _vp, _vp]),
    "mmg_vec_sums": (C.c_int, [_P(SumJobT), _i32, _vp]),
    "mmg_counters_add": (C.c_int, [_P(C.c_void_p), _P(C.c_int64), _i32, _vp]),
    "mmg_seed_advance": (C.c_int, [_vp, _vp]),
    "mmg_fill_zero": (C.c_int, [_vp, _sz, _vp]),
    "mmg_seg_reduce_ws_bytes": (_sz, [_i64, _i32]),
    "mmg_seg_moments": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "mmg_seg_metrics": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _f32, _vp, _vp, _vp, _sz, _vp]),
    "mmg_pair_select_ws_bytes": (_sz, [_i64]),
    "mmg_pair_select": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def load():
    """Load libmmgnn.so (once).  Raises MmgError loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmgError(
            f"{LIB_PATH} not found: the HIP library is mandatory (no CPU fallback). Build it with "
            f"`make -C {os.path.join(_HERE, 'csrc')}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch (the package's device-memory plumbing) bundles its own HIP runtime.  It has to be in the process BEFORE this
    # library is mapped: libmmgnn.so then binds to it; mapped first, the library pulls in /opt/rocm's runtime, PyTorch adds
    # its own later, and the second runtime of a process sees no device ("no ROCm-capable device is detected" from the
    # first library call -- `python __graft_entry__.py smoke`, where build() loaded the library before anything imported torch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mmg_last_error()
        raise MmgError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
