"""ctypes binding of libnwhead_hip.so (the C ABI in include/nwhead_hip.h).

There is NO fallback: if the shared library is missing, or a tensor is not on a HIP device, the
ops raise.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C nwhead_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NW_HIP_LIB: another build of the same library (A/B timing of kernel variants on one device)
LIB_PATH = os.environ.get("NW_HIP_LIB") or os.path.join(_HERE, "csrc", "libnwhead_hip.so")

SCORE_KINDS = {"euclidean": 0, "hypersphere_euclidean": 1, "cosine": 2, "dotproduct": 3, "clip": 4}

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/nwhead_hip.h one to one
SIGNATURES = {
    "nw_abi_version": (_int, []),
    "nw_status_string": (C.c_char_p, [_int]),
    "nw_device_check": (_int, []),
    "nw_scores_f32": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p, _int, _p]),
    "nw_fwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "nw_row_norm2_f32": (_int, [_p, _p, _i64, _i64, _p]),
    "nw_split_rows_f16x2": (_int, [_p, _p, _p, _p, _i64, _i64, _p]),
    "nw_fwd_f32": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _i64, _i64, _i64, _i64, _int, _p, _int, _int, _p, _p]),
    "nw_fwd_partial_f32": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _i64, _i64, _i64, _i64, _int, _p, _p, _p]),
    "nw_merge_finalize_f32": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "nw_bank_tables_bytes": (_sz, [_i64]),
    "nw_bank_tables_build": (_int, [_p, _i64, _i64, _p, _sz, _p]),
    "nw_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _int, _int]),
    "nw_bwd_f32": (_int, [_p] * 10 + [_p, _sz, _i64, _i64, _i64, _i64, _int, _p, _int, _int, _p]),
    "nw_bwd_uses_split": (_int, [_i64, _i64, _i64, _i64, _int]),
    "nw_bwd_bank_f32": (_int, [_p] * 13 + [_p, _sz, _i64, _i64, _i64, _i64, _int, _p, _int, _int, _p]),
    "nw_support_influence_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "nw_aggregate_f32": (_int, [_p] * 5 + [_i64, _i64, _i64, _int, _p]),
    "nw_aggregate_bwd_f32": (_int, [_p] * 6 + [_i64, _i64, _i64, _int, _p]),
    "nw_fwd_influence_f32": (_int, [_p] * 11 + [_sz, _i64, _i64, _i64, _i64, _int, _p, _p, _p]),
    "nw_topk_f32": (_int, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "nw_scale_shift_relu_f32": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "nw_bias_act_nhwc_f32": (_int, [_p, _p, _p, _int, _p, _i64, _i64, _p]),
    "nw_scale_shift_relu_avgpool2_f32": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "nw_conv3x3_workgroups": (_i64, [_i64, _i64, _i64, _i64, _i64]),
    "nw_conv3x3_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "nw_conv3x3_f32": (_int, [_p, _i64, _p, _p, _p, _i64, _int, _p, _i64, _p, _sz, _i64, _i64, _i64, _i64, _i64, _p]),
    "nw_conv1x1_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "nw_conv1x1_f32": (_int, [_p, _i64, _p, _p, _int, _p, _p, _int, _p, _i64, _p, _sz, _i64, _i64, _i64, _i64, _p]),
    "nw_bn_relu_train_fwd_f32": (_int, [_p] * 10 + [_i64, _i64, _i64, _i64, C.c_float, C.c_float, _int, _p]),
    "nw_bn_relu_train_bwd_f32": (_int, [_p] * 12 + [_i64, _i64, _i64, _i64, _i64, _int, _p]),
    "nw_absmax_f32": (_int, [_p, _i64, _p, _p]),
    "nw_to_nhwc_pad_f32": (_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "nw_conv2d_nhwc_supported": (_int, [_i64] * 9),
    "nw_conv2d_nhwc_f16x2": (_int, [_p, _p, _p, _p, _p, _p, _int, _p, _p] + [_i64] * 11 + [_p, _p]),
    "nw_conv2d_nhwc_moments_groups": (_i64, [_i64] * 9),
    "nw_conv2d_nhwc_bnstat_f16x2": (_int, [_p, _p, _p, _p, _p, _p] + [_i64] * 11 + [_p, _p]),
    "nw_bn_relu_nhwc_train_bwd_from_partials_f32": (_int, [_p, _i64] + [_p] * 6 + [_i64, _p, _p, _p, _p, _i64, _i64, _p, _p, _sz, _i64,
                                                           _i64, _p]),
    "nw_stem7x7s2_relu_maxpool_supported": (_int, [_i64] * 4),
    "nw_stem7x7s2_relu_maxpool_f16x2": (_int, [_p] * 7 + [_i64] * 4 + [_p]),
    "nw_bn_relu_avgpool2x2_nhwc_f32": (_int, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "nw_add_relu_f32": (_int, [_p, _p, _p, _p, _i64, _p]),
    "nw_relu_bwd_f32": (_int, [_p, _p, _p, _p, _i64, _p]),
    "nw_bn_dgrad1x1_workspace_bytes": (_sz, [_i64, _i64]),
    "nw_bn_dgrad1x1_bwd_f16x2": (_int, [_p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _sz, _i64, _i64, _i64, _p]),
    "nw_bn_nhwc_moments_f32": (_int, [_p, _i64, _i64, _i64, C.c_float, _p, _p, _p, _p, _sz, _p]),
    "nw_bn_nhwc_moments_from_partials_f32": (_int, [_p, _i64, _i64, C.c_float, _p, _p, _p, _p]),
    "nw_bn_nhwc_moments_minmax_f32": (_int, [_p, _i64, _i64, _i64, C.c_float, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "nw_bn_nhwc_minmax_workspace_bytes": (_sz, [_i64, _i64]),
    "nw_bn_nhwc_prep_from_partials_f32": (_int, [_p, _i64, _i64, C.c_float, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_float, _int,
                                                 _p, _p, _p]),
    "nw_bn_nhwc_prep_window_from_partials_f32": (_int, [_p, _i64, _i64, _i64, _i64, _i64, C.c_float, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                                        _p, C.c_float, _int, _p, _p, _p]),
    "nw_bn_nhwc_prep_f32": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_float, _int, _i64, _i64, _p, _p, _p]),
    "nw_conv2d_nhwc_bnrelu_f16x2": (_int, [_p] * 3 + [_i64] + [_p] * 3 + [_int, _p, _p] + [_i64] * 11 + [_p, _p]),
    "nw_bn_relu_nhwc_apply_f32": (_int, [_p, _i64] + [_p] * 8 + [C.c_float, _p, _p, _i64, _i64, _int, _p]),
    "nw_split_conv_weights_f16x2": (_int, [_p, _i64, _i64, _p, _p, _p]),
    "nw_conv2d_nhwc_wgrad_supported": (_int, [_i64] * 9),
    "nw_conv2d_nhwc_wgrad_workspace_bytes": (_sz, [_i64] * 9),
    "nw_conv2d_nhwc_wgrad_f16x2": (_int, [_p, _p, _p, _p, _p, _p, _sz] + [_i64] * 11 + [_p]),
    "nw_conv_zero_page": (_p, []),
    "nw_conv2d_nhwc_wgrad_batch_workspace_bytes": (_sz, [_p, _i64]),
    "nw_conv2d_nhwc_wgrad_batch_f16x2": (_int, [_p, _i64, _p, _sz, _p]),
    "nw_bn_nhwc_workspace_bytes": (_sz, [_i64, _i64]),
    "nw_bn_relu_nhwc_train_fwd_f32": (_int, [_p, _i64] + [_p] * 10 + [_sz, _i64, _i64, C.c_float, C.c_float, _int, _p]),
    "nw_bn_relu_nhwc_train_bwd_f32": (_int, [_p, _i64] + [_p] * 9 + [_i64, _i64, _p, _p, _sz, _i64, _i64, _int, _p]),
    "nw_avgpool2x2_nhwc_f32": (_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "nw_avgpool2x2_nhwc_bwd_f32": (_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "nw_maxpool3x3s2_nhwc_f32": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "nw_maxpool3x3s2_nhwc_bwd_f32": (_int, [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "nw_sgd_step_f32": (_int, [_p, _i64, C.c_float, C.c_float, C.c_float, _int, _int, _p]),
    "nw_debug_set": (_int, [C.c_char_p, _int]),
    "nw_debug_tile_timing": (_int, [_int]),
    "nw_debug_tile_timing_read": (_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


class NWHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NWHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run __graft_entry__.build() "
            "(or `make -C nwhead_amd/csrc`). There is no CPU fallback.")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be in the process before this
    # library's NEEDED libamdhip64 is resolved, or the loader pulls /opt/rocm's copy in and the process ends
    # up with two runtimes, the second of which cannot open the device (nw_device_check: no gfx950 device).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.nw_abi_version() != 2:
        raise NWHipError("libnwhead_hip.so ABI version mismatch")
    _lib = lib
    sync_knobs()
    return lib


class WgradJob(C.Structure):
    """nw_wgrad_job (include/nwhead_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("amax_x", C.c_void_p), ("gy", C.c_void_p), ("amax_g", C.c_void_p), ("dw", C.c_void_p)] + \
               [(k, C.c_int64) for k in ("n", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "ldx", "ldg", "out_oihw")] + \
               [("pre_x", C.c_void_p)] + [(k, C.c_int64) for k in ("rowrun_stride", "in_H", "in_W", "row0", "col0")]


class SgdParam(C.Structure):
    """nw_sgd_param (include/nwhead_hip.h)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum_buf", C.c_void_p), ("n", C.c_int64)]


class ConvBnStat(C.Structure):
    """nw_conv_bnstat (include/nwhead_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("partials", C.c_void_p)]


class FwdOpts(C.Structure):
    """nw_fwd_opts (include/nwhead_hip.h): the options of ONE forward call."""
    _fields_ = [("struct_size", C.c_uint32), ("persistent_wgs", C.c_int32), ("force_split", C.c_int32),
                ("reserved", C.c_int32), ("tables", C.c_void_p), ("tables_bytes", C.c_size_t),
                ("tables_sy", C.c_void_p), ("tables_N", C.c_int64)]


# Environment switches of the Python layer (the C library never reads the environment).  NW_SPLIT_ALWAYS travels in every
# forward call's nw_fwd_opts; the others are diagnostic knobs, forwarded once at load time (and again by sync_knobs()).
KNOBS = ("pvar", "qg", "tile_rs", "merge_mq", "merge_per_query", "merge_no_global_tables", "persistent_any_rs", "no_persistent",
         "split_queries", "bwd_no_mfma", "bwd_split", "coeff_threads", "xgemm_wgs", "xgemm_nbuf", "split_lbits", "conv_gather",
         "conv_max_wgs", "wgrad_min_stages", "conv_skip_cfgs", "wgrad_batch_wgs", "bn_inline_fin", "conv_moments_per_tile",
         "conv_force_cfg")
_KNOB_UNSET = -2 ** 31
_knob_state = {}


def sync_knobs():
    """Forward the NW_<KNOB> environment variables to the library's diagnostic knobs (nw_debug_set)."""
    if _lib is None:
        return
    for k in KNOBS:
        v = os.environ.get("NW_" + k.upper())
        val = _KNOB_UNSET if v is None or v == "" else int(v)
        if _knob_state.get(k, _KNOB_UNSET) != val:
            _lib.nw_debug_set(k.encode(), val)
            _knob_state[k] = val


def force_split():
    return os.environ.get("NW_SPLIT_ALWAYS", "") == "1"


def fwd_opts(tables=None, tables_bytes=0, persistent_wgs=0, tables_sy=None, tables_n=-1):
    """An nw_fwd_opts for one forward call (a ctypes object: keep it alive across the call)."""
    return FwdOpts(C.sizeof(FwdOpts), int(persistent_wgs), int(force_split()), 0, tables, int(tables_bytes), tables_sy,
                   int(tables_n))


def check(status: int, what: str):
    if status != 0:
        msg = load().nw_status_string(status).decode()
        raise NWHipError(f"{what}: {msg} (status {status})")
