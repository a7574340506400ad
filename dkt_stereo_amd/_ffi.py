"""ctypes binding of libdktstereo.so (include/dktstereo.h).

This is the whole Python<->native boundary of the package: device pointers come
from ``tensor.data_ptr()``, the stream from ``torch.cuda.current_stream()``; no
torch type crosses the ABI.  There is no CPU fallback -- if the library is
missing or a call fails this module raises.
"""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DKT_LIB_PATH") or os.path.join(_HERE, "lib", "libdktstereo.so")

_c_f32p = ctypes.c_void_p
_i, _l, _f, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)
_ip = ctypes.POINTER(ctypes.c_int)
_lp = ctypes.POINTER(ctypes.c_long)

class ConvDesc(ctypes.Structure):
    """dkt_conv_desc of include/dktstereo.h."""
    _fields_ = [("src", ctypes.c_void_p * 4), ("src_bstride", ctypes.c_long * 4), ("src_channels", ctypes.c_int * 4),
                ("nsrc", ctypes.c_int), ("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("out_scale", ctypes.c_float), ("in_scale", ctypes.c_float), ("out", ctypes.c_void_p),
                ("out_bstride", ctypes.c_long), ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("Cout", ctypes.c_int), ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("relu", ctypes.c_int),
                ("epilogue", ctypes.c_int), ("e0", ctypes.c_void_p), ("e0_bstride", ctypes.c_long),
                ("e1", ctypes.c_void_p), ("e1_bstride", ctypes.c_long), ("h", ctypes.c_void_p),
                ("h_bstride", ctypes.c_long), ("out2", ctypes.c_void_p), ("out2_bstride", ctypes.c_long),
                ("in_norm", ctypes.c_void_p), ("stride", ctypes.c_int), ("stats_ws", ctypes.c_void_p),
                ("stats_part", ctypes.c_void_p)]


class ConvC8Desc(ctypes.Structure):
    """dkt_conv_c8_desc of include/dktstereo.h."""
    _fields_ = [("src", ctypes.c_void_p * 4), ("src_bstride", ctypes.c_long * 4), ("src_channels", ctypes.c_int * 4),
                ("nsrc", ctypes.c_int), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("out_scale", ctypes.c_float), ("act_scale", ctypes.c_float),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cout", ctypes.c_int), ("relu", ctypes.c_int),
                ("epilogue", ctypes.c_int), ("out", ctypes.c_void_p), ("out_bstride", ctypes.c_long),
                ("out_c8", ctypes.c_void_p), ("out_c8_bstride", ctypes.c_long), ("out_c8_ch0", ctypes.c_int),
                ("e0", ctypes.c_void_p), ("e0_bstride", ctypes.c_long), ("e1", ctypes.c_void_p), ("e1_bstride", ctypes.c_long),
                ("h", ctypes.c_void_p), ("h_bstride", ctypes.c_long), ("out2", ctypes.c_void_p), ("out2_bstride", ctypes.c_long),
                ("out2_c8", ctypes.c_void_p), ("out2_c8_bstride", ctypes.c_long), ("out2_c8_ch0", ctypes.c_int),
                ("tail", ctypes.c_void_p), ("tail_bstride", ctypes.c_long), ("tail_channels", ctypes.c_int),
                ("head_w", ctypes.c_void_p), ("head_out", ctypes.c_void_p), ("head_out_bstride", ctypes.c_long),
                ("head_outputs", ctypes.c_int), ("f32_c4", ctypes.c_int), ("tail_scale", ctypes.c_float), ("passes", ctypes.c_int)]


class GruC8Desc(ctypes.Structure):
    """dkt_gru_c8_desc of include/dktstereo.h."""
    _fields_ = [("h_c8", ctypes.c_void_p), ("h_c8_bstride", ctypes.c_long),
                ("x", ctypes.c_void_p * 3), ("x_bstride", ctypes.c_long * 3), ("x_channels", ctypes.c_int * 3), ("nx", ctypes.c_int),
                ("rh_c8", ctypes.c_void_p), ("rh_c8_bstride", ctypes.c_long),
                ("w_zr", ctypes.c_void_p), ("w_q", ctypes.c_void_p),
                ("bz", ctypes.c_void_p), ("br", ctypes.c_void_p), ("bq", ctypes.c_void_p),
                ("cz", ctypes.c_void_p), ("cr", ctypes.c_void_p), ("cq", ctypes.c_void_p),
                ("cz_bstride", ctypes.c_long), ("cr_bstride", ctypes.c_long), ("cq_bstride", ctypes.c_long),
                ("h", ctypes.c_void_p), ("h_bstride", ctypes.c_long),
                ("scale_zr", ctypes.c_float), ("scale_q", ctypes.c_float), ("act_scale", ctypes.c_float),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("hidden", ctypes.c_int),
                ("flags", ctypes.c_void_p), ("passes", ctypes.c_int)]


class MotionFrontDesc(ctypes.Structure):
    """dkt_motion_front_desc of include/dktstereo.h."""
    _fields_ = [("skew", ctypes.POINTER(ctypes.c_void_p)),
                ("planes", ctypes.c_void_p), ("planes_bstride", ctypes.c_long), ("n_co", ctypes.c_int),
                ("head_bias", ctypes.c_void_p),
                ("x_old", ctypes.c_void_p), ("x_old_bstride", ctypes.c_long),
                ("x_new", ctypes.c_void_p), ("x_new_bstride", ctypes.c_long),
                ("x0", ctypes.c_void_p), ("x0_bstride", ctypes.c_long),
                ("flow", ctypes.c_void_p), ("flow_bstride", ctypes.c_long),
                ("w_cor", ctypes.c_void_p), ("b_cor", ctypes.c_void_p), ("cor_channels", ctypes.c_int),
                ("cor_c8", ctypes.c_void_p), ("cor_c8_bstride_bytes", ctypes.c_long), ("cor_c8_ch0", ctypes.c_int),
                ("cor_act_scale", ctypes.c_float),
                ("stem_w_hi", ctypes.c_void_p), ("stem_w_lo", ctypes.c_void_p), ("stem_bias", ctypes.c_void_p),
                ("stem_out_scale", ctypes.c_float), ("stem_in_scale", ctypes.c_float), ("stem_cin", ctypes.c_int),
                ("stem_cout", ctypes.c_int),
                ("flo_c8", ctypes.c_void_p), ("flo_c8_bstride_bytes", ctypes.c_long), ("flo_c8_ch0", ctypes.c_int),
                ("flo_act_scale", ctypes.c_float),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("W1", ctypes.c_int), ("W2", ctypes.c_int), ("L", ctypes.c_int),
                ("r", ctypes.c_int)]


class ResampleC8Job(ctypes.Structure):
    """dkt_resample_c8_job of include/dktstereo.h."""
    _fields_ = [("x", ctypes.c_void_p), ("x_bstride", ctypes.c_long), ("dst", ctypes.c_void_p), ("dst_bstride_bytes", ctypes.c_long),
                ("B", ctypes.c_int), ("C", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Ho", ctypes.c_int),
                ("Wo", ctypes.c_int), ("ch0", ctypes.c_int), ("kind", ctypes.c_int), ("scale", ctypes.c_float)]


class C8RangeJob(ctypes.Structure):
    """dkt_c8_range_job of include/dktstereo.h."""
    _fields_ = [("t", ctypes.c_void_p), ("bstride_bytes", ctypes.c_long), ("B", ctypes.c_int), ("C", ctypes.c_int),
                ("H", ctypes.c_int), ("W", ctypes.c_int), ("tail", ctypes.c_int)]


#: DKT_STATUS_MAX_JOBS of include/dktstereo.h
STATUS_MAX_JOBS = 8

# name -> argtypes, mirrors include/dktstereo.h one to one
SIGNATURES = {
    "dkt_loop_status": [ctypes.POINTER(C8RangeJob), _i, _vp, _l, _vp, _vp, _i, _vp],
    "dkt_motion_front_c8": [ctypes.POINTER(MotionFrontDesc), _i, _vp],
    "dkt_resample_pair_c8": [ctypes.POINTER(ResampleC8Job), ctypes.POINTER(ResampleC8Job), _i, _vp],
    "dkt_gru_c8_flag_words": [_i, _i, _i],
    "dkt_conv2d_stats_ws_floats": [_i, _i, _i, _i],
    "dkt_gru_c8": [ctypes.POINTER(GruC8Desc), _vp, _i, _vp],
    "dkt_gru_c8_pair": [ctypes.POINTER(GruC8Desc), ctypes.POINTER(GruC8Desc), _vp, _i, _vp],
    "dkt_pool2x_c8": [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dkt_interp_c8": [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dkt_conv2d_stem7_c8": [_vp, _l, _vp, _vp, _vp, _f, _f, _vp, _l, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_stem7_dual": [_vp, _l, _vp, _vp, _vp, _f, _f, _vp, _l, _vp, _l, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_instance_norm_join_c8": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _l, _i, _f, _i, _i, _i, _i, _i, _vp],
    "dkt_normalize_pair": [_vp, _l, _vp, _l, _vp, _i, _l, _i, _vp],
    "dkt_geo_lookup_conv1x1": [_pp, _pp, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp, _l, _i, _f, _vp, _l,
                               _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_corr1d_lookup_conv1x1_c8": [_pp, _vp, _l, _vp, _vp, _vp, _l, _i, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_c8_head_blocks": [_i, _i],
    "dkt_head_finish": [_vp, _l, _i, _vp, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp],
    "dkt_act_c8_dims": [_i, _i, _ip, _ip],
    "dkt_act_c8_pack": [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dkt_act_c8_unpack": [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dkt_conv_c8_packed_bytes": [_ip, _i, _i],
    "dkt_conv_c8_pack_weights": [_vp, _ip, _i, _i, _f, _vp, _i, _vp],
    "dkt_conv2d_c8": [ctypes.POINTER(ConvC8Desc), _i, _i, _vp],
    "dkt_conv2d_c8_pair": [ctypes.POINTER(ConvC8Desc), ctypes.POINTER(ConvC8Desc), _i, _i, _vp],
    "dkt_conv2d_c8_chain": [ctypes.POINTER(ConvC8Desc), ctypes.POINTER(ConvC8Desc), _i, ctypes.POINTER(ConvC8Desc), _i, _vp, _vp, _i, _i,
                            _i, _vp],
    "dkt_conv2d_c8_chain_flag_words": [ctypes.POINTER(ConvC8Desc), _i, _i],
    "dkt_conv2d_f16s_pair": [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), _i, _i, _vp],
    "dkt_conv2d_f16s_desc": [ctypes.POINTER(ConvDesc), _i, _i, _vp],
    "dkt_corr1d_build": [_vp, _vp, _pp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dkt_corr1d_lookup": [_pp, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_convex_upsample": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_context_upsample": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "dkt_corr1d_lookup_bwd": [_vp, _vp, _l, _pp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_geo_lookup_bwd": [_vp, _vp, _vp, _pp, _pp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_geo_pool_bwd": [_pp, _vp, _l, _i, _l, _i, _i, _vp],
    "dkt_corr1d_pool_bwd": [_pp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dkt_pool_rows": [_vp, _vp, _l, _i, _i, _i, _vp],
    "dkt_pcv_lookup": [_pp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_group_l2norm": [_vp, _vp, _i, _i, _l, _i, ctypes.c_float, _i, _vp],
    "dkt_corr1d_skew_pitch": [_i],
    "dkt_corr1d_skew": [_pp, _pp, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_corr1d_lookup_skew": [_pp, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_corr1d_lookup_conv1x1": [_pp, _vp, _l, _vp, _vp, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_corr1d_lookup_otf": [_vp, _pp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_pool_w": [_vp, _vp, _l, _i, _i, _vp],
    "dkt_l2norm_channels": [_vp, _vp, _i, _i, _l, _i, _vp],
    "dkt_pool_d": [_vp, _vp, _l, _i, _l, _i, _vp],
    "dkt_geo_lookup": [_pp, _pp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_gwc_volume": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _i, _vp],
    "dkt_gwc_volume_mfma": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _i, _vp],
    "dkt_concat_volume": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _i, _vp],
    "dkt_gru_gate_zr": [_vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _l, _i, _i, _l, _i, _vp],
    "dkt_gru_gate_out": [_vp, _vp, _l, _vp, _vp, _l, _vp, _l, _i, _i, _l, _i, _vp],
    "dkt_conv2d_packed_elems": [_ip, _i, _i, _i, _i],
    "dkt_conv2d_pack_weights": [_vp, _ip, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp],
    "dkt_conv2d_f16s": [_pp, _ip, _lp, _i, _vp, _vp, _vp, _f, _f, _vp, _l,
                        _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_instance_norm_stats": [_vp, _vp, _i, _l, _i, _vp],
    "dkt_instance_norm_add_relu": [_vp, _vp, _vp, _vp, _i, _l, _f, _i, _vp],
    "dkt_instance_norm_finalize": [_vp, _i, _l, _f, _vp, _i, _vp],
    "dkt_instance_norm_add_relu_lazy": [_vp, _vp, _i, _vp, _vp, _vp, _i, _l, _f, _i, _vp],
    "dkt_conv2d_stem7_packed_elems": [_i],
    "dkt_conv2d_stem7_pack": [_vp, _i, _i, _f, _vp, _vp, _i, _vp],
    "dkt_conv2d_stem7": [_vp, _l, _vp, _vp, _vp, _f, _f, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_f16s_strided": [_pp, _ip, _lp, _i, _vp, _vp, _vp, _f, _f, _vp, _l,
                                _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_f16s_gate_zr": [_pp, _ip, _lp, _i, _vp, _vp, _vp, _f, _f, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l,
                                _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_f16s_gate_out": [_pp, _ip, _lp, _i, _vp, _vp, _vp, _f, _f, _vp, _l, _vp, _l, _vp, _l, _vp, _l,
                                 _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_direct": [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_direct_accumulate": [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_conv2d_direct_accumulate_diff": [_vp, _l, _vp, _vp, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "dkt_instance_norm_workspace": [_i, _l],
    "dkt_instance_norm": [_vp, _vp, _vp, _i, _l, _f, _i, _i, _vp],
    "dkt_add_relu": [_vp, _vp, _vp, _l, _i, _vp],
    "dkt_pool2x": [_vp, _vp, _l, _i, _i, _i, _vp],
    "dkt_interp_bilinear": [_vp, _vp, _l, _i, _i, _i, _i, _i, _vp],
}
#: entry points that do not return an int status
RESTYPES = {"dkt_gru_c8_flag_words": ctypes.c_long, "dkt_conv2d_c8_chain_flag_words": ctypes.c_long, "dkt_conv2d_stats_ws_floats": ctypes.c_long, "dkt_conv_c8_packed_bytes": ctypes.c_long, "dkt_conv2d_packed_elems": ctypes.c_long, "dkt_conv2d_stem7_packed_elems": ctypes.c_long, "dkt_instance_norm_workspace": ctypes.c_long}

#: DKT_E_UNSUPPORTED of include/dktstereo.h
E_UNSUPPORTED = -7

_lib = None


class DktError(RuntimeError):
    pass


def lib():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DktError(
                "libdktstereo.so is missing (%s). Build it with "
                "`python -m dkt_stereo_amd.build` (needs hipcc); there is no fallback path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.dkt_version.restype = ctypes.c_int
        L.dkt_version.argtypes = []
        L.dkt_strerror.restype = ctypes.c_char_p
        L.dkt_strerror.argtypes = [ctypes.c_int]
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = RESTYPES.get(name, ctypes.c_int)
            fn.argtypes = argtypes
        _lib = L
    return _lib


_LAUNCH_LOG = threading.local()


class launch_log:
    """Collects the names of the C-ABI launches this thread makes inside the block (every launch is followed by check()):
    how the tests and bench.py count the dispatches of a captured loop unit."""

    def __enter__(self):
        self.prev = getattr(_LAUNCH_LOG, "names", None)
        self.names = _LAUNCH_LOG.names = []
        return self.names

    def __exit__(self, *exc):
        _LAUNCH_LOG.names = self.prev
        if self.prev is not None:
            self.prev.extend(self.names)
        return False


def check(rc, what):
    if rc != 0:
        raise DktError("%s failed: %s (rc=%d)" % (what, lib().dkt_strerror(rc).decode(), rc))
    names = getattr(_LAUNCH_LOG, "names", None)
    if names is not None:
        names.append(what)


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return ctypes.cast(arr, _pp)


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def device_of(t):
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def require_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise DktError("dkt_stereo_amd operators run on a HIP device only (got a %s tensor); "
                           "there is no CPU path in the product" % t.device)
        if t.dtype != torch.float32:
            raise DktError("dkt_stereo_amd operators are float32 (got %s)" % t.dtype)


def require_no_grad(*tensors):
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        raise DktError("the HIP path is inference-only: call under torch.no_grad() "
                       "(autograd through these operators is not implemented)")
