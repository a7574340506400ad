"""Cost-volume builders with the reference's free-function signatures.

``build_gwc_volume(ref, tgt, maxdisp, num_groups)`` is identical in
meta_arch/igev_stereo/submodule.py:160-170, meta_arch/gwcnet/submodules.py:48-58
and meta_arch/cgi/submodule.py.  ``build_concat_volume`` has two upstream
definitions that differ in the reference half; both are here:

    build_concat_volume          GwcNet semantics (gwcnet/submodules.py:25-36)
    build_concat_volume_igev     IGEV copy       (igev_stereo/submodule.py:207-218)

``build_gwc_concat_volume`` writes both into one (B, G+2C', D, H, W) buffer,
replacing the torch.cat of gwc_main.py:315.
"""
import torch

from . import _ffi


def _check_pair(ref, tgt):
    _ffi.require_gpu(ref, tgt)
    _ffi.require_no_grad(ref, tgt)
    if ref.shape != tgt.shape:
        raise ValueError("feature maps disagree: %s vs %s" % (tuple(ref.shape), tuple(tgt.shape)))
    return ref.contiguous(), tgt.contiguous()


#: "mfma": the banded matrix product on the exact-fp32 matrix cores (dkt_gwc_volume_mfma) where it applies, else the VALU
#: kernel; "exact": always the VALU kernel, bit-identical to the C restatement's summation order.  (submodule.gwc_mode(...) switches.)
import os as _os
GWC_MODE = "mfma"


import contextlib as _contextlib


@_contextlib.contextmanager
def gwc_mode(mode):
    """Temporarily select the group-wise correlation kernel: "mfma" (default) or "exact" (VALU, the C oracle's order)."""
    global GWC_MODE
    if mode not in ("mfma", "exact"):
        raise ValueError("gwc_mode: 'mfma' or 'exact'")
    prev, GWC_MODE = GWC_MODE, mode
    try:
        yield
    finally:
        GWC_MODE = prev


def _gwc_into(ref, tgt, vol, maxdisp, num_groups, bstride):
    B, C, H, W = ref.shape
    assert C % num_groups == 0  # groupwise_correlation, submodule.py:154
    if GWC_MODE == "mfma":
        rc = _ffi.lib().dkt_gwc_volume_mfma(ref.data_ptr(), tgt.data_ptr(), vol.data_ptr(), B, C, H, W,
                                            maxdisp, num_groups, bstride, _ffi.device_of(ref), _ffi.stream_of(ref))
        if rc == 0:
            return
        if rc != -7:                              # DKT_E_UNSUPPORTED: shape outside the MFMA form -> the general kernel
            _ffi.check(rc, "dkt_gwc_volume_mfma")
    rc = _ffi.lib().dkt_gwc_volume(ref.data_ptr(), tgt.data_ptr(), vol.data_ptr(), B, C, H, W,
                                   maxdisp, num_groups, bstride, _ffi.device_of(ref), _ffi.stream_of(ref))
    _ffi.check(rc, "dkt_gwc_volume")


def _concat_into(ref, tgt, vol, maxdisp, ref_masked, bstride):
    B, C, H, W = ref.shape
    rc = _ffi.lib().dkt_concat_volume(ref.data_ptr(), tgt.data_ptr(), vol.data_ptr(), B, C, H, W,
                                      maxdisp, int(ref_masked), bstride,
                                      _ffi.device_of(ref), _ffi.stream_of(ref))
    _ffi.check(rc, "dkt_concat_volume")


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    ref, tgt = _check_pair(refimg_fea, targetimg_fea)
    B, C, H, W = ref.shape
    vol = torch.empty((B, num_groups, maxdisp, H, W), device=ref.device, dtype=torch.float32)
    _gwc_into(ref, tgt, vol, maxdisp, num_groups, num_groups * maxdisp * H * W)
    return vol


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    ref, tgt = _check_pair(refimg_fea, targetimg_fea)
    B, C, H, W = ref.shape
    vol = torch.empty((B, 2 * C, maxdisp, H, W), device=ref.device, dtype=torch.float32)
    _concat_into(ref, tgt, vol, maxdisp, True, 2 * C * maxdisp * H * W)
    return vol


def build_concat_volume_igev(refimg_fea, targetimg_fea, maxdisp):
    ref, tgt = _check_pair(refimg_fea, targetimg_fea)
    B, C, H, W = ref.shape
    vol = torch.empty((B, 2 * C, maxdisp, H, W), device=ref.device, dtype=torch.float32)
    _concat_into(ref, tgt, vol, maxdisp, False, 2 * C * maxdisp * H * W)
    return vol


def build_gwc_concat_volume(gwc_ref, gwc_tgt, cat_ref, cat_tgt, maxdisp, num_groups):
    """GWCNet.forward with use_concat_volume (gwc_main.py:310-315) in one buffer:
    channels [0:G] group-wise correlation, [G:G+2C'] concat volume."""
    gref, gtgt = _check_pair(gwc_ref, gwc_tgt)
    cref, ctgt = _check_pair(cat_ref, cat_tgt)
    B, _, H, W = gref.shape
    Cc = cref.shape[1]
    ch = num_groups + 2 * Cc
    vol = torch.empty((B, ch, maxdisp, H, W), device=gref.device, dtype=torch.float32)
    bstride = ch * maxdisp * H * W
    _gwc_into(gref, gtgt, vol, maxdisp, num_groups, bstride)
    _concat_into(cref, ctgt, vol[:, num_groups:], maxdisp, True, bstride)
    return vol


def _group_l2norm(x, num_groups):
    """x / (||x||_2 per channel group + 1e-05): meta_arch/cgi/submodule.py:149,168."""
    _ffi.require_gpu(x)
    _ffi.require_no_grad(x)
    x = x.float().contiguous()
    B, C, H, W = x.shape
    if C % num_groups != 0:
        raise AssertionError("C %% num_groups != 0")          # the reference asserts (submodule.py:145)
    y = torch.empty_like(x)
    rc = _ffi.lib().dkt_group_l2norm(x.data_ptr(), y.data_ptr(), B, C, H * W, num_groups, 1e-05,
                                     _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_group_l2norm")
    return y


def build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """meta_arch/cgi/submodule.py:154-164: group-wise correlation of features normalised per
    channel group.  The norm reduces over channels only, so normalising the two maps once and
    running the plain group-wise volume is the same computation as the reference's per-
    disparity slices."""
    return build_gwc_volume(_group_l2norm(refimg_fea, num_groups), _group_l2norm(targetimg_fea, num_groups),
                            maxdisp, num_groups)


def build_norm_correlation_volume(refimg_fea, targetimg_fea, maxdisp):
    """meta_arch/cgi/submodule.py:171-180 (== igev_stereo/submodule.py:179): (B,1,D,H,W)."""
    return build_gwc_volume_norm(refimg_fea, targetimg_fea, maxdisp, 1)


def context_upsample(disp_low, up_weights):
    """meta_arch/igev_stereo/submodule.py:242-254: (B,1,h,w), (B,9,4h,4w) -> (B,4h,4w)."""
    _ffi.require_gpu(disp_low, up_weights)
    _ffi.require_no_grad(disp_low, up_weights)
    b, c, h, w = disp_low.shape
    if c != 1 or tuple(up_weights.shape) != (b, 9, 4 * h, 4 * w):
        raise ValueError("context_upsample: disp_low %s / up_weights %s" % (tuple(disp_low.shape), tuple(up_weights.shape)))
    disp_low = disp_low.float().contiguous()
    up_weights = up_weights.float().contiguous()
    out = torch.empty((b, 4 * h, 4 * w), device=disp_low.device, dtype=torch.float32)
    rc = _ffi.lib().dkt_context_upsample(disp_low.data_ptr(), up_weights.data_ptr(), out.data_ptr(), b, h, w,
                                         _ffi.device_of(out), _ffi.stream_of(out))
    _ffi.check(rc, "dkt_context_upsample")
    return out


def disparity_regression(x, maxdisp):
    """igev_stereo/submodule.py:220-224 (keepdim=True flavour)."""
    assert len(x.shape) == 4
    disp_values = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(x * disp_values, 1, keepdim=True)
