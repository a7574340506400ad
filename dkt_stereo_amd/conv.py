"""Single entry point for every 2-D convolution on the hot path.

``conv2d(x, layer, relu=False)``: ``x`` is a tensor or a LIST of tensors that the
reference would ``torch.cat`` along channels first (core/update.py:24-25,29,83);
``layer`` is anything with ``weight``/``bias``/``padding`` (an ``nn.Conv2d`` or
the merged z|r pair built by ``ConvGRU``).

Backends (``set_backend``; default "f16x3"):
  "f16x3"    dkt_conv2d_f16s, passes=3: fp32 emulated on the fp16 matrix cores
             with split operands (w_hi*x_hi + w_lo*x_hi + w_hi*x_lo, fp32
             accumulate) -- ~22 bits per operand, 1.5e-6 relative to an fp64
             convolution (the vendor fp32 path: 0.9e-6); list inputs are read in
             place.  Meets the 1e-3 final-disparity bound with 25x margin.
  "miopen"   vendor fp32 convolution via torch (concatenates list inputs).
  "f16x2"    passes=2: activations rounded to fp16, weights split.
  "f16"      passes=1: plain fp16 operands, fp32 accumulate.
1x1 / 3x3 layers with padding K/2, stride 1 or 2, groups 1 and dilation 1 run on the kernel; the
7x7 stems on dkt_conv2d_stem7; anything else on the vendor path.

Activation range of the split-fp16 backends.  An activation x is represented as fp16 hi + fp16 lo
of ``x * 2**layer.dkt_in_exp`` (``dkt_in_exp`` = 0 unless set): 22 significant bits for
2^-3 <= |x * 2^e| < 65520, an ABSOLUTE resolution of 2^-25 below that (harmless next to O(1)
activations, a loss of relative precision for tensors that are tiny throughout), and NON-FINITE
results above it -- an out-of-range, Inf or NaN activation is never saturated silently, it shows up
as Inf/NaN in the output (``RAFTStereo.forward`` checks its result).  Layers whose inputs live far
from O(1) get an exponent: by hand (``layer.dkt_in_exp = -4``) or from ``calibrate()``.

Thread safety (the reference drives replicas from one Python thread per GPU, tools/ft_dkt.py:119):
the packed-weight caches are keyed per device and guarded by a lock, the backend can be overridden
per thread (``use_backend``), nothing here mutates shared module state during a forward.
"""
import contextlib
import ctypes
import math
import threading

import torch
import torch.nn.functional as F

from . import _ffi

import os as _os

_PASSES = {"f16x3": 3, "f16x2": 2, "f16": 1}
_FEW_DIRECT = True
_BACKEND = "f16x3"
_TLS = threading.local()
_CACHE_LOCK = threading.RLock()


def _check_backend(name):
    if name != "miopen" and name not in _PASSES:
        raise ValueError("unknown conv backend %r" % (name,))


def set_backend(name):
    """Process-wide default backend (threads may override it with ``use_backend``)."""
    global _BACKEND
    _check_backend(name)
    _BACKEND = name


def get_backend():
    return getattr(_TLS, "backend", None) or _BACKEND


@contextlib.contextmanager
def use_backend(name):
    """Backend override for the calling thread only."""
    _check_backend(name)
    prev = getattr(_TLS, "backend", None)
    _TLS.backend = name
    try:
        yield
    finally:
        _TLS.backend = prev


def in_exp_of(layer):
    return int(getattr(layer, "dkt_in_exp", 0) or 0)


@contextlib.contextmanager
def calibrate(margin_bits=2):
    """Records max|x| of every convolution input inside the ``with`` block (synchronising: run it
    once, on representative inputs, outside any timed region or stream capture) and sets
    ``layer.dkt_in_exp`` so that the largest activation seen lands ``margin_bits`` binades below the
    fp16 limit.  Layers whose inputs are already comfortably inside [2^-3, 2^13] keep exponent 0."""
    rec = {}
    _TLS.calib = rec
    try:
        yield rec
    finally:
        _TLS.calib = None
        for layer, amax in rec.values():
            if not (amax > 0.0) or not math.isfinite(amax):
                continue
            e = (15 - margin_bits) - math.floor(math.log2(amax)) - 1        # amax * 2^e in [2^(14-m), 2^(15-m))
            layer.dkt_in_exp = 0 if (-2 <= e <= 16 - margin_bits) else e


def calibrating():
    """True inside ``with calibrate():`` on this thread (range recording synchronises: no stream capture then)."""
    return getattr(_TLS, "calib", None) is not None


def _vendor(x, layer, relu):
    if isinstance(x, (list, tuple)):
        x = x[0] if len(x) == 1 else torch.cat(list(x), dim=1)
    y = F.conv2d(x, layer.weight, layer.bias, stride=_stride_of(layer), padding=layer.padding,
                 dilation=getattr(layer, "dilation", 1), groups=getattr(layer, "groups", 1))
    return torch.relu_(y) if relu else y


def _record_range(layer, srcs):
    rec = getattr(_TLS, "calib", None)
    if rec is None:
        return
    amax = max(float(s.detach().abs().max()) for s in srcs)
    prev = rec.get(id(layer))
    rec[id(layer)] = (layer, amax if prev is None else max(amax, prev[1]))


class _Packed:
    __slots__ = ("key", "hi", "lo", "inv_scale", "bias")


def _packed_weights(layer, src_channels):
    """Split-fp16 weight image for dkt_conv2d_f16s, cached on the layer PER DEVICE (the shallow module
    copies of nn.parallel.replicate share the cache dict; their parameters live on different devices)
    and rebuilt when the parameter tensor is replaced or written."""
    with _CACHE_LOCK:
        return _packed_weights_locked(layer, src_channels)


def _packed_weights_locked(layer, src_channels):
    w = layer.weight
    b = layer.bias
    key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version), tuple(src_channels))
    cache = layer.__dict__.setdefault("_dkt_packed", {})
    slot = (str(w.device), tuple(src_channels))
    hit = cache.get(slot)
    if hit is not None and hit.key == key:
        return hit
    cout, cin, kh, kw = w.shape
    if cin != sum(src_channels):
        raise ValueError("conv operands carry %d channels, layer expects %d" % (sum(src_channels), cin))
    L = _ffi.lib()
    n = len(src_channels)
    ch = (ctypes.c_int * n)(*src_channels)
    elems = L.dkt_conv2d_packed_elems(ch, n, cout, kh, kw)
    if elems <= 0:
        raise _ffi.DktError("dkt_conv2d_packed_elems rejected the layer shape")
    wmax = float(w.detach().abs().max())
    # power-of-two scale putting max|w| in [2^12, 2^13): keeps w_lo out of the fp16 subnormals
    e = 12 - math.floor(math.log2(wmax)) if wmax > 0 else 0
    scale = 2.0 ** e
    p = _Packed()
    p.hi = torch.empty(elems, device=w.device, dtype=torch.float16)
    p.lo = torch.empty(elems, device=w.device, dtype=torch.float16)
    wc = w.detach().float().contiguous()
    rc = L.dkt_conv2d_pack_weights(wc.data_ptr(), ch, n, cout, kh, kw, scale, p.hi.data_ptr(), p.lo.data_ptr(),
                                   _ffi.device_of(w), _ffi.stream_of(w))
    _ffi.check(rc, "dkt_conv2d_pack_weights")
    p.inv_scale = 1.0 / scale
    p.bias = None if b is None else b.detach().float().contiguous()
    p.key = key
    cache[slot] = p
    return p


def clear_weight_cache(module):
    """Drops the packed fp16 weight images below `module` (needed only after writes
    that bypass the Parameter's version counter, e.g. ``weight.data.mul_()``)."""
    with _CACHE_LOCK:
        for m in module.modules():
            m.__dict__.pop("_dkt_packed", None)
            m.__dict__.pop("_dkt_folded", None)
            m.__dict__.pop("_dkt_stem7", None)
            m.__dict__.pop("_dkt_wt", None)
            m.__dict__.pop("_dkt_view", None)
            if hasattr(m, "_zr_cache"):
                m._zr_cache = None


def _dense(t):
    hw = t.shape[2] * t.shape[3]
    return t.stride(3) == 1 and t.stride(2) == t.shape[3] and t.stride(1) == hw


def _stride_of(layer):
    st = getattr(layer, "stride", 1)
    st = (st, st) if isinstance(st, int) else tuple(st)
    return st


def _plain_conv(layer):
    """groups == 1, dilation == 1, zero padding (what the kernels implement)."""
    dil = getattr(layer, "dilation", 1)
    dil = (dil, dil) if isinstance(dil, int) else tuple(dil)
    return (getattr(layer, "groups", 1) == 1 and dil == (1, 1)
            and getattr(layer, "padding_mode", "zeros") == "zeros")


def hip_eligible(layer):
    """True when `layer` runs on dkt_conv2d_f16s[_strided] under the current backend:
    1x1 / 3x3, padding K/2, stride 1 or 2, no groups / dilation."""
    kh, kw = layer.weight.shape[2:]
    pad = layer.padding
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    return (get_backend() in _PASSES and kh == kw and kh in (1, 3) and pad == (kh // 2, kw // 2)
            and _stride_of(layer) in ((1, 1), (2, 2)) and _plain_conv(layer))


def direct_eligible(layer):
    """The 7x7 stems (Cin <= 4, stride 1) run on dkt_conv2d_stem7 (matrix cores, K laid out over the
    taps).  The exact-fp32 direct kernel (dkt_conv2d_direct, 26 us for 2->64 @184x312) stays available
    as _conv2d_direct."""
    if get_backend() not in _PASSES or not _plain_conv(layer):
        return False
    cout, cin, kh, kw = layer.weight.shape
    pad = layer.padding
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    if kh != kw or pad != (kh // 2, kw // 2) or _stride_of(layer) != (1, 1):
        return False
    return kh == 7 and cin <= 4


def few_eligible(layer):
    """3x3, stride 1, padding 1, at most 4 output channels (flow_head.conv2 256 -> 2, disp_head.conv2 256 -> 1):
    an HBM-bound layer that runs on the exact-fp32 DMA-staged kernel (dkt_conv2d_direct) instead of padding its
    outputs to a 32-channel matrix-core tile.  (`_FEW_DIRECT = False` sends it back to dkt_conv2d_f16s.)"""
    if get_backend() not in _PASSES or not _plain_conv(layer) or not _FEW_DIRECT:
        return False
    cout, cin, kh, kw = layer.weight.shape
    pad = layer.padding
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    if not (kh == 3 and kw == 3 and pad == (1, 1) and _stride_of(layer) == (1, 1) and cout <= 4):
        return False
    # what launch_few (conv_direct.hip) can stage in 160 KB of LDS: 110592 B of patches + 384 B per (8-channel slice, output);
    # wider layers (Cin > 1104 / 552 / 272 for 1 / 2 / 3-4 outputs) stay on dkt_conv2d_f16s
    to = 1 if cout <= 1 else 2 if cout <= 2 else 4
    return 110592 + 384 * ((cin + 7) // 8) * to <= 160 * 1024


def _conv2d_direct(x, layer, relu, out):
    _ffi.require_gpu(x)
    _ffi.require_no_grad(x)
    if not _dense(x):
        x = x.contiguous()
    B, cin, H, W = x.shape
    cout, _, kh, kw = layer.weight.shape
    if out is None:
        out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    w = layer.weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    b = layer.bias
    rc = _ffi.lib().dkt_conv2d_direct(x.data_ptr(), x.stride(0), w.data_ptr(), None if b is None else b.data_ptr(),
                                      out.data_ptr(), out.stride(0), B, cin, cout, H, W, kh, kw, int(bool(relu)),
                                      _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_conv2d_direct")
    return out


def conv2d_accumulate(x, layer, out, diff=None):
    """out += conv(x) + bias for a few-output 3x3 layer (``few_eligible``): the head layer of the refinement loop adds its
    result to the running coordinates / disparity in its epilogue (one elementwise launch less per iteration).
    `out`: (B, Cout, H, W) view, dense per batch element."""
    _ffi.require_gpu(x, out)
    _ffi.require_no_grad(x)
    if not _dense(x):
        x = x.contiguous()
    B, cin, H, W = x.shape
    cout, _, kh, kw = layer.weight.shape
    if not _dense(out) or out.dtype != torch.float32 or tuple(out.shape) != (B, cout, H, W):
        raise ValueError("conv2d_accumulate(out=...) must be a dense-per-batch fp32 tensor of the result shape")
    w = layer.weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    b = layer.bias
    if diff is not None:
        # diff = (ref, dst): dst = out_new - ref, both shaped like `out` (dense per batch element)
        ref, dst = diff
        for t in (ref, dst):
            if not _dense(t) or t.dtype != torch.float32 or tuple(t.shape) != (B, cout, H, W):
                raise ValueError("conv2d_accumulate(diff=(ref, dst)): both must be dense-per-batch fp32 tensors shaped like out")
        rc = _ffi.lib().dkt_conv2d_direct_accumulate_diff(
            x.data_ptr(), x.stride(0), w.data_ptr(), None if b is None else b.data_ptr(), out.data_ptr(), out.stride(0),
            ref.data_ptr(), ref.stride(0), dst.data_ptr(), dst.stride(0), B, cin, cout, H, W, kh, kw,
            _ffi.device_of(x), _ffi.stream_of(x))
        _ffi.check(rc, "dkt_conv2d_direct_accumulate_diff")
        return out
    rc = _ffi.lib().dkt_conv2d_direct_accumulate(x.data_ptr(), x.stride(0), w.data_ptr(), None if b is None else b.data_ptr(),
                                                 out.data_ptr(), out.stride(0), B, cin, cout, H, W, kh, kw,
                                                 _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_conv2d_direct_accumulate")
    return out


def _conv2d_stem7(x, layer, relu, out):
    """7x7 stem (Cin <= 4) on the matrix cores (dkt_conv2d_stem7); packed weights cached on the layer."""
    _ffi.require_gpu(x)
    _ffi.require_no_grad(x)
    if not _dense(x):
        x = x.contiguous()
    B, cin, H, W = x.shape
    w, b = layer.weight, layer.bias
    cout = w.shape[0]
    _record_range(layer, [x])
    key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
    L = _ffi.lib()
    with _CACHE_LOCK:
        pk = _stem7_packed(layer, key, L)
    in_scale = 2.0 ** in_exp_of(layer)
    if out is None:
        out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    rc = L.dkt_conv2d_stem7(x.data_ptr(), x.stride(0), pk.hi.data_ptr(), pk.lo.data_ptr(),
                            None if pk.bias is None else pk.bias.data_ptr(), pk.inv_scale / in_scale, in_scale,
                            out.data_ptr(), out.stride(0), B, cin, cout, H, W, int(bool(relu)),
                            _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_conv2d_stem7")
    return out


def _stem7_packed(layer, key, L):
    w, b = layer.weight, layer.bias
    cout, cin = w.shape[:2]
    cache = layer.__dict__.setdefault("_dkt_stem7", {})
    pk = cache.get(str(w.device))
    if pk is None or pk.key != key:
        wmax = float(w.detach().abs().max())
        scale = 2.0 ** (12 - math.floor(math.log2(wmax)) if wmax > 0 else 0)
        pk = _Packed()
        n = L.dkt_conv2d_stem7_packed_elems(cout)
        pk.hi = torch.empty(n, device=w.device, dtype=torch.float16)
        pk.lo = torch.empty(n, device=w.device, dtype=torch.float16)
        wc = w.detach().float().contiguous()
        rc = L.dkt_conv2d_stem7_pack(wc.data_ptr(), cout, cin, scale, pk.hi.data_ptr(), pk.lo.data_ptr(),
                                     _ffi.device_of(w), _ffi.stream_of(w))
        _ffi.check(rc, "dkt_conv2d_stem7_pack")
        pk.inv_scale = 1.0 / scale
        pk.bias = None if b is None else b.detach().float().contiguous()
        pk.key = key
        cache[str(w.device)] = pk
    return pk


class _Operands:
    """The cat operands of one convolution marshalled for the C ABI (keeps them alive)."""

    def __init__(self, x, layer):
        srcs = list(x) if isinstance(x, (list, tuple)) else [x]
        if len(srcs) > 4:
            srcs = srcs[:3] + [torch.cat(srcs[3:], dim=1)]
        _ffi.require_gpu(*srcs)
        _ffi.require_no_grad(*srcs)
        self.srcs = srcs = [s if _dense(s) else s.contiguous() for s in srcs]
        _record_range(layer, srcs)
        self.B, _, self.H, self.W = srcs[0].shape
        chans = [int(s.shape[1]) for s in srcs]
        self.n = n = len(srcs)
        self.pk = _packed_weights(layer, chans)
        self.ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
        self.ch = (ctypes.c_int * n)(*chans)
        self.bs = (ctypes.c_long * n)(*[s.stride(0) for s in srcs])
        self.kh, self.kw = layer.weight.shape[2:]
        self.cout = layer.weight.shape[0]
        self.bias = None if self.pk.bias is None else self.pk.bias.data_ptr()
        self.device = srcs[0].device
        self.in_scale = 2.0 ** in_exp_of(layer)
        self.passes = _PASSES[get_backend()]

    def head(self):
        return (self.ptrs, self.ch, self.bs, self.n, self.pk.hi.data_ptr(), self.pk.lo.data_ptr(),
                self.bias, self.pk.inv_scale / self.in_scale, self.in_scale)


def _desc(op, out, relu=False, epilogue=0, e0=None, e1=None, h=None, out2=None):
    """dkt_conv_desc for one convolution of a paired launch (keeps `op` alive through the returned object)."""
    d = _ffi.ConvDesc()
    for i in range(op.n):
        d.src[i] = op.srcs[i].data_ptr()
        d.src_bstride[i] = op.srcs[i].stride(0)
        d.src_channels[i] = int(op.srcs[i].shape[1])
    d.nsrc = op.n
    d.w_hi, d.w_lo = op.pk.hi.data_ptr(), op.pk.lo.data_ptr()
    d.bias = op.bias
    d.out_scale, d.in_scale = op.pk.inv_scale / op.in_scale, op.in_scale
    d.out, d.out_bstride = out.data_ptr(), out.stride(0)
    d.B, d.H, d.W, d.Cout, d.KH, d.KW, d.relu = op.B, op.H, op.W, op.cout, op.kh, op.kw, int(bool(relu))
    d.epilogue = epilogue
    for name, t in (("e0", e0), ("e1", e1), ("h", h), ("out2", out2)):
        if t is not None:
            setattr(d, name, t.data_ptr())
            setattr(d, name + "_bstride", t.stride(0))
    d._keep = (op, out, e0, e1, h, out2)
    return d


def fused_eligible(layer, in_norm=False):
    """`layer` can take a residual operand (epilogue 3) and, with in_norm, the instance norm of its input in the
    staging (dkt_conv2d_f16s_desc): stride 1, 1x1 / 3x3 on the split-fp16 kernel; in_norm: 3x3, 33..128 outputs."""
    if not hip_eligible(layer) or _stride_of(layer) != (1, 1):
        return False
    cout, _, kh, _ = layer.weight.shape
    return (kh == 3 and 32 < cout <= 128) if in_norm else True


def conv2d_fused(x, layer, relu=False, residual=None, in_norm=None):
    """One stride-1 convolution with the neighbouring streaming passes folded in (core/extractor.py:46-60):
      in_norm : (B*Cin, 2) float (mean, 1/std) per input plane -> the layer reads relu((x - mean) * invstd);
      residual: tensor shaped like the result -> relu(residual + [relu](conv(x) + bias)).
    Bit-identical to the separate passes (same arithmetic, same order)."""
    op = _Operands(x, layer)
    out = torch.empty((op.B, op.cout, op.H, op.W), device=op.device, dtype=torch.float32)
    if residual is not None:
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError("conv2d_fused: residual shape %s != result shape %s" % (tuple(residual.shape), tuple(out.shape)))
        residual = residual if _dense(residual) else residual.contiguous()
        _ffi.require_gpu(residual)
        _ffi.require_no_grad(residual)
    d = _desc(op, out, relu=relu, epilogue=3 if residual is not None else 0, e0=residual)
    if in_norm is not None:
        if op.n != 1 or in_norm.dtype != torch.float32 or in_norm.numel() != 2 * op.B * int(op.srcs[0].shape[1]):
            raise ValueError("conv2d_fused: in_norm must hold (mean, 1/std) for every (batch, channel) plane of one operand")
        in_norm = in_norm.contiguous()
        d.in_norm = in_norm.data_ptr()
        d._keep = d._keep + (in_norm,)
    rc = _ffi.lib().dkt_conv2d_f16s_desc(ctypes.byref(d), op.passes, _ffi.device_of(out), _ffi.stream_of(out))
    _ffi.check(rc, "dkt_conv2d_f16s_desc")
    return out


class OutStats:
    """Instance-norm statistics of a convolution's output, accumulated in its epilogue: `part` is the partial-sum workspace
    the dkt_instance_norm_* kernels read (what dkt_instance_norm_stats would have written in a pass of its own)."""
    __slots__ = ("part", "planes", "hw")

    def __init__(self, part, planes, hw):
        self.part, self.planes, self.hw = part, planes, hw


def stats_eligible(layer):
    """`layer` can accumulate its output's instance-norm statistics in the epilogue (dkt_conv_desc.stats_ws): the split-fp16
    kernel's plain epilogue, stride 1 or 2."""
    return hip_eligible(layer) and _stride_of(layer) in ((1, 1), (2, 2)) and not direct_eligible(layer) and not few_eligible(layer)


def conv2d_stats(x, layer, in_norm=None, _ws=None):
    """conv(x) + bias with the statistics of its OWN output for the InstanceNorm2d that follows it
    (core/extractor.py:46-50): returns (out, OutStats).  `in_norm` as in conv2d_fused (stride 1 only).
    `_ws`: the per-(tile, wave row) scratch of dkt_conv2d_stats_ws_floats floats, handed in by the guarded-buffer test."""
    op = _Operands(x, layer)
    stride = _stride_of(layer)[0]
    Ho, Wo = (op.H - 1) // stride + 1, (op.W - 1) // stride + 1
    out = torch.empty((op.B, op.cout, Ho, Wo), device=op.device, dtype=torch.float32)
    L = _ffi.lib()
    d = _desc(op, out)
    d.stride = stride
    planes = op.B * op.cout
    n_ws = int(L.dkt_conv2d_stats_ws_floats(op.B, op.cout, Ho, Wo))
    ws = torch.empty(n_ws, device=op.device, dtype=torch.float32) if _ws is None else _ws
    if ws.numel() < n_ws or ws.dtype != torch.float32 or not ws.is_contiguous():
        raise ValueError("conv2d_stats: scratch of dkt_conv2d_stats_ws_floats floats expected")
    part = torch.empty(int(L.dkt_instance_norm_workspace(planes, Ho * Wo)), device=op.device, dtype=torch.uint8)
    d.stats_ws, d.stats_part = ws.data_ptr(), part.data_ptr()
    keep = (ws, part)
    if in_norm is not None:
        if stride != 1 or op.n != 1 or in_norm.dtype != torch.float32 or in_norm.numel() != 2 * op.B * int(op.srcs[0].shape[1]):
            raise ValueError("conv2d_stats: in_norm must hold (mean, 1/std) for every (batch, channel) plane of one operand")
        in_norm = in_norm.contiguous()
        d.in_norm = in_norm.data_ptr()
        keep = keep + (in_norm,)
    d._keep = d._keep + keep
    rc = L.dkt_conv2d_f16s_desc(ctypes.byref(d), op.passes, _ffi.device_of(out), _ffi.stream_of(out))
    _ffi.check(rc, "dkt_conv2d_f16s_desc")
    return out, OutStats(part, planes, Ho * Wo)


def pair_eligible(layer_a, layer_b):
    """Two stride-1 layers can share a launch (dkt_conv2d_f16s_pair): same filter size, same output-width class."""
    def cls(c):
        return 0 if c <= 32 else 1 if c <= 64 else 2 if c <= 128 else 3
    wa, wb = layer_a.weight, layer_b.weight
    return (hip_eligible(layer_a) and hip_eligible(layer_b) and _stride_of(layer_a) == (1, 1) == _stride_of(layer_b)
            and wa.shape[2] == wb.shape[2] and cls(wa.shape[0]) == cls(wb.shape[0]))


def _launch_pair(da, db, passes, ref):
    rc = _ffi.lib().dkt_conv2d_f16s_pair(ctypes.byref(da), ctypes.byref(db), passes, _ffi.device_of(ref), _ffi.stream_of(ref))
    _ffi.check(rc, "dkt_conv2d_f16s_pair")


def conv2d_pair(a, b):
    """Two independent stride-1 convolutions (bias + optional ReLU) in one launch.  a, b = (x, layer, relu);
    the layers must satisfy ``pair_eligible``.  Returns (y_a, y_b)."""
    res, descs, passes = [], [], None
    for x, layer, relu in (a, b):
        op = _Operands(x, layer)
        out = torch.empty((op.B, op.cout, op.H, op.W), device=op.device, dtype=torch.float32)
        descs.append(_desc(op, out, relu=relu))
        res.append(out)
        passes = op.passes
    _launch_pair(descs[0], descs[1], passes, res[0])
    return res


def conv2d_fused_pair(a, b):
    """conv2d_fused (residual epilogue) for two independent layers in one launch.  a, b = (x, layer, relu, residual);
    the layers must satisfy ``pair_eligible`` and ``fused_eligible``.  Returns (y_a, y_b)."""
    res, descs, passes = [], [], None
    for x, layer, relu, residual in (a, b):
        op = _Operands(x, layer)
        out = torch.empty((op.B, op.cout, op.H, op.W), device=op.device, dtype=torch.float32)
        if residual is not None:
            if tuple(residual.shape) != tuple(out.shape):
                raise ValueError("conv2d_fused_pair: residual shape %s != result shape %s" % (tuple(residual.shape), tuple(out.shape)))
            residual = residual if _dense(residual) else residual.contiguous()
            _ffi.require_gpu(residual)
            _ffi.require_no_grad(residual)
        descs.append(_desc(op, out, relu=relu, epilogue=3 if residual is not None else 0, e0=residual))
        res.append(out)
        passes = op.passes
    _launch_pair(descs[0], descs[1], passes, res[0])
    return res


def conv2d_gate_zr_pair(a, b):
    """conv2d_gate_zr for two independent GRUs in one launch.  a, b = (x, zr_layer, cz, cr, h);
    returns ((z_a, rh_a), (z_b, rh_b))."""
    res, descs, passes = [], [], None
    for x, zr_layer, cz, cr, h in (a, b):
        op = _Operands(x, zr_layer)
        ch = op.cout // 2
        z = torch.empty((op.B, ch, op.H, op.W), device=op.device, dtype=torch.float32)
        rh = torch.empty_like(z)
        descs.append(_desc(op, z, epilogue=1, e0=cz, e1=cr, h=h, out2=rh))
        res.append((z, rh))
        passes = op.passes
    _launch_pair(descs[0], descs[1], passes, res[0][0])
    return res


def conv2d_gate_out_pair(a, b):
    """conv2d_gate_out for two independent GRUs in one launch.  a, b = (x, q_layer, cq, z, h, out)."""
    res, descs, passes = [], [], None
    for x, q_layer, cq, z, h, out in (a, b):
        op = _Operands(x, q_layer)
        if out is None:
            out = torch.empty((op.B, op.cout, op.H, op.W), device=op.device, dtype=torch.float32)
        descs.append(_desc(op, out, epilogue=2, e0=cq, e1=z, h=h))
        res.append(out)
        passes = op.passes
    _launch_pair(descs[0], descs[1], passes, res[0])
    return res


def conv2d(x, layer, relu=False, out=None):
    """`out`: optional (B,Cout,H,W) fp32 destination whose batch elements are dense (e.g. a
    channel slice of a wider buffer -- replaces a torch.cat of the result)."""
    if isinstance(x, (list, tuple)) and len(x) == 1:
        x = x[0]
    if direct_eligible(layer) and not isinstance(x, (list, tuple)):
        return _conv2d_stem7(x, layer, relu, out)
    if few_eligible(layer) and not isinstance(x, (list, tuple)):
        return _conv2d_direct(x, layer, relu, out)
    if not hip_eligible(layer):
        y = _vendor(x, layer, relu)
        if out is not None:
            out.copy_(y)
            return out
        return y
    op = _Operands(x, layer)
    stride = _stride_of(layer)[0]
    Ho, Wo = (op.H - 1) // stride + 1, (op.W - 1) // stride + 1
    if out is None:
        out = torch.empty((op.B, op.cout, Ho, Wo), device=op.device, dtype=torch.float32)
    elif not _dense(out) or out.dtype != torch.float32 or tuple(out.shape) != (op.B, op.cout, Ho, Wo):
        raise ValueError("conv2d(out=...) must be a dense-per-batch fp32 tensor of the result shape")
    if stride == 1:
        rc = _ffi.lib().dkt_conv2d_f16s(*op.head(), out.data_ptr(), out.stride(0), op.B, op.H, op.W, op.cout,
                                        op.kh, op.kw, int(bool(relu)), op.passes,
                                        _ffi.device_of(out), _ffi.stream_of(out))
    else:
        rc = _ffi.lib().dkt_conv2d_f16s_strided(*op.head(), out.data_ptr(), out.stride(0), op.B, op.H, op.W, op.cout,
                                                op.kh, op.kw, stride, int(bool(relu)), op.passes,
                                                _ffi.device_of(out), _ffi.stream_of(out))
    _ffi.check(rc, "dkt_conv2d_f16s")
    return out


def conv2d_gate_zr(x, zr_layer, cz, cr, h):
    """ConvGRU first stage with the gates in the convolution epilogue:
    returns (z, r*h) = (sigmoid(convz(x)+cz), sigmoid(convr(x)+cr)*h); `zr_layer` is the
    merged convz|convr pair.  cz, cr, h: (B,Ch,H,W), dense per batch element."""
    op = _Operands(x, zr_layer)
    ch = op.cout // 2
    z = torch.empty((op.B, ch, op.H, op.W), device=op.device, dtype=torch.float32)
    rh = torch.empty_like(z)
    rc = _ffi.lib().dkt_conv2d_f16s_gate_zr(*op.head(), cz.data_ptr(), cz.stride(0), cr.data_ptr(), cr.stride(0),
                                            h.data_ptr(), h.stride(0), z.data_ptr(), z.stride(0),
                                            rh.data_ptr(), rh.stride(0), op.B, op.H, op.W, ch, op.kh, op.kw,
                                            op.passes, _ffi.device_of(z), _ffi.stream_of(z))
    _ffi.check(rc, "dkt_conv2d_f16s_gate_zr")
    return z, rh


def conv2d_gate_out(x, q_layer, cq, z, h, out=None):
    """ConvGRU second stage: (1-z)*h + z*tanh(convq(x)+cq), written to `out` (default: new
    tensor; `out` may be `h` itself for an in-place state update)."""
    op = _Operands(x, q_layer)
    if out is None:
        out = torch.empty((op.B, op.cout, op.H, op.W), device=op.device, dtype=torch.float32)
    rc = _ffi.lib().dkt_conv2d_f16s_gate_out(*op.head(), cq.data_ptr(), cq.stride(0), z.data_ptr(), z.stride(0),
                                             h.data_ptr(), h.stride(0), out.data_ptr(), out.stride(0),
                                             op.B, op.H, op.W, op.cout, op.kh, op.kw,
                                             op.passes, _ffi.device_of(out), _ffi.stream_of(out))
    _ffi.check(rc, "dkt_conv2d_f16s_gate_out")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Autograd of the stride-1 "same" convolution (training through the update operator, tools/ft_dkt.py:223-242): forward and
# the input gradient run on this library's convolution kernels (the input gradient of a stride-1 same convolution is the
# same convolution with the weights transposed over (Cout, Cin) and rotated by 180 degrees), the weight gradient -- a
# reduction over all pixels, a different loop nest -- on the vendor library (torch.nn.grad.conv2d_weight), the bias gradient
# is a sum.
# ---------------------------------------------------------------------------------------------------------------------
class _LayerShim:
    """Duck-types the `layer` argument of conv2d() for detached tensors inside the autograd function."""

    def __init__(self, weight, bias, padding):
        self.weight, self.bias, self.padding = weight, bias, padding
        self.stride, self.dilation, self.groups, self.padding_mode = (1, 1), (1, 1), 1, "zeros"


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        kh, kw = weight.shape[2:]
        pad = (kh // 2, kw // 2)
        with torch.no_grad():
            y = conv2d(x.detach(), _LayerShim(weight.detach(), None if bias is None else bias.detach(), pad), relu=relu)
        ctx.relu, ctx.pad, ctx.has_bias = bool(relu), pad, bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous().float()
        if ctx.relu:
            gy = gy * (y > 0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            with torch.no_grad():
                wt = weight.detach().transpose(0, 1).flip(2, 3).contiguous()
                gx = conv2d(gy, _LayerShim(wt, None, ctx.pad))
        if ctx.needs_input_grad[1]:
            gw = torch.nn.grad.conv2d_weight(x.detach(), weight.shape, gy, stride=1, padding=ctx.pad)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(0, 2, 3))
        return gx, gw, gb, None


def conv2d_autograd(x, layer, relu=False):
    """[relu](conv(x) + bias) for a stride-1 "same" layer (odd square kernel, no groups / dilation) as an autograd node:
    `x` a tensor or a list of tensors (the reference's torch.cat operands).  Other layers run as plain torch."""
    if isinstance(x, (list, tuple)):
        x = x[0] if len(x) == 1 else torch.cat(list(x), dim=1)
    w = layer.weight
    kh, kw = w.shape[2:]
    pad = layer.padding
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    if (not x.is_cuda or x.dtype != torch.float32 or kh != kw or kh % 2 == 0 or pad != (kh // 2, kw // 2)
            or _stride_of(layer) != (1, 1) or not _plain_conv(layer)):
        y = F.conv2d(x, w, layer.bias, stride=_stride_of(layer), padding=layer.padding,
                     dilation=getattr(layer, "dilation", 1), groups=getattr(layer, "groups", 1))
        return F.relu(y) if relu else y
    return _Conv2dFn.apply(x.contiguous(), w, layer.bias, relu)
