"""The 3x3 convolutions of the update operator on pre-split activations (csrc/conv_c8.hip).

``ActC8`` is a "C8S" activation tensor: fp16 (hi, lo) pairs, ``[B][2*ceil(C/16)][2][Hp][Wp][8]`` with a zero border
(include/dktstereo.h).  Producers write it from their epilogues (``conv2d_c8(..., out_c8=...)``, the resampling kernels);
``pack`` / ``unpack`` convert from / to fp32 NCHW for glue and tests.  Reference operators: ConvGRU
(core/update.py:16-32), BasicMotionEncoder (:64-85), FlowHead.conv1 (:9)."""
import ctypes
import math
import threading

import torch

from . import _ffi
from .conv import _CACHE_LOCK


def c8_dims(H, W):
    return (H + 7) // 8 * 8 + 2, (W + 31) // 32 * 32 + 2


class ActC8:
    """A C8S activation tensor.  ``t``: fp16 storage (B, G, 2, Hp, Wp, 8), zero outside the written interior.
    ``scale``: the power of two its values are multiplied by before the fp16 (hi, lo) split -- the format keeps 22 significant
    bits only while |x * scale| sits in fp16's normal range, so producers scale and consumers fold 1 / scale into their packed
    weights (`channel_scales`); ``tail`` trailing channels (flow / disparity next to features, core/update.py:85) may carry a
    scale of their own (``tail_scale``).  loop_c8.C8Loop.calibrate picks the scales from observed magnitudes."""
    __slots__ = ("t", "C", "H", "W", "scale", "tail", "tail_scale")

    def __init__(self, B, C, H, W, device, scale=1.0, tail=0):
        Hp, Wp = c8_dims(H, W)
        self.t = torch.zeros((B, 2 * ((C + 15) // 16), 2, Hp, Wp, 8), device=device, dtype=torch.float16)
        self.C, self.H, self.W, self.scale = C, H, W, float(scale)
        self.tail, self.tail_scale = int(tail), float(scale)

    def channel_scales(self):
        """((channels, scale), ...) segments in channel order."""
        if self.tail and self.tail_scale != self.scale:
            return ((self.C - self.tail, self.scale), (self.tail, self.tail_scale))
        return ((self.C, self.scale),)

    def absmax(self):
        """(max |x * scale| over the body channels, max |x * tail_scale| over the tail channels) as 0-dim fp32 tensors: what
        the hi halves hold (Inf where the scale overflowed fp16)."""
        hi = self.t[:, :, 0]                                      # (B, G, Hp, Wp, 8)
        if not self.tail:
            return hi.abs().max().float(), None
        c0 = self.C - self.tail
        g0, k0 = c0 // 8, c0 % 8
        tail = hi[:, g0:, :, :, :].abs()
        tmax = tail[:, 0, :, :, k0:].max() if tail.shape[1] == 1 else torch.maximum(tail[:, 0, :, :, k0:].max(), tail[:, 1:].max())
        body = torch.maximum(hi[:, :g0].abs().max(), tail[:, 0, :, :, :k0].max()) if k0 else hi[:, :g0].abs().max()
        return body.float(), tmax.float()

    @property
    def B(self):
        return self.t.shape[0]

    @property
    def bstride_bytes(self):
        return self.t.stride(0) * 2

    def data_ptr(self):
        return self.t.data_ptr()

    @property
    def device(self):
        return self.t.device


def _check_dst(what, dst, B, C, H, W, ch0=0):
    """The C8S producers write (B, [ch0, ch0 + C), H, W) of `dst` and derive its plane geometry from (H, W): a destination
    of another shape, or with fewer (padded) channels, would be written out of bounds."""
    if (dst.B, dst.H, dst.W) != (B, H, W) or (ch0 & 7) or ch0 < 0 or ch0 + C > 8 * dst.t.shape[1]:
        raise ValueError("%s: destination is (B=%d, %d padded channels, %dx%d), the result (B=%d, channels %d..%d, %dx%d)"
                         % (what, dst.B, 8 * dst.t.shape[1], dst.H, dst.W, B, ch0, ch0 + C - 1, H, W))


def pack(x, dst=None, ch0=0, scale=None):
    """fp32 NCHW -> channels [ch0, ch0 + C) of an ActC8 (a new one of exactly C channels by default)."""
    _ffi.require_gpu(x)
    B, C, H, W = x.shape
    if dst is None:
        dst = ActC8(B, C, H, W, x.device, 1.0 if scale is None else scale)
    _check_dst("conv_c8.pack", dst, B, C, H, W, ch0)
    x = x if (x.stride(3) == 1 and x.stride(2) == W and x.stride(1) == H * W) else x.contiguous()
    rc = _ffi.lib().dkt_act_c8_pack(x.data_ptr(), x.stride(0), dst.data_ptr(), dst.bstride_bytes, B, C, H, W, ch0,
                                    dst.scale, _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_act_c8_pack")
    return dst


def unpack(a, C=None, ch0=0):
    C = a.C - ch0 if C is None else C
    y = torch.empty((a.B, C, a.H, a.W), device=a.device, dtype=torch.float32)
    rc = _ffi.lib().dkt_act_c8_unpack(a.data_ptr(), a.bstride_bytes, y.data_ptr(), y.stride(0), a.B, C, a.H, a.W, ch0,
                                      a.scale, _ffi.device_of(y), _ffi.stream_of(y))
    _ffi.check(rc, "dkt_act_c8_unpack")
    return y


class _PackedC8:
    __slots__ = ("key", "img", "inv_scale", "bias")


#: packed images kept per (layer, device, operand split): one per set of operand scales in use.  Two captured loops over the
#: same modules (two threads driving replicas, nn.parallel.replicate shares every layer object) may have picked different
#: activation scales; with a single entry each of them would re-pack on every call, and the other's captured graph would keep
#: the address of a freed image.  Entries of other weight versions are dropped first.
_PACK_KEEP = 6


_PINS = threading.local()
_PASSES = threading.local()


class passes:
    """fp16 MFMA products per weight x activation block of the convolutions described inside the block (this thread):
    3 = fp32-class (default, the parity path), 2 = activations rounded to fp16, 1 = weights and activations rounded to fp16
    (dkt_conv_c8_desc.passes / dkt_gru_c8_desc.passes).  Reduced passes are what loop_c8's precision schedules run their
    early iterations on; nothing else takes them."""

    def __init__(self, n):
        if n not in (1, 2, 3):
            raise ValueError("passes: 1, 2 or 3")
        self.n = n

    def __enter__(self):
        self.prev = getattr(_PASSES, "n", 3)
        _PASSES.n = self.n
        return self

    def __exit__(self, *exc):
        _PASSES.n = self.prev
        return False


def current_passes():
    return getattr(_PASSES, "n", 3)


class pin_packs:
    """Every packed image looked up or made on this thread inside the block is appended to `keep` (strong references): a
    captured graph bakes the images' addresses in, while the per-layer caches evict beyond _PACK_KEEP entries -- the capturing
    loop holds what its graphs point to until it drops them (another thread's recalibrations could free them)."""

    def __init__(self, keep):
        self.keep = keep

    def __enter__(self):
        self.prev = getattr(_PINS, "keep", None)
        _PINS.keep = self.keep
        return self.keep

    def __exit__(self, *exc):
        _PINS.keep = self.prev
        return False


def _pinned(p):
    keep = getattr(_PINS, "keep", None)
    if keep is not None and p is not None:
        keep.append(p)
    return p


def _cache_get(cache, slot, key):
    for p in cache.get(slot, ()):
        if p.key == key:
            return _pinned(p)
    return None


def _cache_put(cache, slot, p, n_weight_keys):
    _pinned(p)
    keep = [q for q in cache.get(slot, ()) if q.key[:n_weight_keys] == p.key[:n_weight_keys]]
    cache[slot] = (keep + [p])[-_PACK_KEEP:]


def _in_scale_vector(seg_lists, device):
    """1 / scale per input channel for operands whose channel_scales() are `seg_lists` (None when every scale is 1)."""
    segs = [sg for lst in seg_lists for sg in lst]
    if all(sc == 1.0 for _, sc in segs):
        return None
    return torch.cat([torch.full((n,), 1.0 / sc, device=device, dtype=torch.float32) for n, sc in segs])


def packed_weights(layer, src_channels, src_scales=None):
    """Step images of `layer` for dkt_conv2d_c8, cached on the layer per device and operand split.  `src_scales`: the
    operands' channel_scales(); their inverses are folded into the weights (powers of two: exact)."""
    with _CACHE_LOCK:
        w, b = layer.weight, layer.bias
        scales = tuple(tuple(x) for x in src_scales) if src_scales is not None else None
        if scales is not None and all(sc == 1.0 for lst in scales for _, sc in lst):
            scales = None
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version), tuple(src_channels), scales)
        cache = layer.__dict__.setdefault("_dkt_packed_c8", {})
        slot = (str(w.device), tuple(src_channels))
        hit = _cache_get(cache, slot, key)
        if hit is not None:
            return hit
        cout, cin, kh, kw = w.shape
        if (kh, kw) != (3, 3) or cin != sum(src_channels):
            raise ValueError("conv2d_c8: 3x3 layers only, operands carry %d channels, layer expects %d" % (sum(src_channels), cin))
        L = _ffi.lib()
        n = len(src_channels)
        ch = (ctypes.c_int * n)(*src_channels)
        nbytes = L.dkt_conv_c8_packed_bytes(ch, n, cout)
        if nbytes <= 0:
            raise _ffi.DktError("dkt_conv_c8_packed_bytes rejected the layer shape")
        wc = w.detach().float()
        if scales is not None:
            wc = wc * _in_scale_vector(scales, w.device).view(1, -1, 1, 1)
        wc = wc.contiguous()
        wmax = float(wc.abs().max())
        e = 12 - math.floor(math.log2(wmax)) if wmax > 0 else 0       # max|w| in [2^12, 2^13): w_lo stays normal
        p = _PackedC8()
        p.img = torch.zeros(nbytes // 2, device=w.device, dtype=torch.float16)
        rc = L.dkt_conv_c8_pack_weights(wc.data_ptr(), ch, n, cout, 2.0 ** e, p.img.data_ptr(), _ffi.device_of(w), _ffi.stream_of(w))
        _ffi.check(rc, "dkt_conv_c8_pack_weights")
        p.inv_scale = 2.0 ** -e
        p.bias = None if b is None else b.detach().float().contiguous()
        p.key = key
        _cache_put(cache, slot, p, 3)
        return p


def _pack_raw(w, src_channels):
    """(Cout, Cin, 3, 3) fp32 -> (step images, 1 / weight scale) for sources of `src_channels` channels (Cin in that order)."""
    L = _ffi.lib()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if tuple(w.shape[2:]) != (3, 3) or cin != sum(src_channels):
        raise ValueError("conv_c8: 3x3 layers only, operands carry %d channels, layer expects %d" % (sum(src_channels), cin))
    n = len(src_channels)
    ch = (ctypes.c_int * n)(*src_channels)
    nbytes = L.dkt_conv_c8_packed_bytes(ch, n, cout)
    if nbytes <= 0:
        raise _ffi.DktError("dkt_conv_c8_packed_bytes rejected the layer shape")
    wmax = float(w.abs().max())
    e = 12 - math.floor(math.log2(wmax)) if wmax > 0 else 0
    img = torch.zeros(nbytes // 2, device=w.device, dtype=torch.float16)
    wc = w.float().contiguous()
    rc = L.dkt_conv_c8_pack_weights(wc.data_ptr(), ch, n, cout, 2.0 ** e, img.data_ptr(), _ffi.device_of(w), _ffi.stream_of(w))
    _ffi.check(rc, "dkt_conv_c8_pack_weights")
    return img, 2.0 ** -e


class _PackedGru:
    __slots__ = ("key", "wzr", "wq", "inv_zr", "inv_q", "bz", "br", "bq")


def gru_packed(gru, x_channels, h_scales=None, x_scales=None):
    """Weights of one ConvGRU (core/update.py:16-21) for dkt_gru_c8: the z|r image with its output channels interleaved in
    blocks of 32 (a wave holds z and r of the same hidden channels), the q image with its input channels reordered to
    [x... | r*h] (the x chunks are consumed before the neighbours' r*h is needed).  `h_scales` / `x_scales`: channel_scales()
    of the state (= of r*h) and of the x operands, folded into the weights.  Cached on the module per device."""
    with _CACHE_LOCK:
        ps = [gru.convz.weight, gru.convr.weight, gru.convq.weight, gru.convz.bias, gru.convr.bias, gru.convq.bias]
        hs = tuple(h_scales) if h_scales is not None else ((128, 1.0),)
        xs = tuple(tuple(x) for x in x_scales) if x_scales is not None else tuple(((c, 1.0),) for c in x_channels)
        key = tuple((p.data_ptr(), p._version) for p in ps) + (tuple(x_channels), hs, xs)
        cache = gru.__dict__.setdefault("_dkt_gru_c8", {})
        slot = (str(ps[0].device), tuple(x_channels))
        hit = _cache_get(cache, slot, key)
        if hit is not None:
            return hit
        wz, wr, wq = (p.detach().float() for p in ps[:3])
        ch, cin = int(wz.shape[0]), int(wz.shape[1])
        if ch != 128 or cin != ch + sum(x_channels):
            raise ValueError("gru_c8: hidden size 128, operands of %d channels expected" % (cin - ch))
        inv = _in_scale_vector([hs] + list(xs), wz.device)            # the reference's input order [h | x...] (= [r*h | x...])
        if inv is not None:
            wz, wr, wq = (w * inv.view(1, -1, 1, 1) for w in (wz, wr, wq))
        wzr = torch.stack([wz.view(ch // 32, 32, cin, 3, 3), wr.view(ch // 32, 32, cin, 3, 3)], dim=1).reshape(2 * ch, cin, 3, 3)
        wq2 = torch.cat([wq[:, ch:], wq[:, :ch]], dim=1)
        p = _PackedGru()
        p.wzr, p.inv_zr = _pack_raw(wzr, [ch] + list(x_channels))
        p.wq, p.inv_q = _pack_raw(wq2, list(x_channels) + [ch])
        p.bz, p.br, p.bq = (b.detach().float().contiguous() for b in ps[3:])
        p.key = key
        _cache_put(cache, slot, p, 6)
        return p


def gru_flags(B, H, W, device):
    """The per-tile flag words of dkt_gru_c8 for one (operator, shape) pair (zero-initialised; every launch increments them)."""
    return torch.zeros(int(_ffi.lib().dkt_gru_c8_flag_words(B, H, W)), device=device, dtype=torch.int32)


def gru_desc(gru, h_c8, xs, rh_c8, cz, cr, cq, h, flags):
    """dkt_gru_c8_desc of one ConvGRU step: h (fp32 NCHW) and its C8S twin h_c8 are updated in place."""
    xs = list(xs)
    for s in xs + [rh_c8]:
        if (s.H, s.W, s.B) != (h_c8.H, h_c8.W, h_c8.B):
            raise ValueError("gru_c8: operands must share batch and size")
    if rh_c8.scale != h_c8.scale or h_c8.tail or rh_c8.tail:
        raise ValueError("gru_c8: the state and the r*h scratch carry one common scale")
    HW = h_c8.H * h_c8.W
    for t in (cz, cr, cq, h):
        if t.shape[1:] != (128, h_c8.H, h_c8.W) or t.stride(3) != 1 or t.stride(2) != h_c8.W or t.stride(1) != HW:
            raise ValueError("gru_c8: context terms and state are fp32 (B, 128, H, W), dense per batch item")
    pk = gru_packed(gru, [s.C for s in xs], h_c8.channel_scales(), [s.channel_scales() for s in xs])
    d = _ffi.GruC8Desc()
    d.h_c8, d.h_c8_bstride = h_c8.data_ptr(), h_c8.bstride_bytes
    for i, s in enumerate(xs):
        d.x[i], d.x_bstride[i], d.x_channels[i] = s.data_ptr(), s.bstride_bytes, s.C
    d.nx = len(xs)
    d.rh_c8, d.rh_c8_bstride = rh_c8.data_ptr(), rh_c8.bstride_bytes
    d.w_zr, d.w_q = pk.wzr.data_ptr(), pk.wq.data_ptr()
    d.bz, d.br, d.bq = pk.bz.data_ptr(), pk.br.data_ptr(), pk.bq.data_ptr()
    d.cz, d.cr, d.cq = cz.data_ptr(), cr.data_ptr(), cq.data_ptr()
    d.cz_bstride, d.cr_bstride, d.cq_bstride = cz.stride(0), cr.stride(0), cq.stride(0)
    d.h, d.h_bstride = h.data_ptr(), h.stride(0)
    d.scale_zr, d.scale_q, d.act_scale = pk.inv_zr, pk.inv_q, h_c8.scale      # (operand scales live in the packed weights)
    d.B, d.H, d.W, d.hidden = h_c8.B, h_c8.H, h_c8.W, 128
    d.flags = flags.data_ptr()
    d.passes = current_passes()
    if d.passes == 1 and sum((s.C + 15) // 16 for s in xs) % 2:
        d.passes = 2                         # (the one-product kernel walks the x chunks in pairs, as conv_c8.desc)
    d._keep = (h_c8, xs, rh_c8, pk, cz, cr, cq, h, flags)
    return d


def gru_launch(d0, d1=None, err=None, ref=None):
    """One launch for one ConvGRU step (d1: a second, independent one riding in the same launch).  Returns False when the
    device cannot run this shape in one launch (the caller falls back to gate_zr + gate_out)."""
    ref = d0._keep[7] if ref is None else ref
    L = _ffi.lib()
    e = None if err is None else err.data_ptr()
    if d1 is not None:
        d0.passes = d1.passes = max(d0.passes, d1.passes)          # (one instantiation runs both problems)
    if d1 is None:
        rc = L.dkt_gru_c8(ctypes.byref(d0), e, _ffi.device_of(ref), _ffi.stream_of(ref))
    else:
        rc = L.dkt_gru_c8_pair(ctypes.byref(d0), ctypes.byref(d1), e, _ffi.device_of(ref), _ffi.stream_of(ref))
    if rc == _ffi.E_UNSUPPORTED:
        return False
    _ffi.check(rc, "dkt_gru_c8")
    return True


def to_c4(x):
    """fp32 NCHW -> the "C4" layout [B][ceil(C/4)][H][W][4] of the epilogue-side tensors (gate operands, state, z)."""
    B, C, H, W = x.shape
    if C % 4:
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 4 - C % 4))
    return x.view(B, -1, 4, H, W).permute(0, 1, 3, 4, 2).contiguous()


def from_c4(x, C=None):
    B, G, H, W, _ = x.shape
    y = x.permute(0, 1, 4, 2, 3).reshape(B, G * 4, H, W)
    return y if C is None or C == G * 4 else y[:, :C]


def desc(srcs, layer, relu=False, out=None, out_c8=None, out_c8_ch0=0, epilogue=0, e0=None, e1=None, h=None,
         out2=None, out2_c8=None, out2_c8_ch0=0, tail=None, f32_c4=False, head_w=None, head_out=None):
    """dkt_conv_c8_desc; keeps its tensors alive through the returned object."""
    srcs = list(srcs)
    s0 = srcs[0]
    for s in srcs:
        if (s.H, s.W, s.B) != (s0.H, s0.W, s0.B):
            raise ValueError("conv2d_c8: operands must share batch and size")
    pk = packed_weights(layer, [s.C for s in srcs], [s.channel_scales() for s in srcs])
    d = _ffi.ConvC8Desc()
    for i, s in enumerate(srcs):
        d.src[i] = s.data_ptr()
        d.src_bstride[i] = s.bstride_bytes
        d.src_channels[i] = s.C
    d.nsrc = len(srcs)
    d.w = pk.img.data_ptr()
    d.bias = None if pk.bias is None else pk.bias.data_ptr()
    d.out_scale = pk.inv_scale               # (the operands' scales are folded into the packed weights)
    d.act_scale = 1.0
    d.B, d.H, d.W, d.Cout, d.relu = s0.B, s0.H, s0.W, int(layer.weight.shape[0]), int(bool(relu))
    d.epilogue = epilogue
    if out is not None:
        d.out, d.out_bstride = out.data_ptr(), out.stride(0)
    for name, a, c0 in (("out_c8", out_c8, out_c8_ch0), ("out2_c8", out2_c8, out2_c8_ch0)):
        if a is not None:
            if (a.H, a.W, a.B) != (s0.H, s0.W, s0.B):
                raise ValueError("conv2d_c8: C8S destination of another size")
            setattr(d, name, a.data_ptr())
            setattr(d, name + "_bstride", a.bstride_bytes)
            setattr(d, name + "_ch0", c0)
            d.act_scale = a.scale
            if name == "out_c8" and tail is not None and a.tail:
                if a.tail != int(tail.shape[1]) or c0 + int(layer.weight.shape[0]) + a.tail != a.C:
                    raise ValueError("conv2d_c8: the destination's tail channels are not where this layer appends them")
                d.tail_scale = a.tail_scale
    for name, t in (("e0", e0), ("e1", e1), ("h", h), ("out2", out2), ("tail", tail)):
        if t is not None:
            setattr(d, name, t.data_ptr())
            setattr(d, name + "_bstride", t.stride(0))
    if tail is not None:
        d.tail_channels = int(tail.shape[1])
    d.f32_c4 = int(bool(f32_c4))
    d.passes = current_passes()
    if d.passes == 1 and sum((s.C + 15) // 16 for s in srcs) % 2:
        d.passes = 2                         # (the one-pass kernel walks its 16-channel chunks in pairs)
    if head_w is not None:
        d.head_w, d.head_out, d.head_out_bstride, d.head_outputs = head_w.data_ptr(), head_out.data_ptr(), head_out.stride(0), int(head_w.shape[0])
    d._keep = (srcs, pk, out, out_c8, e0, e1, h, out2, out2_c8, tail, head_w, head_out)
    return d


def launch(d, ref, cfg=0):
    rc = _ffi.lib().dkt_conv2d_c8(ctypes.byref(d), cfg, _ffi.device_of(ref), _ffi.stream_of(ref))
    _ffi.check(rc, "dkt_conv2d_c8")


def launch_pair(d0, d1, ref, cfg):
    d0.passes = d1.passes = max(d0.passes, d1.passes)          # (one instantiation runs both problems)
    rc = _ffi.lib().dkt_conv2d_c8_pair(ctypes.byref(d0), ctypes.byref(d1), cfg, _ffi.device_of(ref), _ffi.stream_of(ref))
    _ffi.check(rc, "dkt_conv2d_c8_pair")


def chain_flags(d0, cfg0, nprob, device):
    """The flag words of one dkt_conv2d_c8_chain pair of layers (zero-initialised; every launch increments them)."""
    n = int(_ffi.lib().dkt_conv2d_c8_chain_flag_words(ctypes.byref(d0), cfg0, nprob))
    if n <= 0:
        raise _ffi.DktError("dkt_conv2d_c8_chain_flag_words rejected the stage")
    return torch.zeros(n, device=device, dtype=torch.int32)


def launch_chain(d0a, d0b, cfg0, d1, cfg1, flags, ref, err=None, max_blocks=0, timing_only=False):
    """Two dependent layers in one launch (dkt_conv2d_c8_chain): stage 0 = d0a [| d0b], stage 1 = d1 reading stage 0's
    C8S output behind a flag round.  Returns False when the shapes are not covered (the caller launches them one by one)."""
    p = max(d.passes for d in (d0a, d0b, d1) if d is not None)
    for d in (d0a, d0b, d1):
        if d is not None:
            d.passes = p
    rc = _ffi.lib().dkt_conv2d_c8_chain(ctypes.byref(d0a), None if d0b is None else ctypes.byref(d0b), cfg0, ctypes.byref(d1), cfg1,
                                        flags.data_ptr(), None if err is None else err.data_ptr(), max_blocks, int(bool(timing_only)),
                                        _ffi.device_of(ref), _ffi.stream_of(ref))
    if rc == _ffi.E_UNSUPPORTED:
        return False
    _ffi.check(rc, "dkt_conv2d_c8_chain")
    return True


def conv2d_c8(srcs, layer, relu=False, out=None, out_c8=None, out_c8_ch0=0, tail=None, cfg=0):
    """conv(torch.cat(srcs)) + bias [ReLU] -> fp32 NCHW `out` (allocated when neither destination is given) and / or
    channels [out_c8_ch0, ...) of the ActC8 `out_c8`; `tail`: fp32 NCHW channels appended behind the result in
    `out_c8` (the reference's torch.cat([out, flow]), core/update.py:85)."""
    s0 = srcs[0]
    if out is None and out_c8 is None:
        out = torch.empty((s0.B, int(layer.weight.shape[0]), s0.H, s0.W), device=s0.device, dtype=torch.float32)
    d = desc(srcs, layer, relu=relu, out=out, out_c8=out_c8, out_c8_ch0=out_c8_ch0, tail=tail)
    launch(d, s0.t, cfg)
    return out if out is not None else out_c8


def gate_zr(srcs, zr_layer, cz, cr, h, rh_c8=None, rh=None, cfg=0, f32_c4=False):
    """ConvGRU first stage (core/update.py:27-29): z = sigmoid(convz(hx)+cz) as fp32 NCHW, r*h into the ActC8 `rh_c8`
    (the q convolution's operand) and / or the fp32 tensor `rh`."""
    s0 = srcs[0]
    ch = int(zr_layer.weight.shape[0]) // 2
    z = torch.empty_like(h) if f32_c4 else torch.empty((s0.B, ch, s0.H, s0.W), device=s0.device, dtype=torch.float32)
    d = desc(srcs, zr_layer, out=z, epilogue=1, e0=cz, e1=cr, h=h, out2=rh, out2_c8=rh_c8, f32_c4=f32_c4)
    launch(d, z, cfg)
    return z


def gate_out(srcs, q_layer, cq, z, h, out, out_c8=None, cfg=0, f32_c4=False):
    """ConvGRU second stage (core/update.py:30-31): h' = (1-z) h + z tanh(convq(rhx)+cq) -> `out` (may be h) and `out_c8`."""
    d = desc(srcs, q_layer, out=out, out_c8=out_c8, epilogue=2, e0=cq, e1=z, h=h, f32_c4=f32_c4)
    launch(d, out, cfg)
    return out


def pool2x_c8(x, dst, ch0=0):
    """dst[ch0 : ch0 + C] = C8S(avg_pool2d(x, 3, stride=2, padding=1)) (core/update.py:87-88), x fp32 NCHW dense per batch."""
    B, C, H, W = x.shape
    _check_dst("pool2x_c8", dst, B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, ch0)
    rc = _ffi.lib().dkt_pool2x_c8(x.data_ptr(), x.stride(0), dst.data_ptr(), dst.bstride_bytes, B, C, H, W, ch0, dst.scale,
                                  _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_pool2x_c8")
    return dst


def interp_c8(x, dst, ch0=0):
    """dst[ch0 : ch0 + C] = C8S(F.interpolate(x, (dst.H, dst.W), mode="bilinear", align_corners=True)) (core/update.py:93-95)."""
    B, C, H, W = x.shape
    _check_dst("interp_c8", dst, B, C, dst.H, dst.W, ch0)
    rc = _ffi.lib().dkt_interp_c8(x.data_ptr(), x.stride(0), dst.data_ptr(), dst.bstride_bytes, B, C, H, W, dst.H, dst.W, ch0,
                                  dst.scale, _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_interp_c8")
    return dst


def _resample_job(kind, x, dst, ch0):
    B, C, H, W = x.shape
    return _ffi.ResampleC8Job(x=x.data_ptr(), x_bstride=x.stride(0), dst=dst.data_ptr(), dst_bstride_bytes=dst.bstride_bytes,
                              B=B, C=C, H=H, W=W, Ho=dst.H, Wo=dst.W, ch0=ch0, kind=kind, scale=dst.scale)


def resample_pair_c8(job0, job1):
    """Two resampling jobs in one launch (dkt_resample_pair_c8); a job = ("pool" | "interp", x, dst[, ch0]): pool2x_c8 /
    interp_c8 of x into dst.  Bit-identical to the single launches."""
    jobs = []
    for kind, x, dst, *rest in (job0, job1):
        if kind == "pool" and (dst.H, dst.W) != ((x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1):
            raise ValueError("resample_pair_c8: pool2x destination is %dx%d for a %dx%d source" % (dst.H, dst.W, x.shape[2], x.shape[3]))
        _check_dst("resample_pair_c8", dst, x.shape[0], x.shape[1], dst.H, dst.W, rest[0] if rest else 0)
        jobs.append(_resample_job(0 if kind == "pool" else 1, x, dst, rest[0] if rest else 0))
    x0 = job0[1]
    rc = _ffi.lib().dkt_resample_pair_c8(ctypes.byref(jobs[0]), ctypes.byref(jobs[1]), _ffi.device_of(x0), _ffi.stream_of(x0))
    _ffi.check(rc, "dkt_resample_pair_c8")


def stem7_c8(x, layer, dst, relu=True, ch0=0):
    """dst[ch0 : ...] = C8S([relu](conv7x7(x))) for the 2- / 1-channel stems (core/update.py:75)."""
    from . import conv as _conv
    B, cin, H, W = x.shape
    w, b = layer.weight, layer.bias
    key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
    L = _ffi.lib()
    with _CACHE_LOCK:
        pk = _conv._stem7_packed(layer, key, L)
    in_scale = 2.0 ** _conv.in_exp_of(layer)
    if int(w.shape[0]) % 64:
        raise ValueError("stem7_c8: the C8S store writes whole 64-channel blocks (layer has %d outputs)" % int(w.shape[0]))
    _check_dst("stem7_c8", dst, B, int(w.shape[0]), H, W, ch0)
    rc = L.dkt_conv2d_stem7_c8(x.data_ptr(), x.stride(0), pk.hi.data_ptr(), pk.lo.data_ptr(),
                               None if pk.bias is None else pk.bias.data_ptr(), pk.inv_scale / in_scale, in_scale,
                               dst.data_ptr(), dst.bstride_bytes, ch0, dst.scale, B, cin, int(w.shape[0]), H, W, int(bool(relu)),
                               _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_conv2d_stem7_c8")
    return dst


def _head_weights(layer2):
    """(n_out, Cout, 3, 3) weights of the head's second layer as [n_out][Cout][12] fp32 (cached per device / version)."""
    w = layer2.weight
    key = (w.data_ptr(), w._version)
    with _CACHE_LOCK:
        cache = layer2.__dict__.setdefault("_dkt_head_w", {})
        hit = cache.get(str(w.device))
        if hit is None or hit[0] != key:
            t = torch.zeros((w.shape[0], w.shape[1], 12), device=w.device, dtype=torch.float32)
            t[:, :, :9] = w.detach().float().reshape(w.shape[0], w.shape[1], 9)
            hit = cache[str(w.device)] = (key, t)
        return hit[1]


def head_planes(srcs, layer1, layer2, cfg=2):
    """First half of `head`: layer1 with the per-tap projections against layer2's weights in its epilogue (epilogue 3).
    Returns (planes (B, n_out * n_co * 9, H, W), n_co); dkt_head_finish or dkt_motion_front_c8 sums the shifted planes."""
    s0 = srcs[0]
    hw = _head_weights(layer2)
    nout = int(hw.shape[0])
    n_co = _ffi.lib().dkt_conv2d_c8_head_blocks(int(layer1.weight.shape[0]), cfg)
    if n_co <= 0:
        raise _ffi.DktError("conv_c8.head: tile shape %d cannot run the head epilogue" % cfg)
    planes = torch.empty((s0.B, nout * n_co * 9, s0.H, s0.W), device=s0.device, dtype=torch.float32)
    d = desc(srcs, layer1, relu=True, epilogue=3, head_w=hw, head_out=planes)
    launch(d, planes, cfg)
    return planes, n_co


def motion_front_supported(corr, enc):
    """True when dkt_motion_front_c8 covers this correlation block (skewed four-level pyramid, radius 3 / 4) and motion
    encoder (1x1 convc1 with <= 64 outputs on the lookup, 7x7 convf1 on a <= 4 channel flow)."""
    w, wf = enc.convc1.weight, enc.convf1.weight
    K = 2 * getattr(corr, "radius", 0) + 1
    return (getattr(corr, "_skew", None) is not None and getattr(corr, "num_levels", 0) == 4 and corr.radius in (3, 4)
            and (corr._w2 >> 3) >= 2 and tuple(w.shape[2:]) == (1, 1) and w.shape[0] <= 64 and w.shape[1] == 4 * K
            and tuple(wf.shape[2:]) == (7, 7) and wf.shape[1] <= 4 and tuple(enc.convf1.padding) == (3, 3))


def motion_front(corr, planes, n_co, head_bias, x_old, x_new, x0, flow, convc1, cor, convf1, flo):
    """One launch for the step between the flow head and the motion encoder's 3x3 layers (dkt_motion_front_c8):
    x_new = x_old + sum of the head's shifted `planes` + head_bias (what dkt_head_finish computes), flow[:, 0] = x_new - x0,
    cor = C8S(relu(convc1(corr(x_new)))), flo = C8S(relu(convf1(flow))).  x_old, x_new, x0: (B, 1, H, W) views (dense rows)."""
    from . import conv as _conv
    from .corr import _kmajor_weight
    B, _, H, W = x_old.shape
    for t in (x_old, x_new, x0):
        if t.shape != x_old.shape or t.stride(3) != 1 or t.stride(2) != W:
            raise ValueError("motion_front: coordinate planes must be (B, 1, H, W) with dense rows")
    if flow.shape[0] != B or tuple(flow.shape[2:]) != (H, W) or not flow[0].is_contiguous() or flow.shape[1] != convf1.weight.shape[1]:
        raise ValueError("motion_front: flow must be (B, Cin of convf1, H, W), dense per batch item")
    if (cor.B, cor.H, cor.W) != (B, H, W) or (flo.B, flo.H, flo.W) != (B, H, W):
        raise ValueError("motion_front: C8S destinations of another shape")
    if x_new.data_ptr() == x_old.data_ptr():
        raise ValueError("motion_front: x_new must not alias x_old")
    L = _ffi.lib()
    wf, bf = convf1.weight, convf1.bias
    key = (wf.data_ptr(), wf._version, None if bf is None else (bf.data_ptr(), bf._version))
    with _CACHE_LOCK:
        pk = _conv._stem7_packed(convf1, key, L)
    in_scale = 2.0 ** _conv.in_exp_of(convf1)
    wm = _kmajor_weight(convc1)
    bc = convc1.bias
    skew = _ffi.ptr_array(corr._skew)
    d = _ffi.MotionFrontDesc(
        skew=skew, planes=planes.data_ptr(), planes_bstride=planes.stride(0), n_co=n_co,
        head_bias=None if head_bias is None else head_bias.detach().data_ptr(),
        x_old=x_old.data_ptr(), x_old_bstride=x_old.stride(0), x_new=x_new.data_ptr(), x_new_bstride=x_new.stride(0),
        x0=x0.data_ptr(), x0_bstride=x0.stride(0), flow=flow.data_ptr(), flow_bstride=flow.stride(0),
        w_cor=wm.data_ptr(), b_cor=None if bc is None else bc.detach().data_ptr(), cor_channels=int(convc1.weight.shape[0]),
        cor_c8=cor.data_ptr(), cor_c8_bstride_bytes=cor.bstride_bytes, cor_c8_ch0=0, cor_act_scale=cor.scale,
        stem_w_hi=pk.hi.data_ptr(), stem_w_lo=pk.lo.data_ptr(), stem_bias=None if pk.bias is None else pk.bias.data_ptr(),
        stem_out_scale=pk.inv_scale / in_scale, stem_in_scale=in_scale, stem_cin=int(wf.shape[1]), stem_cout=int(wf.shape[0]),
        flo_c8=flo.data_ptr(), flo_c8_bstride_bytes=flo.bstride_bytes, flo_c8_ch0=0, flo_act_scale=flo.scale,
        B=B, H=H, W1=W, W2=corr._w2, L=corr.num_levels, r=corr.radius)
    rc = L.dkt_motion_front_c8(ctypes.byref(d), _ffi.device_of(x_old), _ffi.stream_of(x_old))
    _ffi.check(rc, "dkt_motion_front_c8")


def head(srcs, layer1, layer2, target, diff=None, cfg=2):
    """target += layer2(relu(layer1(cat(srcs)))) for the flow / disparity head (core/update.py:6-14; raft_stereo.py:165-168),
    `layer2` a 3x3 layer with 1 or 2 outputs: the hidden tensor is never written -- layer1's epilogue reduces it against
    layer2's weights tap by tap (epilogue 3), dkt_head_finish adds the shifted planes.  diff = (ref, dst): dst = target - ref."""
    s0 = srcs[0]
    L = _ffi.lib()
    planes, n_co = head_planes(srcs, layer1, layer2, cfg)
    nout = planes.shape[1] // (n_co * 9)
    b2 = layer2.bias
    ref, dst = diff if diff is not None else (None, None)
    rc = L.dkt_head_finish(planes.data_ptr(), planes.stride(0), n_co, None if b2 is None else b2.detach().data_ptr(),
                           target.data_ptr(), target.stride(0), None if ref is None else ref.data_ptr(),
                           0 if ref is None else ref.stride(0), None if dst is None else dst.data_ptr(),
                           0 if dst is None else dst.stride(0), s0.B, nout, s0.H, s0.W, _ffi.device_of(planes), _ffi.stream_of(planes))
    _ffi.check(rc, "dkt_head_finish")
    return target


def residual_c8(srcs, layer, res, relu=True, out=None, out_c8=None, cfg=0):
    """relu(res + [relu](conv(cat(srcs)) + bias)): the tail of a residual block whose norm is folded into `layer`
    (core/extractor.py:52-60) in the convolution's epilogue (epilogue 4).  `res`: fp32 NCHW; `out` (fp32 NCHW, may be
    `res` itself) and / or `out_c8`."""
    s0 = srcs[0]
    if out is None and out_c8 is None:
        out = torch.empty((s0.B, int(layer.weight.shape[0]), s0.H, s0.W), device=s0.device, dtype=torch.float32)
    d = desc(srcs, layer, relu=relu, out=out, out_c8=out_c8, epilogue=4, e0=res)
    launch(d, s0.t, cfg)
    return out if out is not None else out_c8


def stem7_dual(x, layer, out, dst, relu=True):
    """out = [relu](conv7x7(x)) as fp32 NCHW and dst = the same as C8S, one launch (the encoders' first layer,
    core/extractor.py:140-142 / :167-171 with an eval-mode BatchNorm folded into `layer`)."""
    from . import conv as _conv
    B, cin, H, W = x.shape
    w, b = layer.weight, layer.bias
    key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
    L = _ffi.lib()
    with _CACHE_LOCK:
        pk = _conv._stem7_packed(layer, key, L)
    in_scale = 2.0 ** _conv.in_exp_of(layer)
    rc = L.dkt_conv2d_stem7_dual(x.data_ptr(), x.stride(0), pk.hi.data_ptr(), pk.lo.data_ptr(),
                                 None if pk.bias is None else pk.bias.data_ptr(), pk.inv_scale / in_scale, in_scale,
                                 out.data_ptr(), out.stride(0), dst.data_ptr(), dst.bstride_bytes, 0, dst.scale,
                                 B, cin, int(w.shape[0]), H, W, int(bool(relu)), _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_conv2d_stem7_dual")
    return out


def norm_join_c8(c, c_params, c_relu=True, a=None, a_params=None, a_relu=False, y=None, dst=None, ch0=0):
    """t = [relu]((c - mean) * invstd); with `a`: t = relu(a' + t), a' = a or [relu](normalised a) when `a_params` is given
    (the instance-norm glue of core/extractor.py:21-60) -> fp32 `y` and / or C8S `dst`, one pass (dkt_instance_norm_join_c8).
    `*_params`: (B*C, 2) (mean, 1/std) from extractor.instance_norm_params."""
    B, C, H, W = c.shape
    if not c.is_contiguous() or (a is not None and (not a.is_contiguous() or a.shape != c.shape)) \
            or (y is not None and (not y.is_contiguous() or y.shape != c.shape)):
        raise ValueError("norm_join_c8: dense fp32 NCHW tensors of one shape")
    if dst is not None and (dst.H, dst.W, dst.B) != (H, W, B):
        raise ValueError("norm_join_c8: C8S destination of another size")
    rc = _ffi.lib().dkt_instance_norm_join_c8(
        c.data_ptr(), c_params.data_ptr(), int(bool(c_relu)), None if a is None else a.data_ptr(),
        None if a_params is None else a_params.data_ptr(), int(bool(a_relu)), None if y is None else y.data_ptr(),
        None if dst is None else dst.data_ptr(), 0 if dst is None else dst.bstride_bytes, ch0,
        1.0 if dst is None else dst.scale, B, C, H, W, _ffi.device_of(c), _ffi.stream_of(c))
    _ffi.check(rc, "dkt_instance_norm_join_c8")
    return y if y is not None else dst
