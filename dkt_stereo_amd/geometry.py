"""Drop-in replacement for ``meta_arch/igev_stereo/geometry.py``
(``Combined_Geo_Encoding_Volume``, :6-69) on the HIP kernels of libdktstereo.

    geo_fn = Combined_Geo_Encoding_Volume(match_left, match_right, geo_volume,
                                          radius=4, num_levels=2)
    feat = geo_fn(disp, coords)        # (B, L*(2r+1)*(C+1), H, W)

Differences a caller cannot observe through ``__call__``: the geometry volume
is kept in the network's native (B,C,D,H,W) layout (the reference's 88 MB
``permute(0,3,4,1,2).reshape`` copy, :18, is not made); pyramid level i is
(B,C,D>>i,H,W) instead of (B*H*W,C,1,D>>i).

Differentiable w.r.t. ``geo_volume`` and the two feature maps: when they require grad the
pyramids and every lookup are autograd nodes backed by dkt_geo_lookup_bwd / dkt_geo_pool_bwd /
dkt_corr1d_pool_bwd; ``disp`` must be detached, as the reference's loop does.
"""
import torch

from . import _ffi
from .corr import _BuildFn, _WT_LOCK, _build_pyramid


def _pool_geo(geo_volume, num_levels, out=None):
    """(B,C,D,H,W) -> list of num_levels levels, level i = pairwise mean along D of level i-1 (geometry.py:23-25).
    `out`: a previous result of the same shapes whose levels >= 1 are overwritten."""
    b, c, d, h, w = geo_volume.shape
    pyr = [geo_volume]
    for i in range(1, num_levels):
        src = pyr[-1]
        di = src.shape[2]
        dst = out[i] if out is not None else torch.empty((b, c, di // 2, h, w), device=src.device, dtype=torch.float32)
        rc = _ffi.lib().dkt_pool_d(src.data_ptr(), dst.data_ptr(), b * c, di, h * w, _ffi.device_of(src), _ffi.stream_of(src))
        _ffi.check(rc, "dkt_pool_d")
        pyr.append(dst)
    return pyr


class _GeoPyramidFn(torch.autograd.Function):
    """geo_volume -> pyramid levels; backward folds the pooling chain (dkt_geo_pool_bwd)."""

    @staticmethod
    def forward(ctx, geo_volume, num_levels):
        ctx.meta = (tuple(geo_volume.shape), num_levels)
        return tuple(_pool_geo(geo_volume, num_levels))

    @staticmethod
    def backward(ctx, *glv):
        (b, c, d, h, w), L = ctx.meta
        dev = next(g for g in glv if g is not None).device
        glv = [g.contiguous() if g is not None else torch.zeros((b, c, d >> i, h, w), device=dev)
               for i, g in enumerate(glv)]
        out = torch.empty((b, c, d, h, w), device=dev, dtype=torch.float32)
        rc = _ffi.lib().dkt_geo_pool_bwd(_ffi.ptr_array(glv), out.data_ptr(), b * c, d, h * w, L,
                                         _ffi.device_of(out), _ffi.stream_of(out))
        _ffi.check(rc, "dkt_geo_pool_bwd")
        return out, None


class _GeoLookupFn(torch.autograd.Function):
    """(geometry levels, init-correlation levels) -> lookup; backward scatters into zeroed level-shaped tensors
    (dkt_geo_lookup_bwd).  disp / coords carry no gradient (the caller detaches disp, igev_stereo.py:200)."""

    @staticmethod
    def forward(ctx, disp, coords, vol, *levels):
        L = vol.num_levels
        ctx.save_for_backward(disp, coords)
        ctx.vol = vol
        ctx.shapes = [tuple(t.shape) for t in levels]
        return vol._lookup(disp, coords, list(levels[:L]), list(levels[L:]))

    @staticmethod
    def backward(ctx, gout):
        disp, coords = ctx.saved_tensors
        vol = ctx.vol
        L = vol.num_levels
        b, c, d, h, w = vol._shape
        gout = gout.contiguous().float()
        grads = [torch.zeros(s, device=gout.device, dtype=torch.float32) for s in ctx.shapes]
        rc = _ffi.lib().dkt_geo_lookup_bwd(gout.data_ptr(), disp.data_ptr(), coords.data_ptr(),
                                           _ffi.ptr_array(grads[:L]), _ffi.ptr_array(grads[L:]),
                                           b, c, d, h, w, vol._w2, L, vol.radius,
                                           _ffi.device_of(gout), _ffi.stream_of(gout))
        _ffi.check(rc, "dkt_geo_lookup_bwd")
        return (None, None, None) + tuple(grads)


class Combined_Geo_Encoding_Volume:
    def __init__(self, init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        self.init_corr_pyramid = None
        self.geo_volume_pyramid = None
        self._owns_level0 = False
        self.rebuild(init_fmap1, init_fmap2, geo_volume)

    def rebuild(self, init_fmap1, init_fmap2, geo_volume):
        """(Re)computes both pyramids.  With the shapes of the previous build and no autograd involved the
        existing tensors are overwritten in place, so a captured HIP graph of the lookup (igev_loop) keeps
        its pointers -- the counterpart of CorrBlock1D.rebuild."""
        num_levels = self.num_levels
        _ffi.require_gpu(init_fmap1, init_fmap2, geo_volume)
        geo_volume = geo_volume.float().contiguous()
        b, c, d, h, w = geo_volume.shape
        shape = (b, c, d, h, w)
        w2 = init_fmap2.shape[3]
        grad = torch.is_grad_enabled()
        f_grad = grad and (init_fmap1.requires_grad or init_fmap2.requires_grad)
        g_grad = grad and geo_volume.requires_grad
        reuse = (self.geo_volume_pyramid is not None and not f_grad and not g_grad
                 and shape == getattr(self, "_shape", None) and w2 == getattr(self, "_w2", None)
                 and not any(t.requires_grad for t in self.geo_volume_pyramid + self.init_corr_pyramid))
        self._shape = shape
        self._w2 = w2
        # all-pairs correlation WITHOUT the 1/sqrt(C) of RAFT (geometry.py:62-69)
        if f_grad:
            self.init_corr_pyramid = list(_BuildFn.apply(init_fmap1.float(), init_fmap2.float(), num_levels, 1.0))
        else:
            self.init_corr_pyramid = _build_pyramid(init_fmap1.float(), init_fmap2.float(), num_levels, 1.0,
                                                    out=self.init_corr_pyramid if reuse else None)
        if g_grad:
            self.geo_volume_pyramid = list(_GeoPyramidFn.apply(geo_volume, num_levels))
            self._owns_level0 = False
        elif reuse:
            self.own_buffers()
            self.geo_volume_pyramid[0].copy_(geo_volume)
            _pool_geo(self.geo_volume_pyramid[0], num_levels, out=self.geo_volume_pyramid)
        else:
            self.geo_volume_pyramid = _pool_geo(geo_volume, num_levels)
            self._owns_level0 = False

    def own_buffers(self):
        """Level 0 of the geometry pyramid is the caller's tensor after a fresh build; this replaces it
        by a private copy (once), so that later in-place refills never write into the caller's tensor.
        Call it BEFORE capturing a graph of the lookup: it changes the level-0 pointer."""
        if not self._owns_level0:
            self.geo_volume_pyramid[0] = self.geo_volume_pyramid[0].detach().clone()
            self._owns_level0 = True

    def copy_from(self, other):
        """Takes over the pyramids of another volume of the same shapes by device copies into the
        existing tensors (pointers stay valid for a captured graph)."""
        if (other._shape, other._w2, other.num_levels, other.radius) != (self._shape, self._w2, self.num_levels, self.radius):
            raise ValueError("Combined_Geo_Encoding_Volume.copy_from: shapes differ")
        self.own_buffers()
        for dst, src in zip(self.geo_volume_pyramid + self.init_corr_pyramid,
                            other.geo_volume_pyramid + other.init_corr_pyramid):
            dst.copy_(src.detach())

    def _lookup(self, disp, coords, geo_pyr, init_pyr):
        b, c, d, h, w = self._shape
        K = 2 * self.radius + 1
        out = torch.empty((b, self.num_levels * K * (c + 1), h, w), device=disp.device, dtype=torch.float32)
        rc = _ffi.lib().dkt_geo_lookup(_ffi.ptr_array(geo_pyr), _ffi.ptr_array(init_pyr),
                                       disp.data_ptr(), coords.data_ptr(), out.data_ptr(),
                                       b, c, d, h, w, self._w2, self.num_levels, self.radius,
                                       _ffi.device_of(disp), _ffi.stream_of(disp))
        _ffi.check(rc, "dkt_geo_lookup")
        return out

    def __call__(self, disp, coords):
        _ffi.require_gpu(disp, coords)
        disp = disp.float().contiguous()
        coords = coords.float().contiguous()
        levels = self.geo_volume_pyramid + self.init_corr_pyramid
        if torch.is_grad_enabled() and any(t.requires_grad for t in levels):
            if disp.requires_grad or coords.requires_grad:
                raise _ffi.DktError("Combined_Geo_Encoding_Volume: disparity gradients are not implemented; detach "
                                    "disp as the reference's loop does (igev_stereo.py:200)")
            return _GeoLookupFn.apply(disp, coords, self, *levels)
        return self._lookup(disp, coords, self.geo_volume_pyramid, self.init_corr_pyramid)

    def lookup_conv1x1(self, disp, coords, layer, relu=True, tap=False, out_c8=None, out_c8_ch0=0):
        """relu(layer(self(disp, coords))) for a 1x1 ``layer`` with <= 64 outputs (the motion encoder's convc1,
        igev_stereo/update.py:78,86) without ever writing the L*(2r+1)*(C+1)-channel lookup (dkt_geo_lookup_conv1x1).
        Returns None when the fused kernel does not cover the configuration (anything but 2 levels, 8 geometry channels,
        radius 4; autograd involved): run the two steps separately then.  ``out_c8``: write the C8S operand of the next
        convolution (conv_c8) instead of fp32 NCHW.  tap=True additionally returns the sampled values (bit-identical to
        ``self(disp, coords)``)."""
        w = layer.weight
        b, c, d, h, wd = self._shape
        K = 2 * self.radius + 1
        levels = self.geo_volume_pyramid + self.init_corr_pyramid
        if (w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[0] > 64 or w.shape[1] != self.num_levels * K * (c + 1)
                or self.num_levels != 2 or c != 8 or self.radius != 4 or (d >> 1) == 0 or (self._w2 >> 1) == 0
                or getattr(layer, "groups", 1) != 1 or tuple(getattr(layer, "stride", (1, 1))) != (1, 1)
                or (torch.is_grad_enabled() and (w.requires_grad or disp.requires_grad or any(t.requires_grad for t in levels)))):
            return None
        _ffi.require_gpu(disp, coords)
        if disp.dtype != torch.float32 or disp.stride(3) != 1 or disp.stride(2) != wd:
            disp = disp.float().contiguous()
        coords = coords.float().contiguous()
        cout = w.shape[0]
        key = (w.data_ptr(), w._version)
        with _WT_LOCK:
            cache = layer.__dict__.setdefault("_dkt_wt", {})
            hit = cache.get(str(w.device))
            if hit is None or hit[0] != key:
                hit = cache[str(w.device)] = (key, w.detach().reshape(cout, -1).t().float().contiguous())
        wm = hit[1]
        bias = layer.bias
        out = None if out_c8 is not None else torch.empty((b, cout, h, wd), device=disp.device, dtype=torch.float32)
        tp = torch.empty((b, w.shape[1], h, wd), device=disp.device, dtype=torch.float32) if tap else None
        rc = _ffi.lib().dkt_geo_lookup_conv1x1(
            _ffi.ptr_array(self.geo_volume_pyramid), _ffi.ptr_array(self.init_corr_pyramid),
            disp.data_ptr(), disp.stride(0), coords.data_ptr(), wm.data_ptr(),
            None if bias is None else bias.detach().data_ptr(),
            None if out is None else out.data_ptr(), 0 if out is None else out.stride(0),
            None if out_c8 is None else out_c8.data_ptr(), 0 if out_c8 is None else out_c8.bstride_bytes, out_c8_ch0,
            1.0 if out_c8 is None else out_c8.scale, None if tp is None else tp.data_ptr(), 0 if tp is None else tp.stride(0),
            b, c, d, h, wd, self._w2, self.num_levels, self.radius, cout, int(bool(relu)),
            _ffi.device_of(disp), _ffi.stream_of(disp))
        _ffi.check(rc, "dkt_geo_lookup_conv1x1")
        res = out if out_c8 is None else out_c8
        return (res, tp) if tap else res

    @staticmethod
    def corr(fmap1, fmap2):
        B, D, H, W1 = fmap1.shape
        W2 = fmap2.shape[3]
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            lvl0, = _BuildFn.apply(fmap1.float(), fmap2.float(), 1, 1.0)
        else:
            lvl0, = _build_pyramid(fmap1.float(), fmap2.float(), 1, 1.0)
        return lvl0.view(B, H, W1, 1, W2)
