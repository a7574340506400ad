"""Drop-in replacements for the correlation classes of the reference's
``core/corr.py`` (identical copy: ``meta_arch/raft_stereo/corr.py``), backed by
the HIP kernels of libdktstereo.so through ``_ffi``.

Same constructor / call signatures and attributes as the reference, so
``RAFTStereo.forward`` (meta_arch/raft_stereo/raft_stereo.py:118-154) can use
them unchanged:

    corr_fn = CorrBlock1D(fmap1, fmap2, radius=4, num_levels=4)
    corr = corr_fn(coords1)            # (B, L*(2r+1), H, W) float32 contiguous

``CorrBlock1D`` is differentiable w.r.t. the two feature maps: when they
require grad the pyramid and every lookup are autograd nodes whose backward runs
dkt_corr1d_lookup_bwd / dkt_corr1d_pool_bwd and two library GEMMs -- what autograd derives for
the reference's grid_sample / avg_pool2d / einsum chain.  Coordinates must be detached, as the
reference's callers do (raft_stereo.py:152).  The other classes are inference only.
"""
import os
import threading

import torch

from . import _ffi

_WT_LOCK = threading.Lock()


def _build_pyramid(fmap1, fmap2, num_levels, divisor, out=None):
    """dkt_corr1d_build: all-pairs correlation + avg-pool pyramid in one launch.
    Returns L tensors shaped like the reference's pyramid entries (N,1,1,W2_i);
    `out` (a previous result of the same shape) is overwritten in place if given."""
    _ffi.require_gpu(fmap1, fmap2)
    _ffi.require_no_grad(fmap1, fmap2)
    fmap1 = fmap1.contiguous()
    fmap2 = fmap2.contiguous()
    B, C, H, W1 = fmap1.shape
    B2, C2, H2, W2 = fmap2.shape
    if (B, C, H) != (B2, C2, H2):
        raise ValueError("fmap1 %s and fmap2 %s disagree" % (tuple(fmap1.shape), tuple(fmap2.shape)))
    N = B * H * W1
    if out is not None:
        pyr = out
        assert len(pyr) == num_levels and all(p.shape == (N, 1, 1, W2 >> i) for i, p in enumerate(pyr))
    else:
        pyr = [torch.empty((N, 1, 1, W2 >> i), device=fmap1.device, dtype=torch.float32)
               for i in range(num_levels)]
    rc = _ffi.lib().dkt_corr1d_build(fmap1.data_ptr(), fmap2.data_ptr(), _ffi.ptr_array(pyr),
                                     B, C, H, W1, W2, num_levels, float(divisor),
                                     _ffi.device_of(fmap1), _ffi.stream_of(fmap1))
    _ffi.check(rc, "dkt_corr1d_build")
    return pyr


def _lookup(pyramid, coords, radius, w2, skewed=False):
    _ffi.require_gpu(coords)
    B, _, H, W1 = coords.shape
    if coords.stride(3) != 1 or coords.stride(2) != W1:
        coords = coords.contiguous()
    L = len(pyramid)
    K = 2 * radius + 1
    out = torch.empty((B, L * K, H, W1), device=coords.device, dtype=torch.float32)
    # only channel 0 (x) is read, like coords[:, :1] in core/corr.py:129
    fn = _ffi.lib().dkt_corr1d_lookup_skew if skewed else _ffi.lib().dkt_corr1d_lookup
    rc = fn(_ffi.ptr_array(pyramid), coords.data_ptr(), coords.stride(0), out.data_ptr(), B, H, W1, w2, L,
            radius, _ffi.device_of(coords), _ffi.stream_of(coords))
    _ffi.check(rc, "dkt_corr1d_lookup_skew" if skewed else "dkt_corr1d_lookup")
    return out


def _skew_pyramid(pyramid, B, H, W1, W2, out=None):
    """dkt_corr1d_skew: diagonal-major copy of every level (see corr1d_skew.hip)."""
    if out is None:
        pitch = _ffi.lib().dkt_corr1d_skew_pitch(W1)
        out = [torch.empty((B * H, W2 >> i, pitch), device=p.device, dtype=torch.float32)
               for i, p in enumerate(pyramid)]
    rc = _ffi.lib().dkt_corr1d_skew(_ffi.ptr_array(pyramid), _ffi.ptr_array(out), B, H, W1, W2, len(pyramid),
                                    _ffi.device_of(pyramid[0]), _ffi.stream_of(pyramid[0]))
    _ffi.check(rc, "dkt_corr1d_skew")
    return out


class _BuildFn(torch.autograd.Function):
    """(fmap1, fmap2) -> pyramid levels.  Backward: avg-pool chain + 1/sqrt(C) folded by
    dkt_corr1d_pool_bwd, then grad_fmap1 = G x fmap2, grad_fmap2 = G^T x fmap1 (rocBLAS)."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, num_levels, divisor):
        pyr = _build_pyramid(fmap1, fmap2, num_levels, divisor)
        ctx.save_for_backward(fmap1, fmap2)
        ctx.meta = (num_levels, divisor)
        return tuple(pyr)

    @staticmethod
    def backward(ctx, *glv):
        f1, f2 = ctx.saved_tensors
        L, divisor = ctx.meta
        B, C, H, W1 = f1.shape
        W2 = f2.shape[3]
        N = B * H * W1
        glv = [g.contiguous() if g is not None else torch.zeros((N, 1, 1, W2 >> i), device=f1.device)
               for i, g in enumerate(glv)]
        gvol = torch.empty((N, W2), device=f1.device, dtype=torch.float32)
        rc = _ffi.lib().dkt_corr1d_pool_bwd(_ffi.ptr_array(glv), gvol.data_ptr(), B, H, W1, W2, L, float(divisor),
                                            _ffi.device_of(gvol), _ffi.stream_of(gvol))
        _ffi.check(rc, "dkt_corr1d_pool_bwd")
        G = gvol.view(B, H, W1, W2)
        gf1 = torch.einsum('bhwv,bchv->bchw', G, f2.float()) if ctx.needs_input_grad[0] else None
        gf2 = torch.einsum('bhwv,bchw->bchv', G, f1.float()) if ctx.needs_input_grad[1] else None
        return gf1, gf2, None, None


class _LookupFn(torch.autograd.Function):
    """(levels..., coords) -> lookup.  Backward scatters into zeroed level-shaped tensors."""

    @staticmethod
    def forward(ctx, coords, radius, w2, skew, *levels):
        ctx.save_for_backward(coords)
        ctx.meta = (radius, w2, [tuple(l.shape) for l in levels])
        if skew is not None:
            return _lookup(skew, coords, radius, w2, skewed=True)
        return _lookup(list(levels), coords, radius, w2)

    @staticmethod
    def backward(ctx, gout):
        coords, = ctx.saved_tensors
        radius, w2, shapes = ctx.meta
        B, _, H, W1 = coords.shape
        if coords.stride(3) != 1 or coords.stride(2) != W1:
            coords = coords.contiguous()
        gout = gout.contiguous().float()
        grads = [torch.zeros(s, device=gout.device, dtype=torch.float32) for s in shapes]
        rc = _ffi.lib().dkt_corr1d_lookup_bwd(gout.data_ptr(), coords.data_ptr(), coords.stride(0),
                                              _ffi.ptr_array(grads), B, H, W1, w2, len(grads), radius,
                                              _ffi.device_of(gout), _ffi.stream_of(gout))
        _ffi.check(rc, "dkt_corr1d_lookup_bwd")
        return (None, None, None, None) + tuple(grads)


class DeferredLookup:
    """``corr_fn.deferred(coords)``: stands in for the lookup tensor between the loop harness and the
    motion encoder (which is its only consumer, core/update.py:79)."""

    def __init__(self, block, coords):
        self.block, self.coords = block, coords
        self.device = coords.device

    @property
    def is_cuda(self):
        return self.coords.is_cuda

    def materialize(self):
        return self.block(self.coords)

    def conv1x1(self, layer, relu=True):
        out = self.block.lookup_conv1x1(self.coords, layer, relu=relu)
        return out


class CorrBlock1D:
    """core/corr.py:110-156.  ``corr_pyramid`` holds the ``num_levels`` levels that
    ``__call__`` reads (the reference also stores one more pooled level that
    nothing ever reads)."""

    #: "skew": lookups read a diagonal-major copy of the pyramid (neighbouring pixels read
    #: neighbouring floats when disparity is locally smooth); "rows": the reference layout.
    #: Both give bit-identical results.  ``corr_pyramid`` is always the reference layout.
    lookup_layout = "skew"

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        self._w2 = fmap2.shape[3]
        self.corr_pyramid = None
        self._skew = None
        self.rebuild(fmap1, fmap2)

    def rebuild(self, fmap1, fmap2):
        """Recomputes the pyramid for a new pair INTO the existing tensors (same shapes):
        lets a captured HIP graph of the lookup keep its pointers."""
        B, C, H, W1 = fmap1.shape
        # corr / sqrt(C) (core/corr.py:156); the kernel divides like the reference
        divisor = float(torch.sqrt(torch.tensor(C).float()))
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            self.corr_pyramid = list(_BuildFn.apply(fmap1.float(), fmap2.float(), self.num_levels, divisor))
        else:
            self.corr_pyramid = _build_pyramid(fmap1.float(), fmap2.float(), self.num_levels, divisor,
                                               out=self.corr_pyramid)
        if self.lookup_layout == "skew":
            self._skew = _skew_pyramid(self.corr_pyramid, B, H, W1, self._w2, out=self._skew)

    def __call__(self, coords):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.corr_pyramid):
            if coords.requires_grad:
                raise _ffi.DktError("CorrBlock1D: coordinate gradients are not implemented; detach the "
                                    "coordinates as the reference's callers do (raft_stereo.py:152)")
            return _LookupFn.apply(coords, self.radius, self._w2, self._skew, *self.corr_pyramid)
        if self._skew is not None:
            return _lookup(self._skew, coords, self.radius, self._w2, skewed=True)
        return _lookup(self.corr_pyramid, coords, self.radius, self._w2)

    # ---- lookup fused with the 1x1 convolution that consumes it (dkt_corr1d_lookup_conv1x1) ----
    def deferred(self, coords):
        """A lookup that has not been run yet: ``BasicMotionEncoder`` turns it into
        relu(convc1(lookup)) in ONE launch (``DeferredLookup.conv1x1``); anything else that needs the
        tensor calls ``materialize()`` and gets exactly ``self(coords)``."""
        return DeferredLookup(self, coords)

    def lookup_conv1x1(self, coords, layer, relu=True, tap=False, out_c8=None, out_c8_ch0=0):
        """relu(layer(self(coords))) for a 1x1 ``layer`` with <= 64 outputs, without ever writing the
        lookup.  Returns None when the fused kernel does not cover this configuration (row layout,
        unsupported L / r / layer, autograd involved): run the two steps separately then.
        tap=True additionally returns the sampled values (bit-identical to ``self(coords)``)."""
        w = layer.weight
        K = 2 * self.radius + 1
        if (self._skew is None or w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[0] > 64
                or w.shape[1] != self.num_levels * K or self.num_levels not in (2, 3, 4) or self.radius not in (3, 4)
                or (self._w2 >> (self.num_levels - 1)) < 2
                or getattr(layer, "groups", 1) != 1 or tuple(getattr(layer, "stride", (1, 1))) != (1, 1)
                or (torch.is_grad_enabled() and (w.requires_grad or coords.requires_grad
                                                 or any(p.requires_grad for p in self.corr_pyramid)))):
            return None
        _ffi.require_gpu(coords)
        B, _, H, W1 = coords.shape
        if coords.stride(3) != 1 or coords.stride(2) != W1:
            coords = coords.contiguous()
        cout = w.shape[0]
        wm = _kmajor_weight(layer)
        bias = layer.bias
        if out_c8 is not None:
            # straight into the C8S operand of the next convolution (conv_c8.hip); four levels only
            if self.num_levels != 4 or tap:
                return None
            rc = _ffi.lib().dkt_corr1d_lookup_conv1x1_c8(
                _ffi.ptr_array(self._skew), coords.data_ptr(), coords.stride(0), wm.data_ptr(),
                None if bias is None else bias.detach().data_ptr(), out_c8.data_ptr(), out_c8.bstride_bytes, out_c8_ch0,
                out_c8.scale, B, H, W1, self._w2, self.num_levels, self.radius, cout, int(bool(relu)),
                _ffi.device_of(coords), _ffi.stream_of(coords))
            _ffi.check(rc, "dkt_corr1d_lookup_conv1x1_c8")
            return out_c8
        out = torch.empty((B, cout, H, W1), device=coords.device, dtype=torch.float32)
        tp = torch.empty((B, self.num_levels * K, H, W1), device=coords.device, dtype=torch.float32) if tap else None
        rc = _ffi.lib().dkt_corr1d_lookup_conv1x1(
            _ffi.ptr_array(self._skew), coords.data_ptr(), coords.stride(0), wm.data_ptr(),
            None if bias is None else bias.detach().data_ptr(), out.data_ptr(), out.stride(0),
            None if tp is None else tp.data_ptr(), 0 if tp is None else tp.stride(0),
            B, H, W1, self._w2, self.num_levels, self.radius, cout, int(bool(relu)),
            _ffi.device_of(coords), _ffi.stream_of(coords))
        _ffi.check(rc, "dkt_corr1d_lookup_conv1x1")
        return (out, tp) if tap else out

    @staticmethod
    def corr(fmap1, fmap2):
        B, D, H, W1 = fmap1.shape
        W2 = fmap2.shape[3]
        divisor = float(torch.sqrt(torch.tensor(D).float()))
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            lvl0, = _BuildFn.apply(fmap1.float(), fmap2.float(), 1, divisor)
        else:
            lvl0, = _build_pyramid(fmap1.float(), fmap2.float(), 1, divisor)
        return lvl0.view(B, H, W1, 1, W2)


def _kmajor_weight(layer):
    """The 1x1 layer's weight k-major, (L*K, Cout), as the fused lookup kernels read it; cached on the layer per device and
    version (under the lock: a thread that lost a creation race would otherwise free the tensor another thread's captured
    graph already points to)."""
    w = layer.weight
    key = (w.data_ptr(), w._version)
    with _WT_LOCK:
        cache = layer.__dict__.setdefault("_dkt_wt", {})
        hit = cache.get(str(w.device))
        if hit is None or hit[0] != key:
            hit = cache[str(w.device)] = (key, w.detach().reshape(w.shape[0], -1).t().float().contiguous())
    return hit[1]


class CorrBlockFast1D(CorrBlock1D):
    """core/corr.py:31-61 ("reg_cuda").  The reference needs the un-vendored
    ``corr_sampler`` CUDA extension for this class; here it is the same HIP
    lookup as CorrBlock1D (same skewed copy, same autograd nodes -- the reference class is
    differentiable through CorrSampler, :17-29), with the reference's 5-D pyramid views
    in ``corr_pyramid``."""

    def rebuild(self, fmap1, fmap2):
        self.corr_pyramid = getattr(self, "_flat", None)
        super().rebuild(fmap1, fmap2)
        B, _, H, W1 = fmap1.shape
        self._flat = self.corr_pyramid
        self.corr_pyramid = [p.view(B, H, W1, -1, p.shape[-1]) for p in self._flat]

    def __call__(self, coords):
        views, self.corr_pyramid = self.corr_pyramid, self._flat
        try:
            return super().__call__(coords)
        finally:
            self.corr_pyramid = views


class CorrBlock1D_Cosine(CorrBlock1D):
    """core/corr.py:160-209: L2-normalised features, no 1/sqrt(C) scaling.
    (The ``mix=`` training-time blend of meta_arch/raft_stereo/corr.py:216-228 is
    not part of the inference path.)"""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        self._w2 = fmap2.shape[3]
        self.corr_pyramid = _build_pyramid(self._normalise(fmap1), self._normalise(fmap2), num_levels, 1.0)
        B, _, H, W1 = fmap1.shape
        self._skew = (_skew_pyramid(self.corr_pyramid, B, H, W1, self._w2)
                      if self.lookup_layout == "skew" else None)

    @staticmethod
    def _normalise(fmap):
        _ffi.require_no_grad(fmap)           # inference only: never detach silently
        fmap = fmap.float().contiguous()
        _ffi.require_gpu(fmap)
        B, C, H, W = fmap.shape
        out = torch.empty_like(fmap)
        rc = _ffi.lib().dkt_l2norm_channels(fmap.data_ptr(), out.data_ptr(), B, C, H * W,
                                            _ffi.device_of(fmap), _ffi.stream_of(fmap))
        _ffi.check(rc, "dkt_l2norm_channels")
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        B, D, H, W1 = fmap1.shape
        W2 = fmap2.shape[3]
        lvl0, = _build_pyramid(CorrBlock1D_Cosine._normalise(fmap1),
                               CorrBlock1D_Cosine._normalise(fmap2), 1, 1.0)
        return lvl0.view(B, H, W1, 1, W2)


class PytorchAlternateCorrBlock1D:
    """core/corr.py:64-107 ("alt"): no correlation volume, O(H*W*C) memory; every
    call recomputes the 2r+1 correlations per level from the feature maps."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = []
        _ffi.require_gpu(fmap1, fmap2)
        _ffi.require_no_grad(fmap1, fmap2)
        self.fmap1 = fmap1.float().contiguous()
        self.fmap2 = fmap2.float().contiguous()
        # right feature map pooled along W once per level (corr.py:104 does it per call)
        self._f2pyr = [self.fmap2]
        B, C, H, W2 = self.fmap2.shape
        for i in range(1, num_levels):
            src = self._f2pyr[-1]
            dst = torch.empty((B, C, H, src.shape[3] // 2), device=src.device, dtype=torch.float32)
            rc = _ffi.lib().dkt_pool_w(src.data_ptr(), dst.data_ptr(), B * C * H, src.shape[3],
                                       _ffi.device_of(src), _ffi.stream_of(src))
            _ffi.check(rc, "dkt_pool_w")
            self._f2pyr.append(dst)

    def __call__(self, coords):
        _ffi.require_gpu(coords)
        coords = coords.contiguous()
        B, C, H, W1 = self.fmap1.shape
        W2 = self.fmap2.shape[3]
        K = 2 * self.radius + 1
        out = torch.empty((B, self.num_levels * K, H, W1), device=coords.device, dtype=torch.float32)
        rc = _ffi.lib().dkt_corr1d_lookup_otf(self.fmap1.data_ptr(), _ffi.ptr_array(self._f2pyr),
                                              coords.data_ptr(), out.data_ptr(), B, C, H, W1, W2,
                                              self.num_levels, self.radius,
                                              _ffi.device_of(coords), _ffi.stream_of(coords))
        _ffi.check(rc, "dkt_corr1d_lookup_otf")
        return out


class AlternateCorrBlock:
    """core/corr.py:212-241: disabled upstream (raises before doing anything)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        raise NotImplementedError


#: value of ``args.corr_implementation`` -> class, as selected in
#: meta_arch/raft_stereo/raft_stereo.py:118-132
CORR_IMPLEMENTATIONS = {
    "reg": CorrBlock1D,
    "alt": PytorchAlternateCorrBlock1D,
    "reg_cuda": CorrBlockFast1D,
    "alt_cuda": AlternateCorrBlock,
    "cosine": CorrBlock1D_Cosine,
    # raft_stereo.py:133-136: in test_mode "mix_fmap_image" IS the cosine block on the feature maps (the
    # image-mixing branch, :137-142, exists only in training)
    "mix_fmap_image": CorrBlock1D_Cosine,
}
