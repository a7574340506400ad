"""Drop-in replacement for ``meta_arch/igev_stereo/geometry.py``
(``Combined_Geo_Encoding_Volume``, :6-69) on the HIP kernels of libdktstereo.

    geo_fn = Combined_Geo_Encoding_Volume(match_left, match_right, geo_volume,
                                          radius=4, num_levels=2)
    feat = geo_fn(disp, coords)        # (B, L*(2r+1)*(C+1), H, W)

Differences a caller cannot observe through ``__call__``: the geometry volume
is kept in the network's native (B,C,D,H,W) layout (the reference's 88 MB
``permute(0,3,4,1,2).reshape`` copy, :18, is not made); pyramid level i is
(B,C,D>>i,H,W) instead of (B*H*W,C,1,D>>i).
"""
import torch

from . import _ffi
from .corr import _build_pyramid


class Combined_Geo_Encoding_Volume:
    def __init__(self, init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        _ffi.require_gpu(init_fmap1, init_fmap2, geo_volume)
        _ffi.require_no_grad(init_fmap1, init_fmap2, geo_volume)
        geo_volume = geo_volume.float().contiguous()
        b, c, d, h, w = geo_volume.shape
        self._shape = (b, c, d, h, w)
        self._w2 = init_fmap2.shape[3]
        # all-pairs correlation WITHOUT the 1/sqrt(C) of RAFT (geometry.py:62-69)
        self.init_corr_pyramid = _build_pyramid(init_fmap1.float(), init_fmap2.float(), num_levels, 1.0)
        self.geo_volume_pyramid = [geo_volume]
        for i in range(1, num_levels):
            src = self.geo_volume_pyramid[-1]
            di = src.shape[2]
            dst = torch.empty((b, c, di // 2, h, w), device=src.device, dtype=torch.float32)
            rc = _ffi.lib().dkt_pool_d(src.data_ptr(), dst.data_ptr(), b * c, di, h * w,
                                       _ffi.device_of(src), _ffi.stream_of(src))
            _ffi.check(rc, "dkt_pool_d")
            self.geo_volume_pyramid.append(dst)

    def __call__(self, disp, coords):
        _ffi.require_gpu(disp, coords)
        b, c, d, h, w = self._shape
        disp = disp.contiguous()
        coords = coords.contiguous()
        K = 2 * self.radius + 1
        out = torch.empty((b, self.num_levels * K * (c + 1), h, w), device=disp.device, dtype=torch.float32)
        rc = _ffi.lib().dkt_geo_lookup(_ffi.ptr_array(self.geo_volume_pyramid),
                                       _ffi.ptr_array(self.init_corr_pyramid),
                                       disp.data_ptr(), coords.data_ptr(), out.data_ptr(),
                                       b, c, d, h, w, self._w2, self.num_levels, self.radius,
                                       _ffi.device_of(disp), _ffi.stream_of(disp))
        _ffi.check(rc, "dkt_geo_lookup")
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        B, D, H, W1 = fmap1.shape
        W2 = fmap2.shape[3]
        lvl0, = _build_pyramid(fmap1.float(), fmap2.float(), 1, 1.0)
        return lvl0.view(B, H, W1, 1, W2)
