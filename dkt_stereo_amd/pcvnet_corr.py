"""Drop-in replacement for PCVNet's correlation block, ``meta_arch/pcvnet/corr.py:18-61``
(SURVEY.md 8f-3), on the HIP kernels of libdktstereo.so:

    corr_fn = CorrBlock1D(fmap1, fmap2, sample_num=9, num_levels=4, downsample=2)
    corr = corr_fn(coords1, sigma)      # coords, sigma (B,G,H,W) -> (B, L*G*S, H, W) float32

The all-pairs volume is the RAFT one (dkt_corr1d_build); the pyramid is pooled by the
reference's compress factor (4 when ``downsample == 2``, else 2: dkt_pool_rows) and the lookup
samples ``sample_num`` taps spaced by the per-pixel ``sigma`` around each of the G gaussian
means (dkt_pcv_lookup).  Inference only.
"""
import torch

from . import _ffi
from .corr import _build_pyramid


class CorrBlock1D:
    def __init__(self, fmap1, fmap2, sample_num, num_levels=4, downsample=2):
        self.sample_num = sample_num
        self.num_levels = num_levels
        self.compress_factor = 4 if downsample == 2 else 2
        # corr.py:25 -- the reference keeps the tap offsets as a tensor attribute
        half = sample_num // 2
        self.index = torch.arange(-half, half + 1, dtype=torch.float32)
        if self.index.numel() != sample_num:
            raise ValueError("sample_num must be odd (the reference's view(1,1,1,sample_num) needs it)")
        corr = CorrBlock1D.corr(fmap1, fmap2)
        batch, h1, w1, _, w2 = corr.shape
        self._w2 = w2
        lvl = corr.reshape(batch * h1 * w1, 1, 1, w2)
        self.corr_pyramid = [lvl]
        L = _ffi.lib()
        for _ in range(num_levels - 1):
            w = lvl.shape[-1]
            if w // self.compress_factor < 1:
                raise ValueError("width %d cannot be pooled by %d" % (w, self.compress_factor))
            nxt = torch.empty((lvl.shape[0], 1, 1, w // self.compress_factor), device=lvl.device, dtype=torch.float32)
            rc = L.dkt_pool_rows(lvl.data_ptr(), nxt.data_ptr(), lvl.shape[0], w, self.compress_factor,
                                 _ffi.device_of(lvl), _ffi.stream_of(lvl))
            _ffi.check(rc, "dkt_pool_rows")
            self.corr_pyramid.append(nxt)
            lvl = nxt

    def __call__(self, coords, sigma, test_mode=False):
        _ffi.require_gpu(coords, sigma)
        _ffi.require_no_grad(coords, sigma)
        if coords.shape != sigma.shape:
            raise ValueError("coords %s and sigma %s disagree" % (tuple(coords.shape), tuple(sigma.shape)))
        coords = coords.float().contiguous()
        sigma = sigma.float().contiguous()
        B, G, H, W1 = coords.shape
        Lv, S = self.num_levels, self.sample_num
        out = torch.empty((B, Lv * G * S, H, W1), device=coords.device, dtype=torch.float32)
        rc = _ffi.lib().dkt_pcv_lookup(_ffi.ptr_array(self.corr_pyramid), coords.data_ptr(), sigma.data_ptr(),
                                       out.data_ptr(), B, G, H, W1, self._w2, Lv, S, self.compress_factor,
                                       _ffi.device_of(coords), _ffi.stream_of(coords))
        _ffi.check(rc, "dkt_pcv_lookup")
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        B, D, H, W1 = fmap1.shape
        W2 = fmap2.shape[3]
        lvl0, = _build_pyramid(fmap1.float(), fmap2.float(), 1, float(torch.sqrt(torch.tensor(D).float())))
        return lvl0.view(B, H, W1, 1, W2)
