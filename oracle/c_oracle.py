"""TEST INFRASTRUCTURE -- numpy front-end to oracle/dkt_oracle.c (the plain-C
restatement of the hot path's kernels).  See the header of dkt_oracle.c for
what each entry restates (reference file:line) and how it is pinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdkt_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_lib = None


def build(force=False):
    """Compile dkt_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "dkt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_bilinear_1d.restype = ctypes.c_float
        _lib.orc_bilinear_1d.argtypes = [_f32p, ctypes.c_int, ctypes.c_float]
    return _lib


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(_f32p)


def _pp(arrs):
    return (_f32p * len(arrs))(*[_p(a) for a in arrs])


def bilinear_1d(row, x):
    row = _c(row)
    return float(lib().orc_bilinear_1d(_p(row), row.shape[0], ctypes.c_float(x)))


def corr1d_build(f1, f2, num_levels, divide_by_sqrt_c=True):
    """-> list of L arrays (B*H*W1, W2>>i).  core/corr.py:111-125,148-156."""
    f1, f2 = _c(f1), _c(f2)
    B, C, H, W1 = f1.shape
    W2 = f2.shape[3]
    pyr = [np.empty((B * H * W1, W2 >> i), np.float32) for i in range(num_levels)]
    s = float(np.sqrt(np.float32(C))) if divide_by_sqrt_c else 0.0
    lib().orc_corr1d_build(_p(f1), _p(f2), _pp(pyr), B, C, H, W1, W2, num_levels, ctypes.c_float(s))
    return pyr


def pool_pyramid(lvl0, num_levels):
    lvl0 = _c(lvl0)
    N, W2 = lvl0.shape
    outs = [np.empty((N, W2 >> i), np.float32) for i in range(1, num_levels)]
    if outs:
        lib().orc_pool_pyramid(_p(lvl0), _pp(outs), ctypes.c_size_t(N), W2, num_levels)
    return [lvl0] + outs


def corr1d_lookup(pyr, coords, radius):
    """coords (B,2,H,W) -> (B, L*K, H, W).  core/corr.py:127-146."""
    pyr = [_c(p) for p in pyr]
    coords = _c(coords)
    B, _, H, W1 = coords.shape
    L = len(pyr)
    W2 = pyr[0].shape[1]
    K = 2 * radius + 1
    out = np.empty((B, L * K, H, W1), np.float32)
    lib().orc_corr1d_lookup(_pp(pyr), _p(coords), _p(out), B, H, W1, W2, L, radius)
    return out


def corr1d_lookup_alt(f1, f2, coords, num_levels, radius):
    """On-the-fly variant.  core/corr.py:64-107."""
    f1, f2, coords = _c(f1), _c(f2), _c(coords)
    B, C, H, W1 = f1.shape
    W2 = f2.shape[3]
    f2pyr = [f2]
    for i in range(1, num_levels):
        prev = f2pyr[-1]
        w = prev.shape[3]
        nxt = np.empty((B, C, H, w // 2), np.float32)
        lib().orc_pool_rows(_p(prev), _p(nxt), ctypes.c_size_t(B * C * H), w)
        f2pyr.append(nxt)
    K = 2 * radius + 1
    out = np.empty((B, num_levels * K, H, W1), np.float32)
    lib().orc_corr1d_lookup_alt(_p(f1), _pp(f2pyr), _p(coords), _p(out), B, C, H, W1, W2, num_levels, radius)
    return out


def geo_pyramids(fmap1, fmap2, geo_volume, num_levels):
    """Reference-layout pyramids.  meta_arch/igev_stereo/geometry.py:7-29."""
    geo_volume = _c(geo_volume)
    B, C, D, H, W = geo_volume.shape
    g0 = np.empty((B * H * W * C, D), np.float32)
    lib().orc_geo_permute(_p(geo_volume), _p(g0), B, C, D, H, W)
    geo_pyr = pool_pyramid(g0, num_levels)
    init_pyr = corr1d_build(fmap1, fmap2, num_levels, divide_by_sqrt_c=False)
    return geo_pyr, init_pyr


def geo_lookup(geo_pyr, init_pyr, disp, coords, C, radius):
    """disp (B,1,H,W), coords (B,H,W,1) -> (B, L*K*(C+1), H, W).  geometry.py:34-58."""
    geo_pyr = [_c(p) for p in geo_pyr]
    init_pyr = [_c(p) for p in init_pyr]
    disp, coords = _c(disp), _c(coords)
    B, _, H, W = disp.shape
    L = len(geo_pyr)
    D = geo_pyr[0].shape[1]
    W2 = init_pyr[0].shape[1]
    K = 2 * radius + 1
    out = np.empty((B, L * K * (C + 1), H, W), np.float32)
    lib().orc_geo_lookup(_pp(geo_pyr), _pp(init_pyr), _p(disp), _p(coords), _p(out),
                         B, C, D, H, W, W2, L, radius)
    return out


def gwc_volume(ref, tgt, maxdisp, num_groups):
    ref, tgt = _c(ref), _c(tgt)
    B, C, H, W = ref.shape
    vol = np.empty((B, num_groups, maxdisp, H, W), np.float32)
    lib().orc_gwc_volume(_p(ref), _p(tgt), _p(vol), B, C, H, W, maxdisp, num_groups)
    return vol


def concat_volume(ref, tgt, maxdisp, ref_masked):
    ref, tgt = _c(ref), _c(tgt)
    B, C, H, W = ref.shape
    vol = np.empty((B, 2 * C, maxdisp, H, W), np.float32)
    lib().orc_concat_volume(_p(ref), _p(tgt), _p(vol), B, C, H, W, maxdisp, int(ref_masked))
    return vol


def gru_gate_zr(az, ar, cz, cr, h):
    az, ar, cz, cr, h = map(_c, (az, ar, cz, cr, h))
    z = np.empty_like(h)
    rh = np.empty_like(h)
    lib().orc_gru_gate_zr(_p(az), _p(ar), _p(cz), _p(cr), _p(h), _p(z), _p(rh), ctypes.c_size_t(h.size))
    return z, rh


def gru_gate_out(aq, cq, z, h):
    aq, cq, z, h = map(_c, (aq, cq, z, h))
    out = np.empty_like(h)
    lib().orc_gru_gate_out(_p(aq), _p(cq), _p(z), _p(h), _p(out), ctypes.c_size_t(h.size))
    return out


def conv2d_same(x, w, bias):
    x, w = _c(x), _c(w)
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    y = np.empty((B, Cout, H, W), np.float32)
    bp = _p(_c(bias)) if bias is not None else None
    lib().orc_conv2d_same(_p(x), _p(w), bp, _p(y), B, Cin, H, W, Cout, KH, KW)
    return y


def pcv_pyramid(lvl0, num_levels, factor):
    """Rows pooled by `factor` (meta_arch/pcvnet/corr.py:27-31): list of num_levels (N, W_i)."""
    out = [_c(lvl0)]
    for _ in range(num_levels - 1):
        src = out[-1]
        dst = np.empty((src.shape[0], src.shape[1] // factor), np.float32)
        lib().orc_pool_rows_f(_p(src), _p(dst), ctypes.c_size_t(src.shape[0]), src.shape[1], factor)
        out.append(dst)
    return out


def pcv_lookup(pyr, coords, sigma, sample_num, factor):
    """meta_arch/pcvnet/corr.py:33-51 on a given pyramid; coords, sigma: (B,G,H,W1)."""
    coords, sigma = _c(coords), _c(sigma)
    B, G, H, W1 = coords.shape
    pyr = [_c(p) for p in pyr]
    L = len(pyr)
    out = np.empty((B, L * G * sample_num, H, W1), np.float32)
    lib().orc_pcv_lookup(_pp(pyr), _p(coords), _p(sigma), _p(out), B, G, H, W1, pyr[0].shape[1], L, sample_num, factor)
    return out


def group_l2norm(x, num_groups):
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty_like(x)
    lib().orc_group_l2norm(_p(x), _p(y), B, C, H, W, num_groups)
    return y


def gwc_volume_norm(ref, tgt, maxdisp, num_groups):
    """cgi/submodule.py:143-164 (num_groups=1: build_norm_correlation_volume, :167-180)."""
    return gwc_volume(group_l2norm(ref, num_groups), group_l2norm(tgt, num_groups), maxdisp, num_groups)


def corr1d_lookup_bwd(gout, coords, radius, widths, rows):
    """Gradient of every pyramid level (zeros + scatter); widths: list of W2_i, rows = B*H*W1."""
    gout, coords = _c(gout), _c(coords)
    B, _, H, W1 = coords.shape
    g = [np.zeros((rows, w), np.float32) for w in widths]
    lib().orc_corr1d_lookup_bwd(_p(gout), _p(coords), _pp(g), B, H, W1, widths[0], len(widths), radius)
    return g


def corr1d_pool_bwd(gpyr, divisor):
    gpyr = [_c(g) for g in gpyr]
    g0 = np.empty_like(gpyr[0])
    lib().orc_corr1d_pool_bwd(_pp(gpyr), _p(g0), ctypes.c_size_t(g0.shape[0]), g0.shape[1], len(gpyr), ctypes.c_float(divisor))
    return g0


def corr1d_build_bwd(g0_total, f1, f2):
    """d/d fmap of sum_c f1*f2 given the (already /sqrt(C)) volume gradient: two contractions in fp64."""
    f1, f2 = np.asarray(f1, np.float64), np.asarray(f2, np.float64)
    B, C, H, W1 = f1.shape
    W2 = f2.shape[3]
    G = np.asarray(g0_total, np.float64).reshape(B, H, W1, W2)
    gf1 = np.einsum('bhwv,bchv->bchw', G, f2)
    gf2 = np.einsum('bhwv,bchw->bchv', G, f1)
    return gf1.astype(np.float32), gf2.astype(np.float32)


def convex_upsample(flow, mask, factor):
    flow, mask = _c(flow), _c(mask)
    N, D, H, W = flow.shape
    out = np.empty((N, D, factor * H, factor * W), np.float32)
    lib().orc_convex_upsample(_p(flow), _p(mask), _p(out), N, D, H, W, factor)
    return out


def context_upsample(disp_low, up_weights):
    disp_low, up_weights = _c(disp_low), _c(up_weights)
    B, _, h, w = disp_low.shape
    out = np.empty((B, 4 * h, 4 * w), np.float32)
    lib().orc_context_upsample(_p(disp_low), _p(up_weights), _p(out), B, h, w)
    return out


def geo_lookup_bwd(gout, disp, coords, C, D, W2, num_levels, radius):
    """Per-level gradients of the geometry pyramid (reference layout rows (n*C+c, D>>i)) and of the
    init-correlation pyramid (rows (n, W2>>i)) from the gradient of one lookup's output."""
    gout, disp, coords = _c(gout), _c(disp), _c(coords)
    B, _, H, W = disp.shape
    N = B * H * W
    ggeo = [np.zeros((N * C, D >> i), np.float32) for i in range(num_levels)]
    ginit = [np.zeros((N, W2 >> i), np.float32) for i in range(num_levels)]
    lib().orc_geo_lookup_bwd(_p(gout), _p(disp), _p(coords), _pp(ggeo), _pp(ginit), B, C, D, H, W, W2, num_levels, radius)
    return ggeo, ginit
