// geo.hip -- IGEV Combined_Geo_Encoding_Volume lookup for gfx950.
// Reference: meta_arch/igev_stereo/geometry.py:6-58.
//
// The reference first permutes the (B,C,D,H,W) geometry volume to (N,C,1,D)
// (an 88 MB copy at 184x312, geometry.py:18) so that grid_sample can treat D as
// the width axis.  Here the volume is read in its native layout: for a fixed
// (c,d) plane neighbouring pixels are neighbouring floats, so with lanes along
// w a tap load is a (near-)contiguous wave read whenever disparity is locally
// smooth, and no copy is made.
#include "dkt_common.h"

struct GeoArgs {
    DktPtrs geo;   // level i: (B,C,D>>i,H,W)
    DktPtrs init;  // level i: (B*H*W, W2>>i)
    const float *disp;
    const float *coords;
    float *out;
    long HW;
    int C, D, W2, L, ngrp;
    float inv_d[DKT_MAX_LEVELS];   // RN(1 / ((D>>i) - 1)), host-computed
    float inv_w[DKT_MAX_LEVELS];   // RN(1 / ((W2>>i) - 1))
};

__device__ __forceinline__ int geo_clamp_idx(float fl, int W) {
    return (int)fminf(fmaxf(fl, -2.0f), (float)W + 1.0f);
}

// thread = (pixel, level, channel group).  The 2r+1 taps along D depend on the pixel's disparity
// only, so a thread computes them once (reciprocal-based exact division, dkt_tap_rcp) and applies
// them to GEO_GC channels of the geometry volume; the last group of a level is the init-correlation
// row (its own taps).  The first form had one thread per channel: 8x the tap arithmetic, 23 % of HBM;
// groups of 4 with a select per load: 43 %.  Now all 8 channels of IGEV's volume are one group and
// the window loads are unconditional (clamped plane index, zeroed by a select afterwards), so that a
// lane has its 8 x (2r+2) loads in flight together instead of one round trip per guarded load.
#define GEO_GC 8
template <int R>
__global__ __launch_bounds__(256) void geo_lookup_kernel(GeoArgs a) {
    constexpr int K = 2 * R + 1;
    const long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= a.HW) return;
    const int lv = blockIdx.y / (a.ngrp + 1);
    const int grp = blockIdx.y % (a.ngrp + 1);
    const int b = blockIdx.z;
    const size_t n = (size_t)b * a.HW + p;
    const float inv = (float)(1 << lv);
    const float dl = __fdiv_rn(a.disp[n], inv);
    const int per_level = K * (a.C + 1);
    float *obase = a.out + ((size_t)b * a.L * per_level + (size_t)lv * per_level) * a.HW + p;

    if (grp < a.ngrp) {
        // geometry volume: taps along D, plane stride HW
        const int di = a.D >> lv;
        const float wm1 = (float)(di - 1), hwm1 = __fdiv_rn(wm1, 2.0f);
        DktTap taps[K];
        if (di > 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) taps[k] = dkt_tap_rcp(__fadd_rn((float)(k - R), dl), wm1, a.inv_d[lv], hwm1);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) taps[k] = dkt_tap(__fadd_rn((float)(k - R), dl), wm1, hwm1);
        }
        const int i0 = geo_clamp_idx(taps[0].fl, di);
        bool regular = true;                       // all taps inside the K+1 window (always, up to rounding)
#pragma unroll
        for (int k = 0; k < K; ++k) regular = regular && geo_clamp_idx(taps[k].fl, di) == i0 + k;
        const int c0 = grp * GEO_GC;
#pragma unroll
        for (int cc = 0; cc < GEO_GC; ++cc) {
            if (c0 + cc >= a.C) break;
            const int c = c0 + cc;
            const float *base = a.geo.p[lv] + ((size_t)b * a.C + c) * (size_t)di * a.HW + p;
            float *o = obase + (size_t)c * K * a.HW;
            float win[K + 1];
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                const int d = i0 + j;
                const bool in = d >= 0 && d < di;
                const float x = base[(size_t)(in ? d : 0) * a.HW];
                win[j] = in ? x : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float v0 = win[k], v1 = win[k + 1];
                if (!regular) {
                    const int ik = geo_clamp_idx(taps[k].fl, di);
                    if (ik != i0 + k) {
                        v0 = (ik >= 0 && ik < di) ? base[(size_t)ik * a.HW] : 0.0f;
                        v1 = (ik + 1 >= 0 && ik + 1 < di) ? base[(size_t)(ik + 1) * a.HW] : 0.0f;
                    }
                }
                o[(size_t)k * a.HW] = dkt_blend(v0, v1, taps[k]);
            }
        }
    } else {
        // init correlation row: x = (coords/2^i - disp/2^i) + dx   (geometry.py:50)
        const int wi = a.W2 >> lv;
        const float wm1 = (float)(wi - 1), hwm1 = __fdiv_rn(wm1, 2.0f);
        const float *row = a.init.p[lv] + n * (size_t)wi;
        const float cl = __fdiv_rn(a.coords[n], inv);
        const float xc = __fsub_rn(cl, dl);
        float *o = obase + (size_t)a.C * K * a.HW;
        DktTap taps[K];
        if (wi > 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) taps[k] = dkt_tap_rcp(__fadd_rn(xc, (float)(k - R)), wm1, a.inv_w[lv], hwm1);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) taps[k] = dkt_tap(__fadd_rn(xc, (float)(k - R)), wm1, hwm1);
        }
        const int i0 = geo_clamp_idx(taps[0].fl, wi);
        float win[K + 1];
#pragma unroll
        for (int j = 0; j <= K; ++j) {
            const int x = i0 + j;
            const bool in = x >= 0 && x < wi;
            const float v = row[in ? x : 0];
            win[j] = in ? v : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int ik = geo_clamp_idx(taps[k].fl, wi);
            float v0 = win[k], v1 = win[k + 1];
            if (ik != i0 + k) {
                v0 = (ik >= 0 && ik < wi) ? row[ik] : 0.0f;
                v1 = (ik + 1 >= 0 && ik + 1 < wi) ? row[ik + 1] : 0.0f;
            }
            o[(size_t)k * a.HW] = dkt_blend(v0, v1, taps[k]);
        }
    }
}

template <int R>
static void launch_geo(const GeoArgs &a, int B, hipStream_t st) {
    dim3 grid((unsigned)((a.HW + 255) / 256), (unsigned)(a.L * (a.ngrp + 1)), (unsigned)B);
    hipLaunchKernelGGL(geo_lookup_kernel<R>, grid, dim3(256), 0, st, a);
}

extern "C" int dkt_geo_lookup(const float *const *geo_pyr, const float *const *init_pyr,
                              const float *disp, const float *coords, float *out,
                              int B, int C, int D, int H, int W, int W2, int L, int r,
                              int device, void *stream) {
    if (!geo_pyr || !init_pyr || !disp || !coords || !out) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || W2 <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0 || (D >> (L - 1)) == 0) return DKT_E_LEVELS;
    if ((long)L * (C + 1) > 65535) return DKT_E_SHAPE;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    GeoArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.geo.p[i] = i < L ? geo_pyr[i] : nullptr;
        a.init.p[i] = i < L ? init_pyr[i] : nullptr;
        if (i < L && (!geo_pyr[i] || !init_pyr[i])) return DKT_E_NULL;
    }
    DKT_ENTER(device);
    a.disp = disp; a.coords = coords; a.out = out;
    a.HW = (long)H * W; a.C = C; a.D = D; a.W2 = W2; a.L = L;
    a.ngrp = (C + GEO_GC - 1) / GEO_GC;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        const int di = i < L ? (D >> i) : 0, wi = i < L ? (W2 >> i) : 0;
        a.inv_d[i] = di > 1 ? (float)(1.0 / (double)(di - 1)) : 0.0f;
        a.inv_w[i] = wi > 1 ? (float)(1.0 / (double)(wi - 1)) : 0.0f;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (r) {
        case 0: launch_geo<0>(a, B, st); break;
        case 1: launch_geo<1>(a, B, st); break;
        case 2: launch_geo<2>(a, B, st); break;
        case 3: launch_geo<3>(a, B, st); break;
        case 4: launch_geo<4>(a, B, st); break;
        case 5: launch_geo<5>(a, B, st); break;
        case 6: launch_geo<6>(a, B, st); break;
        case 7: launch_geo<7>(a, B, st); break;
        default: launch_geo<8>(a, B, st); break;
    }
    return dkt_launch_status();
}

// (B*C, D, HW) -> (B*C, D/2, HW), pairwise mean along D  (geometry.py:23-25)
__global__ __launch_bounds__(256) void pool_d_kernel(const float *__restrict__ src,
                                                     float *__restrict__ dst, int D, long HW) {
    const int Do = D >> 1;
    const long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= HW) return;
    const long bc = blockIdx.y / Do;
    const int d = blockIdx.y % Do;
    const float *s = src + (bc * D + 2 * d) * HW + p;
    dst[(bc * Do + d) * HW + p] = __fmul_rn(__fadd_rn(s[0], s[HW]), 0.5f);
}

extern "C" int dkt_pool_d(const float *src, float *dst, long BC, int D, long HW, int device, void *stream) {
    if (!src || !dst) return DKT_E_NULL;
    if (BC <= 0 || D < 2 || HW <= 0 || BC * (D >> 1) > 65535) return DKT_E_SHAPE;
    DKT_ENTER(device);
    dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(BC * (D >> 1)));
    hipLaunchKernelGGL(pool_d_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, D, HW);
    return dkt_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Backward of the geometry-volume lookup and of the D-axis pyramid (IGEV flavour): what
// autograd derives for geometry.py:34-58 and :23-25.  disp is detached by the caller (igev_stereo.py:200).
// The (b,c,.,h,w) column of the geometry volume and the init-correlation row of a pixel are touched by
// that pixel only: plain read-modify-write, taps in k order (bit-identical to autograd on the CPU).
// ---------------------------------------------------------------------------------------------
struct GeoBwdArgs {
    DktMutPtrs ggeo;   // level i: (B,C,D>>i,H,W), zero-initialised / accumulated into
    DktMutPtrs ginit;  // level i: (B*H*W, W2>>i)
    const float *gout; // (B, L*K*(C+1), H, W)
    const float *disp;
    const float *coords;
    long HW;
    int C, D, W2, L;
};

template <int R>
__global__ __launch_bounds__(256) void geo_lookup_bwd_kernel(GeoBwdArgs a) {
    constexpr int K = 2 * R + 1;
    const long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= a.HW) return;
    const int lv = blockIdx.y / (a.C + 1);
    const int c = blockIdx.y % (a.C + 1);
    const int b = blockIdx.z;
    const size_t n = (size_t)b * a.HW + p;
    const float inv = (float)(1 << lv);
    const float dl = __fdiv_rn(a.disp[n], inv);
    const int per_level = K * (a.C + 1);
    const float *g = a.gout + ((size_t)b * a.L * per_level + (size_t)lv * per_level + (size_t)c * K) * a.HW + p;
    const bool geo = c < a.C;
    const int width = geo ? (a.D >> lv) : (a.W2 >> lv);
    const long stride = geo ? a.HW : 1;                               // element stride along the sampled axis
    float *base = geo ? a.ggeo.p[lv] + ((size_t)b * a.C + c) * (size_t)width * a.HW + p
                      : a.ginit.p[lv] + n * (size_t)width;
    const float x0 = geo ? dl : __fsub_rn(__fdiv_rn(a.coords[n], inv), dl);
    const float wm1 = (float)(width - 1), hwm1 = __fdiv_rn(wm1, 2.0f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // geometry.py:41 adds dx to disp, :50 adds dx after the subtraction: both are x0 + dx in this order
        const float x = geo ? __fadd_rn((float)(k - R), x0) : __fadd_rn(x0, (float)(k - R));
        const DktTap t = dkt_tap(x, wm1, hwm1);
        const float gv = g[(size_t)k * a.HW];
        if (t.fl >= 0.0f && t.fl <= wm1) {
            float *q = base + (long)(int)t.fl * stride;
            *q = __fadd_rn(*q, __fmul_rn(gv, t.e));
        }
        if (t.fl + 1.0f >= 0.0f && t.fl + 1.0f <= wm1) {
            float *q = base + (long)((int)t.fl + 1) * stride;
            *q = __fadd_rn(*q, __fmul_rn(gv, t.w));
        }
    }
}

template <int R>
static void launch_geo_bwd(const GeoBwdArgs &a, int B, hipStream_t st) {
    dim3 grid((unsigned)((a.HW + 255) / 256), (unsigned)(a.L * (a.C + 1)), (unsigned)B);
    hipLaunchKernelGGL(geo_lookup_bwd_kernel<R>, grid, dim3(256), 0, st, a);
}

extern "C" int dkt_geo_lookup_bwd(const float *grad_out, const float *disp, const float *coords,
                                  float *const *grad_geo, float *const *grad_init,
                                  int B, int C, int D, int H, int W, int W2, int L, int r,
                                  int device, void *stream) {
    if (!grad_out || !disp || !coords || !grad_geo || !grad_init) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || W2 <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0 || (D >> (L - 1)) == 0) return DKT_E_LEVELS;
    if ((long)L * (C + 1) > 65535) return DKT_E_SHAPE;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    GeoBwdArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.ggeo.p[i] = i < L ? grad_geo[i] : nullptr;
        a.ginit.p[i] = i < L ? grad_init[i] : nullptr;
        if (i < L && (!grad_geo[i] || !grad_init[i])) return DKT_E_NULL;
    }
    a.gout = grad_out; a.disp = disp; a.coords = coords;
    a.HW = (long)H * W; a.C = C; a.D = D; a.W2 = W2; a.L = L;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    switch (r) {
        case 0: launch_geo_bwd<0>(a, B, st); break;
        case 1: launch_geo_bwd<1>(a, B, st); break;
        case 2: launch_geo_bwd<2>(a, B, st); break;
        case 3: launch_geo_bwd<3>(a, B, st); break;
        case 4: launch_geo_bwd<4>(a, B, st); break;
        case 5: launch_geo_bwd<5>(a, B, st); break;
        case 6: launch_geo_bwd<6>(a, B, st); break;
        case 7: launch_geo_bwd<7>(a, B, st); break;
        default: launch_geo_bwd<8>(a, B, st); break;
    }
    return dkt_launch_status();
}

// Chain of the pairwise-mean poolings along D folded into the gradient of the un-pooled volume:
//   T_{L-1} = g_{L-1};  T_i[d] = g_i[d] + T_{i+1}[d/2] / 2   (a level's odd last plane receives nothing)
struct GeoPoolBwdArgs {
    DktPtrs g;       // level i: (BC, D>>i, HW)
    float *out;      // (BC, D, HW)
    long HW, total;
    int D, L;
};

__global__ __launch_bounds__(256) void geo_pool_bwd_kernel(GeoPoolBwdArgs a) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < a.total; i += (long)gridDim.x * 256) {
        const long p = i % a.HW;
        const int d = (int)((i / a.HW) % a.D);
        const long bc = i / (a.HW * a.D);
        int deepest = 0;
        for (int l = 1; l < a.L; ++l) {
            if ((d >> l) < (a.D >> l)) deepest = l; else break;
        }
        float t = a.g.p[deepest][(bc * (a.D >> deepest) + (d >> deepest)) * a.HW + p];
        for (int l = deepest - 1; l >= 0; --l)
            t = __fadd_rn(a.g.p[l][(bc * (a.D >> l) + (d >> l)) * a.HW + p], __fdiv_rn(t, 2.0f));
        a.out[i] = t;
    }
}

extern "C" int dkt_geo_pool_bwd(const float *const *grad_geo, float *grad_vol, long BC, int D, long HW, int L,
                                int device, void *stream) {
    if (!grad_geo || !grad_vol) return DKT_E_NULL;
    if (BC <= 0 || D <= 0 || HW <= 0) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (D >> (L - 1)) == 0) return DKT_E_LEVELS;
    GeoPoolBwdArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.g.p[i] = i < L ? grad_geo[i] : nullptr;
        if (i < L && !grad_geo[i]) return DKT_E_NULL;
    }
    a.out = grad_vol; a.HW = HW; a.D = D; a.L = L; a.total = BC * D * HW;
    DKT_ENTER(device);
    long blocks = (a.total + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(geo_pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}
