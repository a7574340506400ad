// pcv_cgi.hip -- the two remaining registered meta-architectures' volumes:
//   * PCVNet correlation block, meta_arch/pcvnet/corr.py:18-61: pyramid pooled by a compress
//     factor of 4 (or 2) and a lookup whose tap spacing is a per-pixel, per-gaussian sigma;
//   * CGI-Stereo normalised correlation, meta_arch/cgi/submodule.py:143-180: features divided
//     by (group L2 norm + 1e-5), then the group-wise mean correlation (dkt_gwc_volume).
// HBM-bound streaming kernels; sampler arithmetic = dkt_tap / dkt_blend (bit-identical to the
// reference's bilinear_sampler, see dkt_common.h).
#include "dkt_common.h"

// F.avg_pool2d(x, [1,f], stride=[1,f]) on rows (corr.py:29-31): window summed left to right,
// divided by f; floor on widths f does not divide.
__global__ __launch_bounds__(256) void pool_rows_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        long rows, int W, int f) {
    const int wo = W / f;
    const long total = rows * wo;
    const float ff = (float)f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / wo;
        const int k = (int)(i - n * wo);
        const float *p = src + n * W + (long)k * f;
        float s = 0.0f;
        for (int j = 0; j < f; ++j) s = __fadd_rn(s, p[j]);
        dst[i] = __fdiv_rn(s, ff);
    }
}

extern "C" int dkt_pool_rows(const float *src, float *dst, long rows, int W, int factor, int device, void *stream) {
    if (!src || !dst) return DKT_E_NULL;
    if (rows <= 0 || W <= 0 || factor < 1 || W / factor < 1) return DKT_E_SHAPE;
    DKT_ENTER(device);
    long blocks = (rows * (W / factor) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pool_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       src, dst, rows, W, factor);
    return dkt_launch_status();
}

struct PcvArgs {
    DktPtrs pyr;              // level i: (B*H*W1, W_i), W_0 = W2, W_{i+1} = W_i / f
    const float *coords;      // (B,G,H,W1)
    const float *sigma;       // (B,G,H,W1)
    float *out;               // (B, L*G*S, H, W1)
    long HW;
    int G, W2, L, S, f;
};

// thread = (pixel, gaussian); grid.y = level, grid.z = batch.  Lanes run along pixels, so the
// coords/sigma loads and all S stores of a wave are contiguous 256-byte segments.
__global__ __launch_bounds__(256) void pcv_lookup_kernel(PcvArgs a) {
    const long t = blockIdx.x * 256L + threadIdx.x;
    if (t >= a.HW * a.G) return;
    const int g = (int)(t / a.HW);
    const long p = t - (long)g * a.HW;
    const int lv = blockIdx.y, b = blockIdx.z;
    int wi = a.W2;
    float div = 1.0f;
    for (int i = 0; i < lv; ++i) {
        wi /= a.f;
        div = __fmul_rn(div, (float)a.f);
    }
    const float *row = a.pyr.p[lv] + ((long)b * a.HW + p) * wi;
    const long pg = ((long)b * a.G + g) * a.HW + p;
    const float c = a.coords[pg], sg = a.sigma[pg];
    const float wm1 = (float)(wi - 1);
    const float hwm1 = __fdiv_rn(wm1, 2.0f);
    float *o = a.out + (((long)b * a.L + lv) * a.G * a.S + (long)g * a.S) * a.HW + p;
    const int half = a.S / 2;
    // batches of 9 samples: the 18 taps are loaded unconditionally (clamped index, zero selected afterwards) so that they
    // are in flight together -- the guarded form was one branch and one round trip per tap
    for (int s0 = 0; s0 < a.S; s0 += 9) {
        DktTap tp[9];
        float v0[9], v1[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int s = min(s0 + k, a.S - 1);
            const float x = __fadd_rn(__fmul_rn((float)(s - half), sg), c);
            tp[k] = dkt_tap(__fdiv_rn(x, div), wm1, hwm1);
            const float f0 = fminf(fmaxf(tp[k].fl, 0.0f), wm1), f1 = fminf(fmaxf(tp[k].fl + 1.0f, 0.0f), wm1);
            v0[k] = row[(int)f0];
            v1[k] = row[(int)f1];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (s0 + k >= a.S) break;
            const float a0 = (tp[k].fl >= 0.0f && tp[k].fl <= wm1) ? v0[k] : 0.0f;
            const float a1 = (tp[k].fl + 1.0f >= 0.0f && tp[k].fl + 1.0f <= wm1) ? v1[k] : 0.0f;
            o[(long)(s0 + k) * a.HW] = dkt_blend(a0, a1, tp[k]);
        }
    }
}

extern "C" int dkt_pcv_lookup(const float *const *pyr, const float *coords, const float *sigma, float *out,
                              int B, int G, int H, int W1, int W2, int L, int S, int factor,
                              int device, void *stream) {
    if (!pyr || !coords || !sigma || !out) return DKT_E_NULL;
    if (B <= 0 || G <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || S <= 0 || (S & 1) == 0 || factor < 2 || B > 65535)
        return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS) return DKT_E_LEVELS;
    PcvArgs a;
    int wi = W2;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.pyr.p[i] = i < L ? pyr[i] : nullptr;
        if (i < L) {
            if (!pyr[i]) return DKT_E_NULL;
            if (wi < 1) return DKT_E_LEVELS;
            wi /= factor;
        }
    }
    a.coords = coords; a.sigma = sigma; a.out = out;
    a.HW = (long)H * W1;
    a.G = G; a.W2 = W2; a.L = L; a.S = S; a.f = factor;
    DKT_ENTER(device);
    dim3 grid((unsigned)((a.HW * G + 255) / 256), (unsigned)L, (unsigned)B);
    hipLaunchKernelGGL(pcv_lookup_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}

// y[b,c,p] = x[b,c,p] / (sqrt(sum_{c' in group(c)} x[b,c',p]^2) + eps)   (cgi/submodule.py:149,168)
__global__ __launch_bounds__(256) void group_l2norm_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                           int C, int G, long HW, float eps, long total) {
    const int cpg = C / G;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long bg = i / HW;                     // b*G + g
        const long p = i - bg * HW;
        const float *src = x + bg * cpg * HW + p;
        float *dst = y + bg * cpg * HW + p;
        // sixteen channel planes in flight per thread (the plain loop waited for every load before issuing the next:
        // 46 us for 96 channels at 184x312); the sum keeps its ascending channel order
        float s = 0.0f;
        for (int j0 = 0; j0 < cpg; j0 += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[(long)min(j0 + j, cpg - 1) * HW];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j0 + j < cpg) s = __fadd_rn(s, __fmul_rn(v[j], v[j]));
        }
        const float d = __fadd_rn(__fsqrt_rn(s), eps);
        for (int j0 = 0; j0 < cpg; j0 += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = src[(long)min(j0 + j, cpg - 1) * HW];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j0 + j < cpg) dst[(long)(j0 + j) * HW] = __fdiv_rn(v[j], d);
        }
    }
}

extern "C" int dkt_group_l2norm(const float *x, float *y, int B, int C, long HW, int G, float eps,
                                int device, void *stream) {
    if (!x || !y) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G != 0) return DKT_E_SHAPE;
    DKT_ENTER(device);
    const long total = (long)B * G * HW;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(group_l2norm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       x, y, C, G, HW, eps, total);
    return dkt_launch_status();
}
