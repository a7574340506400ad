// corr1d_bwd.hip -- backward of the RAFT correlation lookup and pyramid: what
// autograd derives for core/corr.py:127-146 (bilinear_sampler = grid_sample, zero padding,
// align_corners) and :119-125 (avg_pool2d chain).  The caller detaches the coordinates before
// every lookup (raft_stereo.py:152), so only the gradient w.r.t. the volume exists.
//
// A pixel's volume row is touched by that pixel's taps only: the scatter needs no atomics --
// one thread owns (pixel, level) and adds its 2*(2r+1) contributions in tap order, the order
// grid_sampler_2d_backward visits the sample points (results are bit-identical to autograd on
// the CPU, tests/golden/corr_bwd.npz).  HBM-bound: the gradient rows are written once
// (zero-filled by the caller) and the taps touch ~40 B of each.
#include "dkt_common.h"

struct LookupBwdArgs {
    DktMutPtrs gpyr;           // level i: (B*H*W1, W2>>i), zero-initialised, accumulated into
    const float *gout;         // (B, L*K, H, W1)
    const float *coords_x;
    long coords_bstride;
    long HW;
    int W2, L;
};

template <int R>
__global__ __launch_bounds__(256) void corr1d_lookup_bwd_kernel(LookupBwdArgs a) {
    constexpr int K = 2 * R + 1;
    const long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= a.HW) return;
    const int lv = blockIdx.y, b = blockIdx.z;
    const int wi = a.W2 >> lv;
    float *row = a.gpyr.p[lv] + ((long)b * a.HW + p) * wi;
    const float cx = a.coords_x[(size_t)b * a.coords_bstride + p];
    const float xc = __fdiv_rn(cx, (float)(1 << lv));
    const float wm1 = (float)(wi - 1);
    const float hwm1 = __fdiv_rn(wm1, 2.0f);
    const float *g = a.gout + ((size_t)b * a.L * K + (size_t)lv * K) * a.HW + p;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const DktTap t = dkt_tap(__fadd_rn((float)(k - R), xc), wm1, hwm1);
        const float gv = g[(size_t)k * a.HW];
        if (t.fl >= 0.0f && t.fl <= wm1) {
            float *q = row + (int)t.fl;
            *q = __fadd_rn(*q, __fmul_rn(gv, t.e));
        }
        if (t.fl + 1.0f >= 0.0f && t.fl + 1.0f <= wm1) {
            float *q = row + (int)t.fl + 1;
            *q = __fadd_rn(*q, __fmul_rn(gv, t.w));
        }
    }
}

template <int R>
static void launch_lookup_bwd(const LookupBwdArgs &a, int B, hipStream_t st) {
    dim3 grid((unsigned)((a.HW + 255) / 256), (unsigned)a.L, (unsigned)B);
    hipLaunchKernelGGL(corr1d_lookup_bwd_kernel<R>, grid, dim3(256), 0, st, a);
}

extern "C" int dkt_corr1d_lookup_bwd(const float *grad_out, const float *coords_x, long coords_bstride,
                                     float *const *grad_pyr, int B, int H, int W1, int W2, int L, int r,
                                     int device, void *stream) {
    if (!grad_out || !coords_x || !grad_pyr) return DKT_E_NULL;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    LookupBwdArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.gpyr.p[i] = i < L ? grad_pyr[i] : nullptr;
        if (i < L && !grad_pyr[i]) return DKT_E_NULL;
    }
    a.gout = grad_out; a.coords_x = coords_x; a.coords_bstride = coords_bstride;
    a.HW = (long)H * W1; a.W2 = W2; a.L = L;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    switch (r) {
        case 0: launch_lookup_bwd<0>(a, B, st); break;
        case 1: launch_lookup_bwd<1>(a, B, st); break;
        case 2: launch_lookup_bwd<2>(a, B, st); break;
        case 3: launch_lookup_bwd<3>(a, B, st); break;
        case 4: launch_lookup_bwd<4>(a, B, st); break;
        case 5: launch_lookup_bwd<5>(a, B, st); break;
        case 6: launch_lookup_bwd<6>(a, B, st); break;
        case 7: launch_lookup_bwd<7>(a, B, st); break;
        default: launch_lookup_bwd<8>(a, B, st); break;
    }
    return dkt_launch_status();
}

// Chain of avg_pool2d([1,2]) backward passes folded into one pass over level 0:
//   T_{L-1} = g_{L-1};  T_i[c] = g_i[c] + T_{i+1}[c/2] / 2  (a level's odd last column receives
//   nothing);  out = T_0 / divisor  (divisor = sqrt(C): backward of corr / sqrt(C), corr.py:156).
struct PoolBwdArgs {
    DktPtrs gpyr;
    float *out;
    long rows;
    int W2, L;
    float divisor;
};

__global__ __launch_bounds__(256) void corr1d_pool_bwd_kernel(PoolBwdArgs a) {
    const long total = a.rows * a.W2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / a.W2;
        const int c = (int)(i - n * a.W2);
        int deepest = 0;
        for (int l = 1; l < a.L; ++l) {
            if ((c >> l) < (a.W2 >> l)) deepest = l; else break;
        }
        float t = a.gpyr.p[deepest][n * (long)(a.W2 >> deepest) + (c >> deepest)];
        for (int l = deepest - 1; l >= 0; --l)
            t = __fadd_rn(a.gpyr.p[l][n * (long)(a.W2 >> l) + (c >> l)], __fdiv_rn(t, 2.0f));
        a.out[i] = __fdiv_rn(t, a.divisor);
    }
}

extern "C" int dkt_corr1d_pool_bwd(const float *const *grad_pyr, float *grad_vol, int B, int H, int W1, int W2,
                                   int L, float divisor, int device, void *stream) {
    if (!grad_pyr || !grad_vol) return DKT_E_NULL;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || !(divisor > 0.0f)) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    PoolBwdArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.gpyr.p[i] = i < L ? grad_pyr[i] : nullptr;
        if (i < L && !grad_pyr[i]) return DKT_E_NULL;
    }
    a.out = grad_vol; a.rows = (long)B * H * W1; a.W2 = W2; a.L = L; a.divisor = divisor;
    DKT_ENTER(device);
    long blocks = (a.rows * W2 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(corr1d_pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}
