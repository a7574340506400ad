// corr_feat.hip -- correlation lookup fused with the motion encoder's first layer.
//
// Reference: every GRU iteration runs  corr = corr_fn(coords1)  (core/corr.py:127-146, a
// (B, L*K, H, W) tensor: 8.3 MB at 1/4 KITTI) and then  cor = relu(convc1(corr))
// (core/update.py:72,79: a 1x1 convolution L*K -> 64).  As two kernels that is one dependent
// coords -> taps -> store round trip on 3.5 waves per SIMD followed by a launch whose whole input
// is the tensor just written.  Here one wave owns a 32-pixel segment of an image row:
//   1. samples all L levels of the SKEWED pyramid (corr1d_skew.hip; same taps, same blend as
//      dkt_corr1d_lookup_skew: the sampled values are bit-identical -- `tap` exposes them);
//      the two half-waves take alternate levels of the same 32 pixels, values stay in registers;
//   2. multiplies them by the (Cout x L*K) weight matrix on the matrix cores with the EXACT fp32
//      MFMA (v_mfma_f32_32x32x2_f32: an fp32 fma chain, no operand rounding): A = weights, B =
//      samples; with the levels split over the half-waves a lane's sample IS its B-fragment entry;
//   3. adds the bias, applies ReLU and stores (Cout, 32 px) in 128-byte NCHW rows.
// HBM traffic per pixel: L*(K+1)*4 B of pyramid + 4 B of coordinate read, Cout*4 B written; the
// lookup tensor never exists.
#include "dkt_common.h"
#include <cstdlib>
#include <type_traits>

typedef float cf_f32x16 __attribute__((ext_vector_type(16)));

struct CorrFeatArgs {
    DktPtrs skew;
    const float *coords_x;
    long coords_bstride;
    const float *w;          // (L*K, Cout): the layer's weight TRANSPOSED (k-major: lanes along co are contiguous)
    const float *bias;       // (Cout) or null
    float *out;              // (B, Cout, H, W1)
    long out_bstride;
    float *tap;              // optional (B, L*K, H, W1): the sampled correlation values
    long tap_bstride;
    long HW;
    int H, W1, W2, pitch, nseg, Cout, relu;
    float inv_wm1[DKT_MAX_LEVELS];
    // optional C8S destination (the operand layout of conv_c8.hip; 16-pixel form only)
    char *out_c8;
    long out_c8_bs, out_c8_plane;
    int out_c8_Wp, out_c8_ch0;
    float act_scale;
};

__device__ __forceinline__ int cf_clamp_idx(float fl, int W) {
    return (int)fminf(fmaxf(fl, -2.0f), (float)W + 1.0f);
}

// the sampler arithmetic of corr1d_skew.hip (skw_tap): reciprocal form of the reference's division,
// bit-identical to dkt_tap
__device__ __forceinline__ DktTap cf_tap(float x, float wm1, float inv, float half_wm1) {
    const float a2 = __fmul_rn(2.0f, x);
    float q = __fmul_rn(a2, inv);
    q = __fmaf_rn(__fmaf_rn(-wm1, q, a2), inv, q);
    q = __fmaf_rn(__fmaf_rn(-wm1, q, a2), inv, q);
    const float xg = __fsub_rn(q, 1.0f);
    const float ix = __fmul_rn(__fadd_rn(xg, 1.0f), half_wm1);
    DktTap t;
    t.fl = floorf(ix);
    t.w = __fsub_rn(ix, t.fl);
    t.e = __fsub_rn(1.0f, t.w);
    return t;
}

// One wave = one 32-pixel segment of one image row.  Lane l: pixel l & 31; the two half-waves split the
// LEVELS: half g = l >> 5 samples levels g, g + 2, ... (NV = ceil(L/2) * K values per lane).  MFMA step s
// multiplies the pair (half 0's s-th value, half 1's s-th value): that IS the B fragment layout of
// v_mfma_f32_32x32x2_f32 (lane l holds B[k = l >> 5][n = l & 31]) -- no data movement between lanes;
// the A fragment pairs the matching weight columns.  Twice the waves and half the dependent chain of
// a 64-pixel-per-wave form (the kernel is latency-bound: 1 800 waves for 1 024 SIMDs at 1/4 KITTI).
// Block = 4 waves; grid.y = batch.  MC = ceil(Cout / 32) accumulator row blocks.
template <int L, int R, int MC>
__global__ __launch_bounds__(256) void corr_feat_kernel(CorrFeatArgs a) {
    constexpr int K = 2 * R + 1;
    constexpr int NK = L * K;
    constexpr int LH = (L + 1) / 2;           // levels per half-wave
    constexpr int NV = LH * K;                // values per lane = MFMA steps
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, g = lane >> 5;
    const long gw = blockIdx.x * 4L + (threadIdx.x >> 6);
    const long hrow = gw / a.nseg;
    if (hrow >= a.H) return;                  // wave-uniform
    const int seg0 = (int)(gw - hrow * a.nseg) * 32;
    const int w1 = seg0 + li;
    const bool live = w1 < a.W1;
    const int w1c = live ? w1 : a.W1 - 1;     // surplus lanes shadow the last pixel (their results are dropped)
    const int b = blockIdx.y;
    const long p = hrow * a.W1 + w1c;
    const float cx = a.coords_x[(size_t)b * a.coords_bstride + p];

    // ---- weights (k-major: wt[k][co]): A fragment of step s, row block m = W[co = 32m + li][k(g, s)]
    float A[MC][NV];
#pragma unroll
    for (int s = 0; s < NV; ++s) {
        const int lv = g + 2 * (s / K);
        const int k = lv * K + (s % K);
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            const int co = 32 * m + li;
            const bool ok = lv < L && co < a.Cout;
            const float wv = a.w[(long)(ok ? k : 0) * a.Cout + (ok ? co : 0)];
            A[m][s] = ok ? wv : 0.0f;
        }
    }

    // biases of this lane's output rows, fetched as one batch (clamped index, no per-element branch)
    float bv[MC][16];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[m][r] = 0.0f;
    if (a.bias) {
#pragma unroll
        for (int m = 0; m < MC; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * g;
                bv[m][r] = a.bias[co < a.Cout ? co : a.Cout - 1];
            }
    }

    // ---- the lookup: v[j*K + k] = sample k of level g + 2j for this lane's pixel.  All window loads are
    // unconditional (clamped address, zeroed by a select afterwards): the LH*(K+1) loads of a lane are
    // in flight together.  Same taps and blend as corr1d_lookup_skew_kernel -> bit-identical samples.
    float v[NV];
    float win[LH][K + 1];
    DktTap taps[LH][K];
    int i0[LH];
    bool odd = false;                         // some tap of this lane is not at i0 + k (only for non-finite x)
    auto sample_level = [&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        // level of this half-wave: g + 2j (compile-time candidates selected per half: a lane-indexed
        // kernel-argument array would cost a dependent memory load); the padding slot of an odd L
        // samples level L-1 again and is multiplied by zero weights
        constexpr int lvA = 2 * j < L ? 2 * j : L - 1, lvB = 2 * j + 1 < L ? 2 * j + 1 : L - 1;
        const int lv = g ? lvB : lvA;
        const int wi = a.W2 >> lv;
        const int qm = (w1c >> lv) % wi;
        const float *lvl = g ? a.skew.p[lvB] : a.skew.p[lvA];
        const float *base = lvl + (((long)b * a.H + hrow) * wi) * (long)a.pitch + w1c;
        const float xc = __fdiv_rn(cx, (float)(1 << lv));
        const float wm1 = (float)(wi - 1);
        const float hwm1 = __fdiv_rn(wm1, 2.0f);
        const float inv = g ? a.inv_wm1[lvB] : a.inv_wm1[lvA];
        // (levels of width 1 -- division by W-1 = 0 in the reference -- are refused by the host wrapper:
        // a branch here would split the block and serialise the two levels' loads)
#pragma unroll
        for (int k = 0; k < K; ++k) taps[j][k] = cf_tap(__fadd_rn((float)(k - R), xc), wm1, inv, hwm1);
        i0[j] = cf_clamp_idx(taps[j][0].fl, wi);
#pragma unroll
        for (int t = 0; t <= K; ++t) {
            const int c = i0[j] + t;
            const bool in = c >= 0 && c < wi;
            int sidx = (in ? c : 0) - qm;
            if (sidx < 0) sidx += wi;
            const float x = base[(long)sidx * a.pitch];
            win[j][t] = in ? x : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) odd |= cf_clamp_idx(taps[j][k].fl, wi) != i0[j] + k;
    };
    sample_level(std::integral_constant<int, 0>{});
    if constexpr (LH > 1) sample_level(std::integral_constant<int, 1>{});
#pragma unroll
    for (int j = 0; j < LH; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) v[j * K + k] = dkt_blend(win[j][k], win[j][k + 1], taps[j][k]);
    if (__any(odd)) {                         // rare: re-sample the irregular taps one by one
        auto resample = [&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            constexpr int lvA = 2 * j < L ? 2 * j : L - 1, lvB = 2 * j + 1 < L ? 2 * j + 1 : L - 1;
            const int lv = g ? lvB : lvA;
            const int wi = a.W2 >> lv;
            const int qm = (w1c >> lv) % wi;
            const float *base = (g ? a.skew.p[lvB] : a.skew.p[lvA]) + (((long)b * a.H + hrow) * wi) * (long)a.pitch + w1c;
            auto at = [&](int c) -> float {
                if (c < 0 || c >= wi) return 0.0f;
                int sidx = c - qm;
                if (sidx < 0) sidx += wi;
                return base[(long)sidx * a.pitch];
            };
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int ik = cf_clamp_idx(taps[j][k].fl, wi);
                if (ik != i0[j] + k) v[j * K + k] = dkt_blend(at(ik), at(ik + 1), taps[j][k]);
            }
        };
        resample(std::integral_constant<int, 0>{});
        if constexpr (LH > 1) resample(std::integral_constant<int, 1>{});
    }
    if (a.tap && live) {
        float *t = a.tap + (size_t)b * a.tap_bstride + p;
#pragma unroll
        for (int j = 0; j < LH; ++j)
            if (g + 2 * j < L) {
#pragma unroll
                for (int k = 0; k < K; ++k) t[(size_t)((g + 2 * j) * K + k) * a.HW] = v[j * K + k];
            }
    }

    // ---- 1x1 convolution on the exact-fp32 matrix pipe: D[co][px] += W[co][k] * v[k][px]
    cf_f32x16 acc[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[m][r] = 0.0f;
            asm volatile("" ::"v"(bv[m][r]));      // biases resident before the epilogue (no load per store there)
        }
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int m = 0; m < MC; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m][s], v[s], acc[m], 0, 0, 0);

    // ---- epilogue: C/D map col = l & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    float *ob = a.out + (size_t)b * a.out_bstride + hrow * a.W1 + w1c;
    const bool full = 32 * MC <= a.Cout && seg0 + 32 <= a.W1;      // wave-uniform: no guards needed
    if (full) {
#pragma unroll
        for (int m = 0; m < MC; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * g;
                float y = __fadd_rn(acc[m][r], bv[m][r]);
                if (a.relu) y = dkt_relu(y);
                ob[(size_t)co * a.HW] = y;
            }
    } else {
#pragma unroll
        for (int m = 0; m < MC; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * g;
                float y = __fadd_rn(acc[m][r], bv[m][r]);
                if (a.relu) y = dkt_relu(y);
                if (co < a.Cout && live) ob[(size_t)co * a.HW] = y;
            }
    }
}

template <int L, int R>
static int cf_launch(const CorrFeatArgs &a, int B, hipStream_t st) {
    const long waves = (long)a.H * a.nseg;
    dim3 grid((unsigned)((waves + 3) / 4), (unsigned)B);
    if (a.Cout <= 32) hipLaunchKernelGGL((corr_feat_kernel<L, R, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((corr_feat_kernel<L, R, 2>), grid, dim3(256), 0, st, a);
    return dkt_launch_status();
}

// Four-level form (L = 4, the RAFT-Stereo default): one wave = a 16-pixel segment, lane l: pixel l & 15, and the
// FOUR quarter-waves g = l >> 4 take one level each -- 10 window loads and 9 taps per lane, 3 600 waves for
// 1 024 SIMDs at 1/4 KITTI (the stand-alone lookup's parallelism; the 32-pixel form has half of it and twice the
// dependent chain per lane).  MFMA: v_mfma_f32_16x16x4_f32, whose B fragment is lane l -> B[k = l >> 4][n = l & 15]:
// step s multiplies tap s of the four levels.  MT = ceil(Cout / 16) accumulator tiles of 4 registers.
typedef float cf_f32x4 __attribute__((ext_vector_type(4)));

template <int R, int MT>
__global__ __launch_bounds__(256) void corr_feat16_kernel(CorrFeatArgs a) {
    constexpr int K = 2 * R + 1;
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, g = lane >> 4;
    const long gw = blockIdx.x * 4L + (threadIdx.x >> 6);
    const long hrow = gw / a.nseg;
    if (hrow >= a.H) return;                  // wave-uniform
    const int seg0 = (int)(gw - hrow * a.nseg) * 16;
    const int w1 = seg0 + li;
    const bool live = w1 < a.W1;
    const int w1c = live ? w1 : a.W1 - 1;
    const int b = blockIdx.y;
    const long p = hrow * a.W1 + w1c;
    const float cx = a.coords_x[(size_t)b * a.coords_bstride + p];

    // ---- weights (k-major: wt[k][co]): A fragment of step s, tile m = W[co = 16m + li][k = g*K + s]
    float A[MT][K];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = 16 * m + li;
            const bool ok = co < a.Cout;
            const float wv = a.w[(long)(g * K + s) * a.Cout + (ok ? co : 0)];
            A[m][s] = ok ? wv : 0.0f;
        }
    float bv[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[m][r] = 0.0f;
    if (a.bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * m + 4 * g + r;
                bv[m][r] = a.bias[co < a.Cout ? co : a.Cout - 1];
            }
    }

    // ---- the lookup of level g (selected among the four argument slots without a memory-indexed load)
    const int lv = g;
    const int wi = a.W2 >> lv;
    const int qm = (w1c >> lv) % wi;
    const float *lvl = g == 0 ? a.skew.p[0] : g == 1 ? a.skew.p[1] : g == 2 ? a.skew.p[2] : a.skew.p[3];
    const float inv = g == 0 ? a.inv_wm1[0] : g == 1 ? a.inv_wm1[1] : g == 2 ? a.inv_wm1[2] : a.inv_wm1[3];
    const float *base = lvl + (((long)b * a.H + hrow) * wi) * (long)a.pitch + w1c;
    const float xc = __fdiv_rn(cx, (float)(1 << lv));
    const float wm1 = (float)(wi - 1);
    const float hwm1 = __fdiv_rn(wm1, 2.0f);
    DktTap taps[K];
#pragma unroll
    for (int k = 0; k < K; ++k) taps[k] = cf_tap(__fadd_rn((float)(k - R), xc), wm1, inv, hwm1);
    const int i0 = cf_clamp_idx(taps[0].fl, wi);
    float win[K + 1];
#pragma unroll
    for (int t = 0; t <= K; ++t) {
        const int c = i0 + t;
        const bool in = c >= 0 && c < wi;
        int sidx = (in ? c : 0) - qm;
        if (sidx < 0) sidx += wi;
        const float x = base[(long)sidx * a.pitch];
        win[t] = in ? x : 0.0f;
    }
    bool odd = false;
#pragma unroll
    for (int k = 0; k < K; ++k) odd |= cf_clamp_idx(taps[k].fl, wi) != i0 + k;
    float v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = dkt_blend(win[k], win[k + 1], taps[k]);
    if (__any(odd)) {                         // rare (non-finite coordinates): re-sample the irregular taps one by one
        auto at = [&](int c) -> float {
            if (c < 0 || c >= wi) return 0.0f;
            int sidx = c - qm;
            if (sidx < 0) sidx += wi;
            return base[(long)sidx * a.pitch];
        };
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int ik = cf_clamp_idx(taps[k].fl, wi);
            if (ik != i0 + k) v[k] = dkt_blend(at(ik), at(ik + 1), taps[k]);
        }
    }
    if (a.tap && live) {
        float *t = a.tap + (size_t)b * a.tap_bstride + p;
#pragma unroll
        for (int k = 0; k < K; ++k) t[(size_t)(g * K + k) * a.HW] = v[k];
    }

    // ---- 1x1 convolution on the exact-fp32 matrix pipe: D[co][px] += W[co][level g, tap s] * v_g[s][px]
    cf_f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[m][r] = 0.0f;
            asm volatile("" ::"v"(bv[m][r]));      // biases resident before the epilogue
        }
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m][s], v[s], acc[m], 0, 0, 0);

    // ---- epilogue: C/D map of the 16x16 form: col = l & 15 (pixel), row = 4 (l >> 4) + r
    const bool full = 16 * MT <= a.Cout && seg0 + 16 <= a.W1;      // wave-uniform: no guards needed
    if (a.out_c8) {
        // C8S (fp16 hi | lo, 8 consecutive channels per 16 bytes): quarter-wave g holds channels 4g .. 4g+3 of tile m;
        // lanes l and l ^ 16 exchange their packed halves, then the even quarter stores the hi 16 bytes of channel
        // group 2m + (g >> 1), the odd quarter the lo 16 bytes -- one 16-byte store per lane and tile instead of four
        // 4-byte ones into four channel planes
        char *pb = a.out_c8 + (size_t)b * a.out_c8_bs + ((size_t)(hrow + 1) * a.out_c8_Wp + (w1c + 1)) * 16;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            unsigned hw[2], lw[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                float y0 = __fadd_rn(acc[m][2 * d], bv[m][2 * d]), y1 = __fadd_rn(acc[m][2 * d + 1], bv[m][2 * d + 1]);
                if (a.relu) { y0 = dkt_relu(y0); y1 = dkt_relu(y1); }
                if (16 * m + 4 * g + 2 * d >= a.Cout) y0 = 0.0f;
                if (16 * m + 4 * g + 2 * d + 1 >= a.Cout) y1 = 0.0f;
                y0 *= a.act_scale; y1 *= a.act_scale;
                const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
                union { _Float16 h[2]; unsigned u; } t0, t1;
                t0.h[0] = h0; t0.h[1] = h1;
                t1.h[0] = (_Float16)(y0 - (float)h0); t1.h[1] = (_Float16)(y1 - (float)h1);
                hw[d] = t0.u; lw[d] = t1.u;
            }
            // even quarters need the partner's hi words, odd quarters the partner's lo words
            const bool even = (g & 1) == 0;
            const unsigned s0 = __shfl_xor(even ? lw[0] : hw[0], 16), s1 = __shfl_xor(even ? lw[1] : hw[1], 16);
            // what arrived: from an odd partner (we are even) its hi words; from an even partner (we are odd) its lo words
            const uint4 v = even ? make_uint4(hw[0], hw[1], s0, s1) : make_uint4(s0, s1, lw[0], lw[1]);
            if (live) {
                const int grp = ((a.out_c8_ch0 + 16 * m) >> 3) + (g >> 1);
                *(uint4 *)(pb + (size_t)grp * 2 * a.out_c8_plane + (even ? 0 : a.out_c8_plane)) = v;
            }
        }
        return;
    }
    float *ob = a.out + (size_t)b * a.out_bstride + hrow * a.W1 + w1c;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * m + 4 * g + r;
            float y = __fadd_rn(acc[m][r], bv[m][r]);
            if (a.relu) y = dkt_relu(y);
            if (full || (co < a.Cout && live)) ob[(size_t)co * a.HW] = y;
        }
}

// Block form (L = 4): a block of four waves owns ONE 64-pixel row segment.  Phase 1: wave g samples level g for the 64
// pixels (lane = pixel: every window load of the wave is one 256-byte segment of the skewed pyramid -- the 16-pixel form
// moves the same bytes in 64-byte pieces) and leaves its K values per pixel in LDS.  Phase 2: wave m multiplies the
// 36 x 64 sample matrix with output channels 16m .. 16m+15 on v_mfma_f32_16x16x4_f32 (k ascending: level-major) and
// stores them -- C8S: 256 contiguous bytes per 16-pixel tile and half.  At cfg4's batch 8: 76 -> 62 us, DESIGN 3.3.
// (Round 4 tried two segments per block, software-pipelined -- segment 1's window loads in flight under segment 0's MFMAs and
// stores, weights fetched once: 70 us against 62 at B = 8.  Fewer, longer blocks lose more latency hiding than the pipelining
// inside one block wins; not kept.)
// bx = row segment index (row-major over (H, ceil(W1 / 64))), b = batch item; coord(b, p, live, g) = the x coordinate of pixel
// p = hrow * W1 + w1 (called by every lane of every wave g; the plain kernel reads coords_x, the fused motion-encoder front
// below computes the coordinate update of dkt_head_finish there).
template <int R, class CX>
__device__ __forceinline__ void corr_feat64_body(const CorrFeatArgs &a, long bx, int b, CX coord) {
    constexpr int K = 2 * R + 1;
    // sample (level g, tap k) = GEMM row kk = g K + k of pixel px lives at vs[((kk & 3) * 64 + px) * VP + (kk >> 2)]: phase 2's
    // lane (pixel, q = kk & 3) finds its NS = K values of a tile in one 48-byte run (three ds_read_b128 instead of K ds_read_b32)
    constexpr int VP = 12;
    static_assert((4 * K + 3) / 4 <= VP, "row pitch holds the k steps");
    __shared__ __attribute__((aligned(16))) float vs[4 * 64 * VP];
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nseg = (a.W1 + 63) / 64;
    const long hrow = bx / nseg;
    const int seg0 = (int)(bx - hrow * nseg) * 64;
    // phase 2's weight fragments (wave m = g: channels 16m .. 16m+15, k = 4 step + q) are fetched FIRST: left inside the
    // k loop each of them was an L2 round trip on the critical path behind the barrier (the 36 x 64 matrix is shared by
    // every block, but a block touches it once) -- issued here their latency hides under the sampling phase
    constexpr int NS = (4 * K) / 4;           // k steps of 4 (4 K is a multiple of 4)
    float Aw[NS];
    {
        const int j = lane & 15, q = lane >> 4;
        const int co = 16 * g + j;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int kk = 4 * sidx + q;
            const float wv = a.w[(long)kk * a.Cout + (co < a.Cout ? co : 0)];
            Aw[sidx] = co < a.Cout ? wv : 0.0f;
        }
    }
    {
        const int w1 = seg0 + lane;
        const bool live = w1 < a.W1;
        const int w1c = live ? w1 : a.W1 - 1;
        const long p = hrow * a.W1 + w1c;
        const float cx = coord(b, p, live, g);
        const int lv = g;
        const int wi = a.W2 >> lv;
        const int qm = (w1c >> lv) % wi;
        const float *lvl = a.skew.p[lv];
        const float inv = a.inv_wm1[lv];
        const float *base = lvl + (((long)b * a.H + hrow) * wi) * (long)a.pitch + w1c;
        const float xc = __fdiv_rn(cx, (float)(1 << lv));
        const float wm1 = (float)(wi - 1);
        const float hwm1 = __fdiv_rn(wm1, 2.0f);
        DktTap taps[K];
#pragma unroll
        for (int k = 0; k < K; ++k) taps[k] = cf_tap(__fadd_rn((float)(k - R), xc), wm1, inv, hwm1);
        const int i0 = cf_clamp_idx(taps[0].fl, wi);
        float win[K + 1];
#pragma unroll
        for (int t = 0; t <= K; ++t) {
            const int c = i0 + t;
            const bool in = c >= 0 && c < wi;
            int sidx = (in ? c : 0) - qm;
            if (sidx < 0) sidx += wi;
            const float x = base[(long)sidx * a.pitch];
            win[t] = in ? x : 0.0f;
        }
        bool odd = false;
#pragma unroll
        for (int k = 0; k < K; ++k) odd |= cf_clamp_idx(taps[k].fl, wi) != i0 + k;
        float v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = dkt_blend(win[k], win[k + 1], taps[k]);
        if (__any(odd)) {                     // rare (non-finite coordinates): re-sample the irregular taps one by one
            auto at = [&](int c) -> float {
                if (c < 0 || c >= wi) return 0.0f;
                int sidx = c - qm;
                if (sidx < 0) sidx += wi;
                return base[(long)sidx * a.pitch];
            };
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int ik = cf_clamp_idx(taps[k].fl, wi);
                if (ik != i0 + k) v[k] = dkt_blend(at(ik), at(ik + 1), taps[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) vs[(((g * K + k) & 3) * 64 + lane) * VP + ((g * K + k) >> 2)] = v[k];
        if (a.tap && live) {
            float *t = a.tap + (size_t)b * a.tap_bstride + p;
#pragma unroll
            for (int k = 0; k < K; ++k) t[(size_t)(g * K + k) * a.HW] = v[k];
        }
    }
    __syncthreads();
    // ---- phase 2: wave m = g: channels 16m .. 16m+15
    const int m = g;
    if (16 * m >= a.Cout) return;             // wave-uniform
    const int j = lane & 15, q = lane >> 4;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_{0.f, 0.f, 0.f, 0.f};
    // two pixel tiles at a time (24 fragment registers: all four would not fit the 64 the occupancy allows)
#pragma unroll
    for (int t0 = 0; t0 < 4; t0 += 2) {
        float bs[2][VP];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 *row = (const float4 *)(vs + (q * 64 + 16 * (t0 + t) + j) * VP);
#pragma unroll
            for (int i = 0; i < VP / 4; ++i) {
                const float4 f = row[i];
                bs[t][4 * i] = f.x; bs[t][4 * i + 1] = f.y; bs[t][4 * i + 2] = f.z; bs[t][4 * i + 3] = f.w;
            }
        }
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[sidx], bs[t][sidx], acc[t0 + t], 0, 0, 0);
        }
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 16 * m + 4 * q + r;
        bv[r] = a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w1 = seg0 + 16 * t + j;
        const bool live = w1 < a.W1;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[r] = __fadd_rn(acc[t][r], bv[r]);
            if (a.relu) y[r] = dkt_relu(y[r]);
            if (16 * m + 4 * q + r >= a.Cout) y[r] = 0.0f;
        }
        if (a.out_c8) {
            unsigned hw[2], lw[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float y0 = y[2 * d] * a.act_scale, y1 = y[2 * d + 1] * a.act_scale;
                const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
                union { _Float16 h[2]; unsigned u; } t0, t1;
                t0.h[0] = h0; t0.h[1] = h1;
                t1.h[0] = (_Float16)(y0 - (float)h0); t1.h[1] = (_Float16)(y1 - (float)h1);
                hw[d] = t0.u; lw[d] = t1.u;
            }
            const bool even = (q & 1) == 0;
            const unsigned s0 = __shfl_xor(even ? lw[0] : hw[0], 16), s1 = __shfl_xor(even ? lw[1] : hw[1], 16);
            const uint4 v = even ? make_uint4(hw[0], hw[1], s0, s1) : make_uint4(s0, s1, lw[0], lw[1]);
            if (live) {
                const int grp = ((a.out_c8_ch0 + 16 * m) >> 3) + (q >> 1);
                char *pb = a.out_c8 + (size_t)b * a.out_c8_bs + ((size_t)(hrow + 1) * a.out_c8_Wp + (w1 + 1)) * 16;
                *(uint4 *)(pb + (size_t)grp * 2 * a.out_c8_plane + (even ? 0 : a.out_c8_plane)) = v;
            }
        } else if (live) {
            float *ob = a.out + (size_t)b * a.out_bstride + hrow * a.W1 + w1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * m + 4 * q + r;
                if (co < a.Cout) ob[(size_t)co * a.HW] = y[r];
            }
        }
    }
}

template <int R>
__global__ __launch_bounds__(256, 8) void corr_feat64_kernel(CorrFeatArgs a) {
    corr_feat64_body<R>(a, (long)blockIdx.x, (int)blockIdx.y,
                        [&](int b, long p, bool, int) -> float { return a.coords_x[(size_t)b * a.coords_bstride + p]; });
}

// ---------------------------------------------------------------------------------------------------------
// The motion encoder's front as ONE launch (round 4): the coordinate update that closes the flow head
// (dkt_head_finish: coords1 += conv2(relu(conv1(h))), flow = coords1 - coords0; raft_stereo.py:165-168), the lookup
// + convc1 at the NEW coordinate (core/corr.py:127-146, core/update.py:76,84) and the 7x7 stem on the NEW flow
// (core/update.py:77,85) were three dependent launches of 14 + 23 + 39 us on the loop's critical chain.  Here blocks
// [0, nb_stem) are stem tiles (4 rows x 32 columns + 3 pixels of halo: the update is recomputed for the halo, 18 plane
// reads per position) and the rest are lookup row segments; both evaluate the update with dkt_head_finish's own
// operation order, so the lookup, the stem and the stored coordinate see one value per pixel.  The OLD coordinate is
// read from x_old and the new one written to x_new != x_old (the caller alternates two buffers: a stem tile's halo
// belongs to another block's segment, which may already have stored its result); the lookup blocks own the stores.
#include "stem7_body.h"

struct FrontArgs {
    CorrFeatArgs cf;
    Stem7Args s7;
    const float *planes; long planes_bs; int n_co;      // epilogue-3 planes of the head's first layer: [(co block) * 9 + tap][H][W]
    const float *head_bias;                             // bias of the head's second layer (output 0) or null
    const float *x_old; long x_old_bs;                  // coords1[:, 0] before the update
    float *x_new; long x_new_bs;                        // ... after it
    const float *x0; long x0_bs;                        // coords0[:, 0]
    float *flow; long flow_bs;                          // flow (B, Cin, H, W): channel 0 written, channels 1.. read by the stem
    int nb_stem;
};

// dkt_head_finish (conv_c8.hip) for one output and one pixel: the shifted planes in (co block, tap) order, then the bias
__device__ __forceinline__ float front_coord(const FrontArgs &f, int b, int oh, int ow, long p) {
    const int H = f.cf.H, W = f.cf.W1;
    const long HW = f.cf.HW;
    float s = 0.0f;
    for (int cb = 0; cb < f.n_co; ++cb) {
        const float *pl = f.planes + (long)b * f.planes_bs + (long)cb * 9 * HW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
            const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
            const float v = pl[(long)t * HW + (in ? (long)ih * W + iw : 0)];
            s = __fadd_rn(s, in ? v : 0.0f);
        }
    }
    s = __fadd_rn(s, f.head_bias ? f.head_bias[0] : 0.0f);
    return __fadd_rn(f.x_old[(long)b * f.x_old_bs + p], s);
}

template <int R>
__global__ __launch_bounds__(256) void motion_front_kernel(FrontArgs f) {
    const int b = blockIdx.y;
    if ((int)blockIdx.x < f.nb_stem) {
        const long HW = f.cf.HW;
        const float *fl = f.flow + (long)b * f.flow_bs;
        const float *x0 = f.x0 + (long)b * f.x0_bs;
        stem7_tile(f.s7, (int)(blockIdx.x / f.s7.n_co), (int)(blockIdx.x % f.s7.n_co), b, [&](int ih, int iw, long off, float (&v)[4]) {
            v[0] = __fsub_rn(front_coord(f, b, ih, iw, off), x0[off]);
#pragma unroll
            for (int c = 1; c < 4; ++c) v[c] = fl[(long)(c < f.s7.Cin ? c : 0) * HW + off];
        });
        return;
    }
    corr_feat64_body<R>(f.cf, (long)blockIdx.x - f.nb_stem, b, [&](int bb, long p, bool live, int g) -> float {
        const int oh = (int)(p / f.cf.W1), ow = (int)(p - (long)oh * f.cf.W1);
        const float nx = front_coord(f, bb, oh, ow, p);
        if (g == 0 && live) {
            f.x_new[(long)bb * f.x_new_bs + p] = nx;
            f.flow[(long)bb * f.flow_bs + p] = __fsub_rn(nx, f.x0[(long)bb * f.x0_bs + p]);
        }
        return nx;
    });
}

template <int R>
static int cf16_launch(CorrFeatArgs a, int B, hipStream_t st) {
    a.nseg = (a.W1 + 15) / 16;
    const long waves = (long)a.H * a.nseg;
    dim3 grid((unsigned)((waves + 3) / 4), (unsigned)B);
    switch ((a.Cout + 15) / 16) {
        case 1: hipLaunchKernelGGL((corr_feat16_kernel<R, 1>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((corr_feat16_kernel<R, 2>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((corr_feat16_kernel<R, 3>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((corr_feat16_kernel<R, 4>), grid, dim3(256), 0, st, a); break;
    }
    return dkt_launch_status();
}

static int corr_feat_impl(const float *const *skew, const float *coords_x, long coords_bstride,
                          const float *weight, const float *bias, float *out, long out_bstride,
                          float *tap, long tap_bstride, void *out_c8, long out_c8_bstride_bytes, int out_c8_ch0, float act_scale,
                          int B, int H, int W1, int W2, int L, int r, int Cout, int relu,
                          int device, void *stream) {
    if (!skew || !coords_x || !weight || (!out && !out_c8)) return DKT_E_NULL;
    if (out_c8 && (L != 4 || (out_c8_ch0 & 7) || !(act_scale > 0.0f))) return DKT_E_UNSUPPORTED;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || B > 65535 || Cout <= 0) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    if (Cout > 64 || !((L == 4 || L == 2 || L == 3) && (r == 4 || r == 3))) return DKT_E_UNSUPPORTED;
    if ((W2 >> (L - 1)) < 2) return DKT_E_UNSUPPORTED;     // a level of width 1: the general lookup handles its inf/NaN
    CorrFeatArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.skew.p[i] = i < L ? skew[i] : nullptr;
        if (i < L && !skew[i]) return DKT_E_NULL;
        const int wi = i < L ? (W2 >> i) : 0;
        a.inv_wm1[i] = wi > 1 ? (float)(1.0 / (double)(wi - 1)) : 0.0f;
    }
    a.coords_x = coords_x;
    a.coords_bstride = coords_bstride;
    a.w = weight;
    a.bias = bias;
    a.out = out;
    a.out_bstride = out_bstride;
    a.tap = tap;
    a.tap_bstride = tap_bstride;
    a.HW = (long)H * W1;
    a.H = H; a.W1 = W1; a.W2 = W2;
    a.pitch = (W1 + 31) & ~31;                 // dkt_corr1d_skew_pitch
    a.nseg = (W1 + 31) / 32;
    a.Cout = Cout;
    a.relu = relu ? 1 : 0;
    int Hp = 0, Wp = 0;
    dkt_act_c8_dims(H, W1, &Hp, &Wp);
    a.out_c8 = (char *)out_c8; a.out_c8_bs = out_c8_bstride_bytes; a.out_c8_plane = (long)Hp * Wp * 16;
    a.out_c8_Wp = Wp; a.out_c8_ch0 = out_c8_ch0; a.act_scale = act_scale;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    // four levels: the block form (64-pixel segments)
    if (L == 4) {
        const long blocks = (long)H * ((W1 + 63) / 64);
        if (blocks > 0x7fffffffL) return DKT_E_SHAPE;
        dim3 grid((unsigned)blocks, (unsigned)B);
        if (r == 4) hipLaunchKernelGGL(corr_feat64_kernel<4>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(corr_feat64_kernel<3>, grid, dim3(256), 0, st, a);
        return dkt_launch_status();
    }
    if (out_c8) return r == 4 ? cf16_launch<4>(a, B, st) : cf16_launch<3>(a, B, st);
    if (r == 4) {
        if (L == 4) return cf_launch<4, 4>(a, B, st);
        if (L == 3) return cf_launch<3, 4>(a, B, st);
        return cf_launch<2, 4>(a, B, st);
    }
    if (L == 4) return cf_launch<4, 3>(a, B, st);
    if (L == 3) return cf_launch<3, 3>(a, B, st);
    return cf_launch<2, 3>(a, B, st);
}

extern "C" int dkt_corr1d_lookup_conv1x1(const float *const *skew, const float *coords_x, long coords_bstride,
                                         const float *weight, const float *bias, float *out, long out_bstride,
                                         float *tap, long tap_bstride,
                                         int B, int H, int W1, int W2, int L, int r, int Cout, int relu,
                                         int device, void *stream) {
    if (!out) return DKT_E_NULL;
    return corr_feat_impl(skew, coords_x, coords_bstride, weight, bias, out, out_bstride, tap, tap_bstride, nullptr, 0, 0, 1.0f,
                          B, H, W1, W2, L, r, Cout, relu, device, stream);
}

extern "C" int dkt_corr1d_lookup_conv1x1_c8(const float *const *skew, const float *coords_x, long coords_bstride,
                                            const float *weight, const float *bias, void *out_c8, long out_c8_bstride_bytes,
                                            int out_c8_ch0, float act_scale, int B, int H, int W1, int W2, int L, int r, int Cout,
                                            int relu, int device, void *stream) {
    if (!out_c8) return DKT_E_NULL;
    return corr_feat_impl(skew, coords_x, coords_bstride, weight, bias, nullptr, 0, nullptr, 0, out_c8, out_c8_bstride_bytes,
                          out_c8_ch0, act_scale, B, H, W1, W2, L, r, Cout, relu, device, stream);
}

// ---- the fused front (motion_front_kernel above)
extern "C" int dkt_motion_front_c8(const dkt_motion_front_desc *d, int device, void *stream) {
    if (!d) return DKT_E_NULL;
    if (!d->skew || !d->planes || !d->x_old || !d->x_new || !d->x0 || !d->flow || !d->w_cor || !d->cor_c8 || !d->stem_w_hi ||
        !d->stem_w_lo || !d->flo_c8)
        return DKT_E_NULL;
    if (d->x_new == d->x_old) return DKT_E_SHAPE;          // a stem tile's halo would read another block's result
    const int B = d->B, H = d->H, W1 = d->W1, W2 = d->W2, L = d->L, r = d->r;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || B > 65535 || d->n_co < 1) return DKT_E_SHAPE;
    if (L != 4 || (r != 4 && r != 3) || d->cor_channels <= 0 || d->cor_channels > 64 || (W2 >> (L - 1)) < 2) return DKT_E_UNSUPPORTED;
    if (d->stem_cin < 1 || d->stem_cin > 4 || d->stem_cout <= 0) return DKT_E_SHAPE;
    if ((d->cor_c8_ch0 & 7) || (d->flo_c8_ch0 & 7) || !(d->cor_act_scale > 0.0f) || !(d->flo_act_scale > 0.0f) ||
        !(d->stem_in_scale > 0.0f) || !(d->stem_out_scale > 0.0f))
        return DKT_E_SHAPE;
    FrontArgs f;
    CorrFeatArgs &a = f.cf;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.skew.p[i] = i < L ? d->skew[i] : nullptr;
        if (i < L && !d->skew[i]) return DKT_E_NULL;
        const int wi = i < L ? (W2 >> i) : 0;
        a.inv_wm1[i] = wi > 1 ? (float)(1.0 / (double)(wi - 1)) : 0.0f;
    }
    a.coords_x = nullptr; a.coords_bstride = 0;
    a.w = d->w_cor; a.bias = d->b_cor;
    a.out = nullptr; a.out_bstride = 0; a.tap = nullptr; a.tap_bstride = 0;
    a.HW = (long)H * W1;
    a.H = H; a.W1 = W1; a.W2 = W2;
    a.pitch = (W1 + 31) & ~31;
    a.nseg = (W1 + 31) / 32;
    a.Cout = d->cor_channels;
    a.relu = 1;
    int Hp = 0, Wp = 0;
    dkt_act_c8_dims(H, W1, &Hp, &Wp);
    a.out_c8 = (char *)d->cor_c8; a.out_c8_bs = d->cor_c8_bstride_bytes; a.out_c8_plane = (long)Hp * Wp * 16;
    a.out_c8_Wp = Wp; a.out_c8_ch0 = d->cor_c8_ch0; a.act_scale = d->cor_act_scale;
    Stem7Args &s = f.s7;
    s.x = nullptr; s.x_bs = 0;
    s.whi = (const _Float16 *)d->stem_w_hi; s.wlo = (const _Float16 *)d->stem_w_lo;
    s.bias = d->stem_bias; s.out_scale = d->stem_out_scale; s.in_scale = d->stem_in_scale;
    s.y = nullptr; s.y_bs = 0;
    s.Cin = d->stem_cin; s.Cout = d->stem_cout; s.CoutPad = (d->stem_cout + 63) & ~63;
    s.H = H; s.W = W1;
    s.tiles_w = (W1 + 31) / 32;
    s.tiles_xy = s.tiles_w * ((H + 3) / 4);
    s.n_co = s.CoutPad / 64;
    s.relu = 1;
    s.y_c8 = (char *)d->flo_c8; s.y_c8_bs = d->flo_c8_bstride_bytes; s.y_c8_plane = (long)Hp * Wp * 16; s.y_c8_Wp = Wp;
    s.y_c8_ch0 = d->flo_c8_ch0; s.act_scale = d->flo_act_scale;
    f.planes = d->planes; f.planes_bs = d->planes_bstride; f.n_co = d->n_co;
    f.head_bias = d->head_bias;
    f.x_old = d->x_old; f.x_old_bs = d->x_old_bstride;
    f.x_new = d->x_new; f.x_new_bs = d->x_new_bstride;
    f.x0 = d->x0; f.x0_bs = d->x0_bstride;
    f.flow = d->flow; f.flow_bs = d->flow_bstride;
    const long nb_stem = (long)s.tiles_xy * s.n_co, nb_look = (long)H * ((W1 + 63) / 64);
    if (nb_stem + nb_look > 0x7fffffffL) return DKT_E_SHAPE;
    f.nb_stem = (int)nb_stem;
    DKT_ENTER(device);
    dim3 grid((unsigned)(nb_stem + nb_look), (unsigned)B);
    if (r == 4) hipLaunchKernelGGL(motion_front_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, f);
    else hipLaunchKernelGGL(motion_front_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, f);
    return dkt_launch_status();
}
