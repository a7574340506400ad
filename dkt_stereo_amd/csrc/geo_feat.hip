// geo_feat.hip -- IGEV's geometry-encoding lookup (meta_arch/igev_stereo/geometry.py:29-69) fused with the motion encoder's
// 1x1 layer that consumes it (convc1, igev_stereo/update.py:78,86: L*(2r+1)*(C+1) -> 64 channels, ReLU), the IGEV counterpart
// of corr_feat64_kernel (corr_feat.hip): the 162-channel lookup tensor (37 MB per iteration at 184 x 312) is never written.
//
// A block of eight waves owns one 64-pixel row segment (lane = pixel).
//   phase 1: wave g samples level g >> 2, geometry channels [2 (g & 3), 2 (g & 3) + 2) -- per channel the 2r + 2 planes of
//            the pixel's disparity window, plane stride H*W, contiguous across the wave when the disparity is locally
//            smooth -- and, for g & 3 == 3, the level's init-correlation row; the values go to LDS as [k][pixel] (pitch
//            65), k in the reference's channel order (level, channel, tap | level, C, tap).  Arithmetic =
//            geo_lookup_kernel's (geo.hip), i.e. bit-identical to the reference sampler; the optional `tap` output is that
//            lookup itself.  (Four waves with twice the samples each: 48 us against 46; the weight
//            fragments fetched ahead of the sampling: 46 -> 37.6 us.)
//   phase 2: wave (m = g & 3, half = g >> 2) multiplies the K x 32 half of the sample matrix with output channels
//            16m .. 16m+15 on v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, k ascending), adds bias, applies ReLU and stores
//            fp32 NCHW or C8S (conv_c8.hip).
// Supported: L = 2, C = 8, r = 4 (IGEV's configuration), Cout <= 64; anything else: DKT_E_UNSUPPORTED (callers then run
// dkt_geo_lookup + the 1x1 convolution).
#include "dkt_common.h"

#define GF_PITCH 65

struct GeoFeatArgs {
    const float *geo[2];       // level i: (B, C, D >> i, H, W)
    const float *init[2];      // level i: (B*H*W, W2 >> i)
    const float *disp; long disp_bs;
    const float *coords;       // (B, H, W) dense
    const float *w;            // [K][Cout], k-major
    const float *bias;
    float *out; long out_bs;
    char *out_c8; long out_c8_bs, out_c8_plane; int out_c8_Wp, out_c8_ch0;
    float act_scale;
    float *tap; long tap_bs;
    long HW;
    int H, W, D, W2, Cout, relu;
    float inv_d[2], inv_w[2];
};

__device__ __forceinline__ int gf_clamp_idx(float fl, int W) {
    return (int)fminf(fmaxf(fl, -2.0f), (float)W + 1.0f);
}

template <int R, int C>
__global__ __launch_bounds__(512) void geo_feat64_kernel(GeoFeatArgs a) {
    constexpr int K = 2 * R + 1;
    constexpr int PER = K * (C + 1);              // channels per level
    constexpr int KT = 2 * PER;                   // 162
    constexpr int NS = (KT + 3) / 4;              // k steps of 4
    static_assert(C % 4 == 0, "two channel quarters per level");
    __shared__ float vs[NS * 4 * GF_PITCH];
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nseg = (a.W + 63) / 64;
    const long hrow = blockIdx.x / nseg;
    const int seg0 = (int)(blockIdx.x - hrow * nseg) * 64;
    const int b = blockIdx.y;
    // phase 2's weight fragments first (wave: channels 16 (g & 3) .., k = 4 step + q): inside the k loop each was an L2 round
    // trip on the critical path behind the barrier; issued here their latency hides under the sampling phase
    float Aw[NS];
    {
        const int j = lane & 15, q = lane >> 4;
        const int co = 16 * (g & 3) + j;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int kk = 4 * sidx + q;
            const bool ok = co < a.Cout && kk < KT;
            const float wv = a.w[ok ? (long)kk * a.Cout + co : 0];
            Aw[sidx] = ok ? wv : 0.0f;
        }
    }
    {
        const int w1 = seg0 + lane;
        const bool live = w1 < a.W;
        const long p = hrow * a.W + (live ? w1 : a.W - 1);
        const int lv = g >> 2, part = g & 3;
        const float inv = (float)(1 << lv);
        const float dl = __fdiv_rn(a.disp[(size_t)b * a.disp_bs + p], inv);
        float *tp = a.tap ? a.tap + (size_t)b * a.tap_bs + (size_t)lv * PER * a.HW + p : nullptr;
        // ---- geometry volume: taps along D, plane stride HW (geo_lookup_kernel, geo.hip)
        {
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
            const int i0 = gf_clamp_idx(taps[0].fl, di);
            bool regular = true;
#pragma unroll
            for (int k = 0; k < K; ++k) regular = regular && gf_clamp_idx(taps[k].fl, di) == i0 + k;
#pragma unroll
            for (int cc = 0; cc < C / 4; ++cc) {
                const int c = part * (C / 4) + cc;
                const float *base = a.geo[lv] + ((size_t)b * C + c) * (size_t)di * a.HW + p;
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
                        const int ik = gf_clamp_idx(taps[k].fl, di);
                        if (ik != i0 + k) {
                            v0 = (ik >= 0 && ik < di) ? base[(size_t)ik * a.HW] : 0.0f;
                            v1 = (ik + 1 >= 0 && ik + 1 < di) ? base[(size_t)(ik + 1) * a.HW] : 0.0f;
                        }
                    }
                    const float v = dkt_blend(v0, v1, taps[k]);
                    vs[(lv * PER + c * K + k) * GF_PITCH + lane] = v;
                    if (tp && live) tp[(size_t)(c * K + k) * a.HW] = v;
                }
            }
        }
        // ---- init-correlation row: x = (coords / 2^i - disp / 2^i) + dx   (geometry.py:50)
        if (part == 3) {
            const int wi = a.W2 >> lv;
            const float wm1 = (float)(wi - 1), hwm1 = __fdiv_rn(wm1, 2.0f);
            const size_t n = (size_t)b * a.HW + p;
            const float *row = a.init[lv] + n * (size_t)wi;
            const float cl = __fdiv_rn(a.coords[n], inv);
            const float xc = __fsub_rn(cl, dl);
            DktTap taps[K];
            if (wi > 1) {
#pragma unroll
                for (int k = 0; k < K; ++k) taps[k] = dkt_tap_rcp(__fadd_rn(xc, (float)(k - R)), wm1, a.inv_w[lv], hwm1);
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) taps[k] = dkt_tap(__fadd_rn(xc, (float)(k - R)), wm1, hwm1);
            }
            const int i0 = gf_clamp_idx(taps[0].fl, wi);
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
                const int ik = gf_clamp_idx(taps[k].fl, wi);
                float v0 = win[k], v1 = win[k + 1];
                if (ik != i0 + k) {
                    v0 = (ik >= 0 && ik < wi) ? row[ik] : 0.0f;
                    v1 = (ik + 1 >= 0 && ik + 1 < wi) ? row[ik + 1] : 0.0f;
                }
                const float v = dkt_blend(v0, v1, taps[k]);
                vs[(lv * PER + C * K + k) * GF_PITCH + lane] = v;
                if (tp && live) tp[(size_t)(C * K + k) * a.HW] = v;
            }
        } else if (g == 0) {
            // rows KT .. 4 NS - 1 pad the last k step: zeros (their weights are zero too; garbage could be NaN)
#pragma unroll
            for (int k = KT; k < NS * 4; ++k) vs[k * GF_PITCH + lane] = 0.0f;
        }
    }
    __syncthreads();
    // ---- phase 2: wave m = g: channels 16m .. 16m+15 (corr_feat64_kernel's)
    const int m = g & 3, th = g >> 2;         // 16-channel tile, pixel half (n tiles 2 th, 2 th + 1)
    if (16 * m >= a.Cout) return;             // wave-uniform
    const int j = lane & 15, q = lane >> 4;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
        const int kk = 4 * sidx + q;
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[sidx], vs[kk * GF_PITCH + 16 * (2 * th + t) + j], acc[t], 0, 0, 0);
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c2 = 16 * m + 4 * q + r;
        bv[r] = a.bias ? a.bias[c2 < a.Cout ? c2 : a.Cout - 1] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int w1 = seg0 + 16 * (2 * th + t) + j;
        const bool live = w1 < a.W;
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
            float *ob = a.out + (size_t)b * a.out_bs + hrow * a.W + w1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c2 = 16 * m + 4 * q + r;
                if (c2 < a.Cout) ob[(size_t)c2 * a.HW] = y[r];
            }
        }
    }
}

extern "C" int dkt_geo_lookup_conv1x1(const float *const *geo_pyr, const float *const *init_pyr,
                                      const float *disp, long disp_bstride, const float *coords,
                                      const float *weight_t, const float *bias,
                                      float *out, long out_bstride, void *out_c8, long out_c8_bstride_bytes, int out_c8_ch0,
                                      float act_scale, float *tap, long tap_bstride,
                                      int B, int C, int D, int H, int W, int W2, int L, int r, int Cout, int relu,
                                      int device, void *stream) {
    if (!geo_pyr || !init_pyr || !disp || !coords || !weight_t || (!out && !out_c8)) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || W2 <= 0 || Cout <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L != 2 || C != 8 || r != 4 || Cout > 64 || (D >> 1) == 0 || (W2 >> 1) == 0) return DKT_E_UNSUPPORTED;
    if ((out_c8_ch0 & 7) || !(act_scale > 0.0f)) return DKT_E_SHAPE;
    GeoFeatArgs a;
    for (int i = 0; i < 2; ++i) {
        if (!geo_pyr[i] || !init_pyr[i]) return DKT_E_NULL;
        a.geo[i] = geo_pyr[i];
        a.init[i] = init_pyr[i];
        const int di = D >> i, wi = W2 >> i;
        a.inv_d[i] = di > 1 ? (float)(1.0 / (double)(di - 1)) : 0.0f;
        a.inv_w[i] = wi > 1 ? (float)(1.0 / (double)(wi - 1)) : 0.0f;
    }
    a.disp = disp; a.disp_bs = disp_bstride; a.coords = coords;
    a.w = weight_t; a.bias = bias;
    a.out = out; a.out_bs = out_bstride;
    int Hp = 0, Wp = 0;
    dkt_act_c8_dims(H, W, &Hp, &Wp);
    a.out_c8 = (char *)out_c8; a.out_c8_bs = out_c8_bstride_bytes; a.out_c8_plane = (long)Hp * Wp * 16; a.out_c8_Wp = Wp;
    a.out_c8_ch0 = out_c8_ch0; a.act_scale = act_scale;
    a.tap = tap; a.tap_bs = tap_bstride;
    a.HW = (long)H * W; a.H = H; a.W = W; a.D = D; a.W2 = W2; a.Cout = Cout; a.relu = relu ? 1 : 0;
    const long blocks = (long)H * ((W + 63) / 64);
    if (blocks > 0x7fffffffL) return DKT_E_SHAPE;
    DKT_ENTER(device);
    hipLaunchKernelGGL((geo_feat64_kernel<4, 8>), dim3((unsigned)blocks, (unsigned)B), dim3(512), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}
