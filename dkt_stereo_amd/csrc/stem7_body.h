// stem7_body.h -- the tile routine of the 7x7 stems (stem7.hip), shared with the fused motion-encoder front (corr_feat.hip).
#pragma once
#include "dkt_common.h"

typedef _Float16 s7_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s7_f16x4 __attribute__((ext_vector_type(4)));
typedef float s7_f32x16 __attribute__((ext_vector_type(16)));

#define S7_STEPS 14

struct Stem7Args {
    const float *x;
    long x_bs;
    const _Float16 *whi, *wlo;      // [step][coPad][16]
    const float *bias;
    float out_scale, in_scale;
    float *y;
    long y_bs;
    int Cin, Cout, CoutPad, H, W, tiles_w, tiles_xy, n_co;
    int relu;
    // optional C8S destination (conv_c8.hip's operand layout) instead of / besides the fp32 NCHW one
    char *y_c8;
    long y_c8_bs, y_c8_plane;
    int y_c8_Wp, y_c8_ch0;
    float act_scale;
};

// block = 4 waves = 4 output rows x 32 columns x 64 output channels: spatial tile t, channel block cb, batch item b.
// load(ih, iw, off, v[4]): the input channels c < Cin of the in-image pixel (ih, iw) (off = ih * W + iw) into v[c], before
// the input scale; called for every patch position (out-of-image ones with pixel (0, 0)): it must not branch on them.
template <class LD>
__device__ __forceinline__ void stem7_tile(const Stem7Args &a, int t, int cb, int b, LD load) {
    constexpr int TR = 4, PR = TR + 6, PC = 40;          // 32 + 6 halo + 1 (dx = 7 pad) + 1 (even pitch)
    __shared__ __attribute__((aligned(16))) _Float16 phi[PR * PC * 4], plo[PR * PC * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kg = lane >> 5;
    const int w0 = (t % a.tiles_w) * 32, h0 = (t / a.tiles_w) * TR;
    const long HW = (long)a.H * a.W;
    // ---- stage the patch: thread = patch pixel, all (<= 4) channels
    for (int pp = tid; pp < PR * PC; pp += 256) {
        const int pr = pp / PC, pc = pp - pr * PC;
        const int ih = h0 - 3 + pr, iw = w0 - 3 + pc;
        const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        const long off = ok ? (long)ih * a.W + iw : 0;
        s7_f16x4 hv, lv;
        float in[4];
        load(ok ? ih : 0, ok ? iw : 0, off, in);      // unconditional (valid addresses), selected below
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = (ok && c < a.Cin) ? in[c] : 0.0f;
            v *= a.in_scale;       // power of two; out-of-range / non-finite inputs give non-finite outputs
            const _Float16 h = (_Float16)v;
            hv[c] = h;
            lv[c] = (_Float16)(v - (float)h);
        }
        *(s7_f16x4 *)(phi + pp * 4) = hv;
        *(s7_f16x4 *)(plo + pp * 4) = lv;
    }
    __syncthreads();
    // ---- 14 (dy, half) steps; wave = output row `wave`, 2 m-fragments (64 channels), 1 n-fragment
    s7_f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    const int co0 = cb * 64;
    const _Float16 *wh = a.whi + ((long)(co0 + li)) * 16 + kg * 8;
    const _Float16 *wl = a.wlo + ((long)(co0 + li)) * 16 + kg * 8;
    const long wstep = (long)a.CoutPad * 16;
#pragma unroll
    for (int s = 0; s < S7_STEPS; ++s) {
        const int dy = s >> 1, half = s & 1;
        s7_f16x8 Ah[2], Al[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            Ah[m] = *(const s7_f16x8 *)(wh + s * wstep + m * 32 * 16);
            Al[m] = *(const s7_f16x8 *)(wl + s * wstep + m * 32 * 16);
        }
        const int pp = (wave + dy) * PC + li + 4 * half + 2 * kg;
        union { s7_f16x8 v; s7_f16x4 h[2]; } Bh, Bl;
        Bh.h[0] = *(const s7_f16x4 *)(phi + pp * 4);
        Bh.h[1] = *(const s7_f16x4 *)(phi + pp * 4 + 4);
        Bl.h[0] = *(const s7_f16x4 *)(plo + pp * 4);
        Bl.h[1] = *(const s7_f16x4 *)(plo + pp * 4 + 4);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m], Bh.v, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[m], Bh.v, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m], Bl.v, acc[m], 0, 0, 0);
        }
    }
    // ---- epilogue: un-scale, bias, ReLU; lane li = column, kg picks the channel sub-block
    const int oh = h0 + wave, ow = w0 + li;
    const bool inside = oh < a.H && ow < a.W;
    float *yo = a.y ? a.y + (long)b * a.y_bs + (long)(inside ? oh : 0) * a.W + (inside ? ow : 0) : nullptr;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + 4 * kg + (r & 3) + 8 * (r >> 2);
            float t = acc[m][r] * a.out_scale + (a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f);
            if (a.relu) t = dkt_relu(t);
            v[r] = co < a.Cout ? t : 0.0f;
            if (yo && inside && co < a.Cout) yo[(long)co * HW] = v[r];
        }
        if (a.y_c8) {
            // C8S: groups of 8 consecutive channels per 16 bytes -- pairs of the lane's 4-channel groups are completed
            // by exchanging halves with lane ^ 32 (as conv_c8.hip's epilogue)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned ha[2], la[2], hb[2], lb[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const float x0 = v[8 * jp + 2 * d] * a.act_scale, x1 = v[8 * jp + 2 * d + 1] * a.act_scale;
                    const float y0 = v[8 * jp + 4 + 2 * d] * a.act_scale, y1 = v[8 * jp + 4 + 2 * d + 1] * a.act_scale;
                    const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1, b0 = (_Float16)y0, b1 = (_Float16)y1;
                    union { _Float16 h[2]; unsigned u; } t0, t1, t2, t3;
                    t0.h[0] = a0; t0.h[1] = a1;
                    t1.h[0] = (_Float16)(x0 - (float)a0); t1.h[1] = (_Float16)(x1 - (float)a1);
                    t2.h[0] = b0; t2.h[1] = b1;
                    t3.h[0] = (_Float16)(y0 - (float)b0); t3.h[1] = (_Float16)(y1 - (float)b1);
                    ha[d] = t0.u; la[d] = t1.u; hb[d] = t2.u; lb[d] = t3.u;
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    auto r = __builtin_amdgcn_permlane32_swap(ha[d], hb[d], false, false);
                    ha[d] = r[0]; hb[d] = r[1];
                    auto q = __builtin_amdgcn_permlane32_swap(la[d], lb[d], false, false);
                    la[d] = q[0]; lb[d] = q[1];
                }
                if (inside) {
                    const int g = ((a.y_c8_ch0 + co0 + m * 32) >> 3) + 2 * jp + kg;
                    char *p = a.y_c8 + (long)b * a.y_c8_bs + (long)g * 2 * a.y_c8_plane + ((long)(oh + 1) * a.y_c8_Wp + (ow + 1)) * 16;
                    *(uint4 *)p = make_uint4(ha[0], ha[1], hb[0], hb[1]);
                    *(uint4 *)(p + a.y_c8_plane) = make_uint4(la[0], la[1], lb[0], lb[1]);
                }
            }
        }
    }
}
