// stem7.hip -- the 7x7 stems (Cin <= 4) on the fp16 matrix cores with split operands.
//   convf1 2->64 / convd1 1->64 in the motion encoders (core/update.py:75,
//   meta_arch/igev_stereo/update.py:81) and conv1 3->64 of both encoders (core/extractor.py:136).
// A 32-channel chunk of the general kernel (conv2d.hip) would be 87-97 % padding here, so the GEMM
// K axis is laid out over (row tap dy, column tap dx, channel) instead:
//     k16 step = (dy, half) , k inside the step = (dx - 4*half)*4 + ci      (dx padded 7 -> 8, ci -> 4)
// i.e. 14 steps of 16.  The whole input patch of a tile -- (TR+6) x 39 pixels x 4 channels, fp16 hi and
// lo -- is staged ONCE ([pixel][4 channels] = 8 bytes per pixel and plane), so a B fragment of lane
// (pixel li, k group kg) is the two pixels (col + 4*half + 2*kg, +1) of row dy: one 16-byte run, read as
// two ds_read_b64.  Same three passes (w_hi*x_hi + w_lo*x_hi + w_hi*x_lo) and power-of-two weight scale
// as conv2d.hip: fp32-class (1e-6 relative to an fp64 convolution).  The MFMA work is trivial (14 steps);
// the kernel streams its output (64 channels x tile) and is HBM-write bound.
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

// block = 4 waves = 4 output rows x 32 columns x 64 output channels; grid.x = spatial tile * n_co + co block
__global__ __launch_bounds__(256) void conv2d_stem7_kernel(Stem7Args a) {
    constexpr int TR = 4, PR = TR + 6, PC = 40;          // 32 + 6 halo + 1 (dx = 7 pad) + 1 (even pitch)
    __shared__ __attribute__((aligned(16))) _Float16 phi[PR * PC * 4], plo[PR * PC * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kg = lane >> 5;
    const int t = blockIdx.x / a.n_co, cb = blockIdx.x % a.n_co;
    const int b = blockIdx.y;
    const int w0 = (t % a.tiles_w) * 32, h0 = (t / a.tiles_w) * TR;
    const long HW = (long)a.H * a.W;
    const float *xb = a.x + (long)b * a.x_bs;
    // ---- stage the patch: thread = patch pixel, all (<= 4) channels
    for (int pp = tid; pp < PR * PC; pp += 256) {
        const int pr = pp / PC, pc = pp - pr * PC;
        const int ih = h0 - 3 + pr, iw = w0 - 3 + pc;
        const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        const long off = ok ? (long)ih * a.W + iw : 0;
        s7_f16x4 hv, lv;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = (ok && c < a.Cin) ? xb[(long)(c < a.Cin ? c : 0) * HW + off] : 0.0f;
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

// (Cout, Cin, 7, 7) fp32 -> hi/lo fp16 [step = 2*dy + half][coPad][16], k = (dx - 4*half)*4 + ci
__global__ __launch_bounds__(256) void conv2d_stem7_pack_kernel(const float *w, _Float16 *whi, _Float16 *wlo,
                                                                int Cout, int Cin, int CoutPad, float scale) {
    const long total = (long)S7_STEPS * CoutPad * 16;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % 16);
    const int co = (int)((i / 16) % CoutPad);
    const int s = (int)(i / (16L * CoutPad));
    const int dy = s >> 1, dx = 4 * (s & 1) + (k >> 2), ci = k & 3;
    float v = 0.0f;
    if (co < Cout && ci < Cin && dx < 7) v = w[(((long)co * Cin + ci) * 7 + dy) * 7 + dx] * scale;
    const _Float16 h = (_Float16)v;
    whi[i] = h;
    wlo[i] = (_Float16)(v - (float)h);
}

extern "C" long dkt_conv2d_stem7_packed_elems(int Cout) {
    if (Cout <= 0) return DKT_E_SHAPE;
    return (long)S7_STEPS * ((Cout + 63) & ~63) * 16;
}

extern "C" int dkt_conv2d_stem7_pack(const float *w, int Cout, int Cin, float scale, void *w_hi, void *w_lo,
                                     int device, void *stream) {
    if (!w || !w_hi || !w_lo) return DKT_E_NULL;
    if (Cout <= 0 || Cin <= 0 || Cin > 4 || !(scale > 0.0f)) return DKT_E_SHAPE;
    DKT_ENTER(device);
    const int pad = (Cout + 63) & ~63;
    const long total = (long)S7_STEPS * pad * 16;
    hipLaunchKernelGGL(conv2d_stem7_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, (_Float16 *)w_hi, (_Float16 *)w_lo, Cout, Cin, pad, scale);
    return dkt_launch_status();
}

static int stem7_impl(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                      const float *bias, float out_scale, float in_scale, float *y, long y_bstride,
                      void *y_c8, long y_c8_bstride_bytes, int y_c8_ch0, float act_scale,
                      int B, int Cin, int Cout, int H, int W, int relu, int device, void *stream) {
    if (!x || !w_hi || !w_lo || (!y && !y_c8)) return DKT_E_NULL;
    if (B <= 0 || Cin <= 0 || Cin > 4 || Cout <= 0 || H <= 0 || W <= 0 || B > 65535) return DKT_E_SHAPE;
    if (!(in_scale > 0.0f) || !(out_scale > 0.0f) || !(act_scale > 0.0f) || (y_c8_ch0 & 7)) return DKT_E_SHAPE;
    Stem7Args a;
    a.x = x; a.x_bs = x_bstride;
    a.whi = (const _Float16 *)w_hi; a.wlo = (const _Float16 *)w_lo;
    a.bias = bias; a.out_scale = out_scale; a.in_scale = in_scale;
    a.y = y; a.y_bs = y_bstride;
    a.Cin = Cin; a.Cout = Cout; a.CoutPad = (Cout + 63) & ~63;
    a.H = H; a.W = W;
    a.tiles_w = (W + 31) / 32;
    a.tiles_xy = a.tiles_w * ((H + 3) / 4);
    a.n_co = a.CoutPad / 64;
    a.relu = relu ? 1 : 0;
    int Hp = 0, Wp = 0;
    dkt_act_c8_dims(H, W, &Hp, &Wp);
    a.y_c8 = (char *)y_c8; a.y_c8_bs = y_c8_bstride_bytes; a.y_c8_plane = (long)Hp * Wp * 16; a.y_c8_Wp = Wp; a.y_c8_ch0 = y_c8_ch0;
    a.act_scale = act_scale;
    const long blocks = (long)a.tiles_xy * a.n_co;
    if (blocks > 0x7fffffffL) return DKT_E_SHAPE;
    DKT_ENTER(device);
    hipLaunchKernelGGL(conv2d_stem7_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}

extern "C" int dkt_conv2d_stem7(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                                const float *bias, float out_scale, float in_scale, float *y, long y_bstride,
                                int B, int Cin, int Cout, int H, int W, int relu, int device, void *stream) {
    if (!y) return DKT_E_NULL;
    return stem7_impl(x, x_bstride, w_hi, w_lo, bias, out_scale, in_scale, y, y_bstride, nullptr, 0, 0, 1.0f,
                      B, Cin, Cout, H, W, relu, device, stream);
}

extern "C" int dkt_conv2d_stem7_c8(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                                   const float *bias, float out_scale, float in_scale, void *y_c8, long y_c8_bstride_bytes,
                                   int y_c8_ch0, float act_scale, int B, int Cin, int Cout, int H, int W, int relu,
                                   int device, void *stream) {
    if (!y_c8) return DKT_E_NULL;
    return stem7_impl(x, x_bstride, w_hi, w_lo, bias, out_scale, in_scale, nullptr, 0, y_c8, y_c8_bstride_bytes, y_c8_ch0,
                      act_scale, B, Cin, Cout, H, W, relu, device, stream);
}

extern "C" int dkt_conv2d_stem7_dual(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                                     const float *bias, float out_scale, float in_scale, float *y, long y_bstride,
                                     void *y_c8, long y_c8_bstride_bytes, int y_c8_ch0, float act_scale,
                                     int B, int Cin, int Cout, int H, int W, int relu, int device, void *stream) {
    if (!y || !y_c8) return DKT_E_NULL;
    return stem7_impl(x, x_bstride, w_hi, w_lo, bias, out_scale, in_scale, y, y_bstride, y_c8, y_c8_bstride_bytes, y_c8_ch0,
                      act_scale, B, Cin, Cout, H, W, relu, device, stream);
}
