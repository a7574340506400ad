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

#include "stem7_body.h"

// grid.x = spatial tile * n_co + co block, grid.y = batch
__global__ __launch_bounds__(256) void conv2d_stem7_kernel(Stem7Args a) {
    const int b = blockIdx.y;
    const float *xb = a.x + (long)b * a.x_bs;
    const long HW = (long)a.H * a.W;
    stem7_tile(a, (int)(blockIdx.x / a.n_co), (int)(blockIdx.x % a.n_co), b, [&](int, int, long off, float (&v)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = xb[(long)(c < a.Cin ? c : 0) * HW + off];
    });
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
