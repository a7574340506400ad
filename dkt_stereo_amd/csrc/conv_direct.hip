// conv_direct.hip -- direct fp32 convolution for the two extreme layer shapes of the update
// block that a 64-channel-wide MFMA tile wastes:
//   * very few OUTPUT channels: flow_head.conv2 256->2 / disp_head.conv2 256->1, 3x3
//     (core/update.py:10, igev_stereo/update.py:20) -- the MFMA kernel pads 2 -> 64 channels;
//   * very few INPUT channels: convf1 2->64 / convd1 1->64, 7x7 (core/update.py:75,
//     igev_stereo/update.py:81) -- K = 2*49 would be padded to 49 chunks of 32.
// Exact fp32 FMAs on the VALU (no operand splitting): one thread per pixel accumulates TO
// output channels; the input patch of a channel chunk is staged in LDS; weights are read
// with wave-uniform addresses so they live in SGPRs / the scalar cache.  Optional fused ReLU.
#include "dkt_common.h"

struct DirectArgs {
    const float *x;
    long x_bs;
    const float *w;      // (Cout, Cin, KS, KS)
    const float *bias;
    float *y;
    long y_bs;
    int Cin, Cout, H, W, tiles_w;
    int relu;
};

template <int KS, int TO, int CC>      // CC: input channels staged per round
__global__ __launch_bounds__(256) void conv2d_direct_kernel(DirectArgs a) {
    constexpr int HALO = KS / 2;
    constexpr int TR = 8, TC = 32;
    constexpr int PR = TR + 2 * HALO, PC = TC + 2 * HALO;
    constexpr int PCP = PC | 1;                         // odd pitch: conflict-free row-shifted reads
    constexpr int TAPS = KS * KS;
    __shared__ float patch[CC][PR][PCP];
    // weights of the staged channels, [c][tap][TO]: one broadcast ds_read_b64/b128 per (c, tap)
    __shared__ __attribute__((aligned(16))) float wl[CC][TAPS][TO];
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const int w0 = (blockIdx.x % a.tiles_w) * TC, h0 = (blockIdx.x / a.tiles_w) * TR;
    const int co0 = blockIdx.y * TO;
    const int b = blockIdx.z;
    const long HW = (long)a.H * a.W;
    const float *xb = a.x + (long)b * a.x_bs;
    float acc[TO];
#pragma unroll
    for (int o = 0; o < TO; ++o) acc[o] = 0.0f;

    for (int c0 = 0; c0 < a.Cin; c0 += CC) {
        __syncthreads();
        for (int i = tid; i < CC * PR * PC; i += 256) {
            const int c = i / (PR * PC), r = (i / PC) % PR, q = i % PC;
            const int ih = h0 - HALO + r, iw = w0 - HALO + q;
            float v = 0.0f;
            if (c0 + c < a.Cin && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                v = xb[(long)(c0 + c) * HW + (long)ih * a.W + iw];
            patch[c][r][q] = v;
        }
        for (int i = tid; i < CC * TAPS * TO; i += 256) {
            const int o = i % TO, t = (i / TO) % TAPS, c = i / (TO * TAPS);
            float v = 0.0f;
            if (c0 + c < a.Cin && co0 + o < a.Cout) v = a.w[((long)(co0 + o) * a.Cin + (c0 + c)) * TAPS + t];
            wl[c][t][o] = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int c = 0; c < CC; ++c) {
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const float v = patch[c][ty + dy][tx + dx];
#pragma unroll
                    for (int o = 0; o < TO; ++o) acc[o] = __fmaf_rn(v, wl[c][dy * KS + dx][o], acc[o]);
                }
        }
    }
    const int oh = h0 + ty, ow = w0 + tx;
    if (oh < a.H && ow < a.W) {
        float *yb = a.y + (long)b * a.y_bs + (long)oh * a.W + ow;
#pragma unroll
        for (int o = 0; o < TO; ++o)
            if (co0 + o < a.Cout) {
                float v = acc[o] + (a.bias ? a.bias[co0 + o] : 0.0f);
                if (a.relu) v = fmaxf(v, 0.0f);
                yb[(long)(co0 + o) * HW] = v;
            }
    }
}

template <int KS, int TO, int CC>
static int launch_direct(const DirectArgs &a, int B, hipStream_t st) {
    const int tiles_h = (a.H + 7) / 8;
    dim3 grid((unsigned)(a.tiles_w * tiles_h), (unsigned)((a.Cout + TO - 1) / TO), (unsigned)B);
    hipLaunchKernelGGL((conv2d_direct_kernel<KS, TO, CC>), grid, dim3(256), 0, st, a);
    return dkt_launch_status();
}

extern "C" int dkt_conv2d_direct(const float *x, long x_bstride, const float *w, const float *bias,
                                 float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                                 int KH, int KW, int relu, int device, void *stream) {
    if (!x || !w || !y) return DKT_E_NULL;
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || B > 65535) return DKT_E_SHAPE;
    if (KH != KW || (KH != 3 && KH != 7)) return DKT_E_UNSUPPORTED;
    DirectArgs a;
    a.x = x; a.x_bs = x_bstride; a.w = w; a.bias = bias; a.y = y; a.y_bs = y_bstride;
    a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.tiles_w = (W + 31) / 32;
    a.relu = relu ? 1 : 0;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    if (KH == 3) {
        if (Cout > 4) return DKT_E_UNSUPPORTED;       // wide layers belong to dkt_conv2d_f16s
        if (Cout <= 2) return launch_direct<3, 2, 16>(a, B, st);
        return launch_direct<3, 4, 16>(a, B, st);
    }
    if (Cin > 4) return DKT_E_UNSUPPORTED;
    return launch_direct<7, 16, 2>(a, B, st);
}
