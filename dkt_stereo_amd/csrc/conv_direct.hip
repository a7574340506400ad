// conv_direct.hip -- direct fp32 convolution for the two extreme layer shapes of the update
// block that a 64-channel-wide MFMA tile wastes:
//   * very few OUTPUT channels: flow_head.conv2 256->2 / disp_head.conv2 256->1, 3x3
//     (core/update.py:10, igev_stereo/update.py:20) -- the MFMA kernel pads 2 -> 64 channels;
//   * very few INPUT channels: convf1 2->64 / convd1 1->64, 7x7 (core/update.py:75,
//     igev_stereo/update.py:81) -- K = 2*49 would be padded to 49 chunks of 32.
// Exact fp32 FMAs on the VALU (no operand splitting), accumulated in (channel, row, column) tap
// order; the input patch of a channel chunk and its weights are staged in LDS.  Optional fused ReLU.
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

// Thread = 4 horizontally adjacent pixels x TO output channels (register blocking: the 4 + 2*HALO
// activations of a patch row are read from LDS once -- one ds_read_b128 + remainder -- and feed
// 4*KS*TO FMAs; a weight, read as an LDS broadcast, feeds 4).  Block = 16 x 16 threads = a
// 16-row x 64-column output tile; CC input channels are staged per round.
template <int KS, int TO, int CC>
__global__ __launch_bounds__(256) void conv2d_direct_kernel(DirectArgs a) {
    constexpr int HALO = KS / 2, PX = 4;
    constexpr int TR = 16, TC = 64;
    constexpr int PR = TR + 2 * HALO, PC = TC + 2 * HALO;
    constexpr int PCP = (PC + 3) & ~3;                   // row pitch: 16-byte aligned rows
    constexpr int TAPS = KS * KS;
    constexpr int NA = PX + 2 * HALO;                    // activations per thread and patch row
    constexpr int NE = (PR * PC + 255) / 256;            // patch elements per thread and channel
    __shared__ __attribute__((aligned(16))) float patch[CC][PR][PCP];
    __shared__ __attribute__((aligned(16))) float wl[CC][TAPS][TO];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int w0 = (blockIdx.x % a.tiles_w) * TC, h0 = (blockIdx.x / a.tiles_w) * TR;
    const int co0 = blockIdx.y * TO;
    const int b = blockIdx.z;
    const long HW = (long)a.H * a.W;
    const float *xb = a.x + (long)b * a.x_bs;
    // this thread's patch elements: the same (row, column) positions in every channel
    int goff[NE], loff[NE];
    bool gok[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int i = tid + 256 * k;
        const int r = i / PC, q = i - r * PC;
        const int ih = h0 - HALO + r, iw = w0 - HALO + q;
        gok[k] = i < PR * PC && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        goff[k] = gok[k] ? ih * a.W + iw : 0;
        loff[k] = i < PR * PC ? r * PCP + q : -1;
    }
    float acc[PX][TO];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int o = 0; o < TO; ++o) acc[p][o] = 0.0f;

    for (int c0 = 0; c0 < a.Cin; c0 += CC) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const bool cok = c0 + c < a.Cin;                            // block-uniform
            const float *xc = xb + (long)(cok ? c0 + c : 0) * HW;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const float v = xc[goff[k]];
                if (loff[k] >= 0) (&patch[c][0][0])[loff[k]] = (cok && gok[k]) ? v : 0.0f;
            }
        }
        for (int i = tid; i < CC * TAPS * TO; i += 256) {
            const int o = i % TO, t = (i / TO) % TAPS, c = i / (TO * TAPS);
            float v = 0.0f;
            if (c0 + c < a.Cin && co0 + o < a.Cout) v = a.w[((long)(co0 + o) * a.Cin + (c0 + c)) * TAPS + t];
            wl[c][t][o] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < CC; ++c) {
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
                float v[NA];
                const float *pr = &patch[c][ty + dy][PX * tx];
#pragma unroll
                for (int j = 0; j + 3 < NA; j += 4) *(float4 *)&v[j] = *(const float4 *)(pr + j);
#pragma unroll
                for (int j = NA & ~3; j < NA; ++j) v[j] = pr[j];
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    float wv[TO];
#pragma unroll
                    for (int o = 0; o < TO; ++o) wv[o] = wl[c][dy * KS + dx][o];
#pragma unroll
                    for (int p = 0; p < PX; ++p)
#pragma unroll
                        for (int o = 0; o < TO; ++o) acc[p][o] = __fmaf_rn(v[p + dx], wv[o], acc[p][o]);
                }
            }
        }
    }
    const int oh = h0 + ty;
    if (oh >= a.H) return;
#pragma unroll
    for (int o = 0; o < TO; ++o) {
        if (co0 + o >= a.Cout) continue;
        const float bias = a.bias ? a.bias[co0 + o] : 0.0f;
        float *yr = a.y + (long)b * a.y_bs + (long)(co0 + o) * HW + (long)oh * a.W;
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int ow = w0 + PX * tx + p;
            if (ow >= a.W) continue;
            float v = acc[p][o] + bias;
            if (a.relu) v = dkt_relu(v);
            yr[ow] = v;
        }
    }
}

template <int KS, int TO, int CC>
static int launch_direct(const DirectArgs &a, int B, hipStream_t st) {
    const int tiles_h = (a.H + 15) / 16;
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
    a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.tiles_w = (W + 63) / 64;
    a.relu = relu ? 1 : 0;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    if (KH == 3) {
        if (Cout > 4) return DKT_E_UNSUPPORTED;       // wide layers belong to dkt_conv2d_f16s
        if (Cout <= 2) return launch_direct<3, 2, 8>(a, B, st);
        return launch_direct<3, 4, 8>(a, B, st);
    }
    if (Cin > 4) return DKT_E_UNSUPPORTED;
    return launch_direct<7, 16, 2>(a, B, st);
}
