// corr_otf.hip -- on-the-fly correlation lookup ("alt": PytorchAlternateCorrBlock1D, core/corr.py:64-107).
//
// No correlation volume exists; every call recomputes, per pixel and level, the 2r+1 correlations between the
// left feature vector and the (bilinearly sampled) pooled right feature map, / sqrt(C).
//
// A block owns a 32-pixel segment of one image row at one level.  Disparity is piecewise smooth, so the sampling
// windows of the segment's pixels cover a short run of right-map columns: that run -- the right-feature
// disparity window, all C channels -- is staged in LDS ONCE (coalesced 256-byte row reads) and serves all
// 2r+1 taps of all 32 pixels.  Lanes run along pixels: the left features are read coalesced and every lane
// accumulates the 2r+2 integer-column dot products of ITS window, D[j] = sum_c f1[c] * f2[c][i0 + j], from the
// staged window; the taps are blends of neighbouring D's (the sampler is linear, so blending the dot products
// equals dotting the blended features up to fp32 round-off).  The channel range is split eight ways -- two
// half-waves x four waves -- and reduced with one lane shuffle (xor 32) plus one pass through LDS.
// Per pixel and level that is (2r+2) LDS reads + (2r+2) FMAs per channel instead of the 4 scattered global
// loads + 6 flops per (channel, tap) of a gather.
//
// grid_sample is bilinear in y too, and the reference's coordinate round trip 2y/(H-1) -> (g+1)(H-1)/2 does not
// always return the integer row exactly (it may come back as row-1 with weight 0.99999994): the window is
// therefore staged for row y0 and, when any pixel gives it weight, for row y0+1, and the four-tap weights of
// the reference (nw, ne, sw, se) are applied to the two rows' dot products.
//
// Blocks whose coordinates defeat the scheme -- pixels of the segment sampling different rows (general optical
// flow), windows spread over more than OTF_WIN columns, non-finite coordinates -- take the general path: the
// reference's four-tap grid_sample arithmetic gathered from global memory, same reduction.
#include "dkt_common.h"

#define OTF_PX 32          // pixels per block
#define OTF_WIN 64         // staged right-map columns per channel
#define OTF_UB 16          // loads in flight per lane in the staging loop
#define OTF_CG 32          // channels per staged group and wave

struct OtfArgs {
    const float *f1;
    DktPtrs f2;
    const float *coords;
    float *out;
    int C, H, W1, W2, L, nseg;
    float sqrtC;
};

__device__ __forceinline__ int otf_clamp_idx(float fl, int W) {
    return (int)fminf(fmaxf(fl, -2.0f), (float)W + 1.0f);
}

template <int R>
__global__ __launch_bounds__(256) void corr1d_otf_kernel(OtfArgs a) {
    constexpr int K = 2 * R + 1, NW = K + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [4 waves][OTF_CG][OTF_WIN] windows, then the partials
    __shared__ int s_lo, s_hi, s_slow, s_ylo, s_yhi, s_need;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int lv = blockIdx.y, b = blockIdx.z;
    const long hrow = blockIdx.x / a.nseg;
    const int w1 = (int)(blockIdx.x - hrow * a.nseg) * OTF_PX + li;
    const bool live = w1 < a.W1;
    const int w1c = live ? w1 : a.W1 - 1;
    const long HW = (long)a.H * a.W1;
    const long p = hrow * a.W1 + w1c;
    const int wi = a.W2 >> lv;
    const size_t cs2 = (size_t)a.H * wi;                             // channel stride of the pooled right map
    const float *img = a.f2.p[lv] + (size_t)b * a.C * cs2;
    const float *pf1 = a.f1 + (size_t)b * a.C * HW + p;
    const float cx = a.coords[(size_t)b * 2 * HW + p];
    const float cy = a.coords[(size_t)b * 2 * HW + HW + p];
    const float wm1 = (float)(wi - 1), hm1 = (float)(a.H - 1);
    const float xc = __fdiv_rn(cx, (float)(1 << lv));
    DktTap taps[K];
#pragma unroll
    for (int k = 0; k < K; ++k) taps[k] = dkt_tap(__fadd_rn(xc, (float)(k - R)), wm1, __fdiv_rn(wm1, 2.0f));
    const DktTap ty = dkt_tap(cy, hm1, __fdiv_rn(hm1, 2.0f));
    // window start: the first tap's column, clamped only where every tap of the window is outside the row on
    // either side anyway (so that the int conversion is defined and the window stays K+1 consecutive columns)
    const float fl0 = taps[0].fl;
    const bool far = fl0 < -(float)(K + 2) || fl0 > (float)(wi + 1);
    const int i0 = (int)fminf(fmaxf(fl0, -(float)(K + 2)), (float)(wi + 1));
    const int y0 = otf_clamp_idx(ty.fl, a.H);
    // this lane fits the windowed scheme: taps at consecutive columns (always, unless x is so large that
    // x + k is not exact, or not finite); the row must be block-uniform (below)
    bool regular = fl0 == fl0;
#pragma unroll
    for (int k = 0; k < K; ++k) regular &= far || taps[k].fl == __fadd_rn(fl0, (float)k);
    // columns of the row this lane really reads (empty when the whole window lies outside the row)
    const int c_lo = i0 < 0 ? 0 : i0, c_hi = i0 + K >= wi ? wi - 1 : i0 + K;
    const bool touches = c_lo <= c_hi;
    if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; s_slow = 0; s_ylo = 0x7fffffff; s_yhi = -0x7fffffff; s_need = 0; }
    __syncthreads();
    if (wave == 0 && half == 0) {
        if (!regular) atomicOr(&s_slow, 1);
        else if (touches) { atomicMin(&s_lo, c_lo); atomicMax(&s_hi, c_hi); }
        atomicMin(&s_ylo, y0);
        atomicMax(&s_yhi, y0);
        // which of the two rows carry weight for some pixel (bit 0: row y0, bit 1: row y0 + 1)
        atomicOr(&s_need, (ty.e != 0.0f ? 1 : 0) | (ty.w != 0.0f ? 2 : 0));
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    const bool slow = s_slow != 0 || s_ylo != s_yhi || (hi >= lo && hi - lo + 1 > OTF_WIN);
    // channel split: wave -> quarter, half-wave -> eighth
    const int cq = (a.C + 3) / 4, c_w0 = wave * cq, c_w1 = min(a.C, c_w0 + cq);
    const int ch = (c_w1 - c_w0 + 1) / 2;
    const int c0 = c_w0 + half * ch, c1 = min(c_w1, c0 + ch);
    float out_k[K];

    if (!slow) {
        const int ncol = hi >= lo ? hi - lo + 1 : 0;
        bool in[NW];
        int off[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int c = i0 + j;
            in[j] = c >= 0 && c < wi;
            off[j] = in[j] ? c - lo : 0;
        }
        float D[2][NW];                                              // dot products against rows y0 and y0 + 1
#pragma unroll
        for (int j = 0; j < NW; ++j) D[0][j] = D[1][j] = 0.0f;
        const int need = s_need;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = y0 + pass;                               // block-uniform
            if (!(need & (1 << pass)) || row < 0 || row >= a.H) continue;
            // ---- wave w walks its channel quarter in groups of OTF_CG channels: stage the group's window
            // (lane = column; 16 row reads in flight), then each half-wave dots half of the group.  The
            // group buffer is wave-private (program order suffices: no barrier) and small (8 KB per wave),
            // so that 5 blocks = 20 waves per CU hide the latency of these two short load batches.
            float *wbuf = lds + (size_t)wave * OTF_CG * OTF_WIN;
            const float *rowp = img + (size_t)row * wi + lo + (lane < ncol ? lane : 0);
            for (int cg0 = c_w0; cg0 < c_w1; cg0 += OTF_CG) {
                const int cg1 = min(c_w1, cg0 + OTF_CG);
#pragma unroll
                for (int ub = 0; ub < OTF_CG; ub += OTF_UB) {
                    float t[OTF_UB];
#pragma unroll
                    for (int u = 0; u < OTF_UB; ++u) t[u] = rowp[(size_t)min(cg0 + ub + u, cg1 - 1) * cs2];
#pragma unroll
                    for (int u = 0; u < OTF_UB; ++u)
                        if (lane < ncol) wbuf[(ub + u) * OTF_WIN + lane] = t[u];
                }
                // half-wave `half` takes channels [cg0 + half*CG/2, +CG/2) of the group
                const int h0 = cg0 + half * (OTF_CG / 2);
                float f[OTF_CG / 2];
#pragma unroll
                for (int u = 0; u < OTF_CG / 2; ++u) f[u] = (h0 + u < cg1) ? pf1[(size_t)min(h0 + u, cg1 - 1) * HW] : 0.0f;
#pragma unroll
                for (int u = 0; u < OTF_CG / 2; ++u) {
                    const float *wrow = wbuf + (half * (OTF_CG / 2) + u) * OTF_WIN;
#pragma unroll
                    for (int j = 0; j < NW; ++j) D[pass][j] = __fmaf_rn(f[u], wrow[off[j]], D[pass][j]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                D[r][j] = in[j] ? D[r][j] : 0.0f;
                D[r][j] = __fadd_rn(D[r][j], __shfl_xor(D[r][j], 32));   // the two channel halves of the wave
            }
        __syncthreads();                                             // every wave is done with its window
        float *part = lds;                                           // [4 waves][2][NW][32]
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < NW; ++j) part[((wave * 2 + r) * NW + j) * 32 + li] = D[r][j];
        }
        __syncthreads();
        if (wave != 0 || half != 0) return;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < NW; ++j)
                D[r][j] = __fadd_rn(__fadd_rn(part[((0 * 2 + r) * NW + j) * 32 + li], part[((1 * 2 + r) * NW + j) * 32 + li]),
                                    __fadd_rn(part[((2 * 2 + r) * NW + j) * 32 + li], part[((3 * 2 + r) * NW + j) * 32 + li]));
        // the reference's four-tap weights (ATen grid_sample: nw, ne, sw, se) on the dot products
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const DktTap tx = taps[k];
            const float nw = __fmul_rn(ty.e, tx.e), ne = __fmul_rn(ty.e, tx.w);
            const float sw = __fmul_rn(ty.w, tx.e), se = __fmul_rn(ty.w, tx.w);
            out_k[k] = __fmaf_rn(D[1][k + 1], se, __fmaf_rn(D[1][k], sw, __fmaf_rn(D[0][k + 1], ne, __fmul_rn(D[0][k], nw))));
        }
    } else {
        // ---- general path: the reference's four-tap sample per (channel, tap), gathered from global memory
        const bool y0ok = y0 >= 0 && y0 < a.H, y1ok = y0 + 1 >= 0 && y0 + 1 < a.H;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const DktTap tx = taps[k];
            const int x0 = otf_clamp_idx(tx.fl, wi);
            const bool x0ok = x0 >= 0 && x0 < wi, x1ok = x0 + 1 >= 0 && x0 + 1 < wi;
            const float nw = __fmul_rn(ty.e, tx.e), ne = __fmul_rn(ty.e, tx.w);
            const float sw = __fmul_rn(ty.w, tx.e), se = __fmul_rn(ty.w, tx.w);
            const long onw = (long)y0 * wi + x0;
            float acc = 0.0f;
            for (int c = c0; c < c1; ++c) {
                const float *im = img + (size_t)c * cs2;
                const float vnw = (x0ok && y0ok) ? im[onw] : 0.0f;
                const float vne = (x1ok && y0ok) ? im[onw + 1] : 0.0f;
                const float vsw = (x0ok && y1ok) ? im[onw + wi] : 0.0f;
                const float vse = (x1ok && y1ok) ? im[onw + wi + 1] : 0.0f;
                const float s = __fmaf_rn(vse, se, __fmaf_rn(vsw, sw, __fmaf_rn(vne, ne, __fmul_rn(vnw, nw))));
                acc = __fmaf_rn(s, pf1[(size_t)c * HW], acc);
            }
            out_k[k] = __fadd_rn(acc, __shfl_xor(acc, 32));
        }
        __syncthreads();
        float *part = lds;                                           // [4 waves][K][32]
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) part[(wave * K + k) * 32 + li] = out_k[k];
        }
        __syncthreads();
        if (wave != 0 || half != 0) return;
#pragma unroll
        for (int k = 0; k < K; ++k)
            out_k[k] = __fadd_rn(__fadd_rn(part[(0 * K + k) * 32 + li], part[(1 * K + k) * 32 + li]),
                                 __fadd_rn(part[(2 * K + k) * 32 + li], part[(3 * K + k) * 32 + li]));
    }
    if (!live) return;
    float *o = a.out + ((size_t)b * a.L * K + (size_t)lv * K) * HW + p;
#pragma unroll
    for (int k = 0; k < K; ++k) o[(size_t)k * HW] = __fdiv_rn(out_k[k], a.sqrtC);
}

template <int R>
static int otf_launch(const OtfArgs &a, int B, hipStream_t st) {
    const size_t need = (size_t)4 * OTF_CG * OTF_WIN * sizeof(float);
    const size_t part = (size_t)4 * 2 * (2 * R + 2) * 32 * sizeof(float);
    const size_t lds = need > part ? need : part;
    if (lds > 160 * 1024) return DKT_E_UNSUPPORTED;
    auto kern = corr1d_otf_kernel<R>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)((long)a.H * a.nseg), (unsigned)a.L, (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return dkt_launch_status();
}

extern "C" int dkt_corr1d_lookup_otf(const float *f1, const float *const *f2pyr, const float *coords,
                                     float *out, int B, int C, int H, int W1, int W2, int L, int r,
                                     int device, void *stream) {
    if (!f1 || !f2pyr || !coords || !out) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    OtfArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.f2.p[i] = i < L ? f2pyr[i] : nullptr;
        if (i < L && !f2pyr[i]) return DKT_E_NULL;
    }
    if ((long)H * ((W1 + OTF_PX - 1) / OTF_PX) > 0x7fffffffL) return DKT_E_SHAPE;
    DKT_ENTER(device);
    a.f1 = f1; a.coords = coords; a.out = out;
    a.C = C; a.H = H; a.W1 = W1; a.W2 = W2; a.L = L;
    a.nseg = (W1 + OTF_PX - 1) / OTF_PX;
    a.sqrtC = sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    switch (r) {
        case 0: return otf_launch<0>(a, B, st);
        case 1: return otf_launch<1>(a, B, st);
        case 2: return otf_launch<2>(a, B, st);
        case 3: return otf_launch<3>(a, B, st);
        case 4: return otf_launch<4>(a, B, st);
        case 5: return otf_launch<5>(a, B, st);
        case 6: return otf_launch<6>(a, B, st);
        case 7: return otf_launch<7>(a, B, st);
        default: return otf_launch<8>(a, B, st);
    }
}
