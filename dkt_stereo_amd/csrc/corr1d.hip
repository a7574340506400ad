// corr1d.hip -- RAFT-Stereo 1-D correlation for gfx950 (MI355X):
//   dkt_corr1d_build      all-pairs correlation (exact-fp32 MFMA) fused with
//                         scaling and the avg-pool pyramid epilogue
//   dkt_corr1d_lookup     per-iteration 9-tap bilinear pyramid lookup
//   dkt_corr1d_lookup_otf volume-free lookup ("alt" implementation)
//   dkt_pool_w, dkt_l2norm_channels helpers
// Reference behaviour: core/corr.py:64-156, core/utils/utils.py:59-74.
#include "dkt_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Bijective XCD-aware block remap: hardware places block b on XCD b%8; give
// every XCD a contiguous range of virtual ids so that blocks sharing one
// (b,h) feature row also share one L2.
__device__ __forceinline__ unsigned dkt_xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned nx = 8;
    unsigned q = nblk / nx, r = nblk % nx;
    unsigned xcd = bid % nx, idx = bid / nx;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------------------------------------------------------------------
// Build: one wave owns a 64(w1) x 64(w2) output tile of one (b,h) row and walks
// the channel (K) dimension two at a time with v_mfma_f32_32x32x2_f32.  Both
// operands are channel-major in memory (NCHW), which is exactly the MFMA A/B
// fragment order (lane l: A[m = l&31][k = l>>5], B[k = l>>5][n = l&31]): every
// fragment load is two coalesced 128-byte rows, no LDS transpose is needed.
// The accumulator (C/D map: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5))
// keeps w2 along lanes, so level-0 stores are 128-byte rows and the pyramid is
// pooled with cross-lane shuffles straight from registers.
// ---------------------------------------------------------------------------
struct CorrBuildArgs {
    const float *f1;
    const float *f2;
    DktMutPtrs pyr;
    int C, H, W1, W2, L;
    int tiles_m, tiles_n;
    unsigned total_tiles;
    float divisor;
};

__global__ __launch_bounds__(256) void corr1d_build_kernel(CorrBuildArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned nblk = gridDim.x;
    const unsigned vb = dkt_xcd_remap(blockIdx.x, nblk);
    const unsigned tile = vb * 4 + wave;
    if (tile >= a.total_tiles) return;
    const unsigned per_row = (unsigned)(a.tiles_m * a.tiles_n);
    const unsigned bh = tile / per_row;
    const unsigned t = tile % per_row;
    const int m0 = (int)(t / a.tiles_n) * 64;
    const int n0 = (int)(t % a.tiles_n) * 64;
    const int b = (int)(bh / a.H), h = (int)(bh % a.H);

    const int kk = lane >> 5;   // which of the two channels this half-wave feeds
    const int li = lane & 31;
    const size_t cs1 = (size_t)a.H * a.W1, cs2 = (size_t)a.H * a.W2;
    const float *pa = a.f1 + ((size_t)b * a.C + kk) * cs1 + (size_t)h * a.W1;
    const float *pb = a.f2 + ((size_t)b * a.C + kk) * cs2 + (size_t)h * a.W2;
    const int ma0 = m0 + li, ma1 = m0 + 32 + li;
    const int nb0 = n0 + li, nb1 = n0 + 32 + li;
    const bool va0 = ma0 < a.W1, va1 = ma1 < a.W1, vb0 = nb0 < a.W2, vb1 = nb1 < a.W2;
    // branch-free edge handling: out-of-range lanes read a clamped (valid) column
    // and their operand is zeroed afterwards, so every load is unconditional and
    // the loads of group g+1 can be in flight under the MFMAs of group g.
    const int ca0 = va0 ? ma0 : a.W1 - 1, ca1 = va1 ? ma1 : a.W1 - 1;
    const int cb0 = vb0 ? nb0 : a.W2 - 1, cb1 = vb1 ? nb1 : a.W2 - 1;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int C2 = a.C & ~1;
    constexpr int P = 8;                         // channel pairs per software-pipeline stage
    const int ngroups = (C2 / 2) / P;
    float xa0[P], xa1[P], xb0[P], xb1[P];
    if (ngroups > 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            xa0[j] = pa[(size_t)j * 2 * cs1 + ca0];
            xa1[j] = pa[(size_t)j * 2 * cs1 + ca1];
            xb0[j] = pb[(size_t)j * 2 * cs2 + cb0];
            xb1[j] = pb[(size_t)j * 2 * cs2 + cb1];
        }
    }
    for (int g = 0; g < ngroups; ++g) {
        pa += (size_t)P * 2 * cs1;
        pb += (size_t)P * 2 * cs2;
        float ya0[P], ya1[P], yb0[P], yb1[P];
        const bool more = g + 1 < ngroups;       // wave-uniform
        if (more) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                ya0[j] = pa[(size_t)j * 2 * cs1 + ca0];
                ya1[j] = pa[(size_t)j * 2 * cs1 + ca1];
                yb0[j] = pb[(size_t)j * 2 * cs2 + cb0];
                yb1[j] = pb[(size_t)j * 2 * cs2 + cb1];
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float a0 = va0 ? xa0[j] : 0.0f, a1 = va1 ? xa1[j] : 0.0f;
            const float b0 = vb0 ? xb0[j] : 0.0f, b1 = vb1 ? xb1[j] : 0.0f;
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                xa0[j] = ya0[j]; xa1[j] = ya1[j]; xb0[j] = yb0[j]; xb1[j] = yb1[j];
            }
        }
    }
    for (int c = ngroups * P * 2; c < C2; c += 2) {   // remaining pairs (C/2 not a multiple of P)
        const float l0 = pa[ca0], l1 = pa[ca1], l2 = pb[cb0], l3 = pb[cb1];
        const float a0 = va0 ? l0 : 0.0f, a1 = va1 ? l1 : 0.0f, b0 = vb0 ? l2 : 0.0f, b1 = vb1 ? l3 : 0.0f;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        pa += 2 * cs1;
        pb += 2 * cs2;
    }
    if (a.C & 1) {  // odd channel count: upper half-wave feeds zeros
        float a0 = (va0 && kk == 0) ? pa[ma0] : 0.0f;
        float a1 = (va1 && kk == 0) ? pa[ma1] : 0.0f;
        float b0 = (vb0 && kk == 0) ? pb[nb0] : 0.0f;
        float b1 = (vb1 && kk == 0) ? pb[nb1] : 0.0f;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }

    // epilogue: scale, store level 0, pool in registers, store levels 1..L-1
    const size_t nrow0 = (size_t)bh * a.W1;
    const int half = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const f32x16 &acc = mi == 0 ? (ni == 0 ? acc00 : acc01) : (ni == 0 ? acc10 : acc11);
            const int col = n0 + ni * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool mok = m < a.W1;
                float v = __fdiv_rn(acc[r], a.divisor);
                if (mok && col < a.W2) a.pyr.p[0][(nrow0 + m) * (size_t)a.W2 + col] = v;
                int wi = a.W2;
#pragma unroll
                for (int lv = 1; lv < 6; ++lv) {
                    if (lv >= a.L) break;
                    wi >>= 1;
                    float o = __shfl_xor(v, 1 << (lv - 1));
                    v = __fmul_rn(__fadd_rn(v, o), 0.5f);
                    const int k = col >> lv;
                    if (mok && (li & ((1 << lv) - 1)) == 0 && k < wi)
                        a.pyr.p[lv][(nrow0 + m) * (size_t)wi + k] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void pool_w_kernel(const float *__restrict__ src,
                                                     float *__restrict__ dst, long rows, int W) {
    const int wo = W >> 1;
    const long total = rows * (long)wo;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long n = i / wo;
        int k = (int)(i - n * wo);
        const float *s = src + n * (long)W + 2 * k;
        dst[i] = __fmul_rn(__fadd_rn(s[0], s[1]), 0.5f);
    }
}

static int launch_pool_w(const float *src, float *dst, long rows, int W, hipStream_t st) {
    long total = rows * (long)(W >> 1);
    if (total <= 0) return DKT_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipLaunchKernelGGL(pool_w_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, rows, W);
    return dkt_launch_status();
}

extern "C" int dkt_pool_w(const float *src, float *dst, long rows, int W, int device, void *stream) {
    if (!src || !dst) return DKT_E_NULL;
    if (rows <= 0 || W < 2) return DKT_E_SHAPE;
    DKT_ENTER(device);
    return launch_pool_w(src, dst, rows, W, (hipStream_t)stream);
}

extern "C" int dkt_corr1d_build(const float *f1, const float *f2, float *const *pyr,
                                int B, int C, int H, int W1, int W2, int L, float divisor,
                                int device, void *stream) {
    if (!f1 || !f2 || !pyr) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || !(divisor != 0.0f)) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    for (int i = 0; i < L; ++i)
        if (!pyr[i]) return DKT_E_NULL;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    CorrBuildArgs a;
    a.f1 = f1;
    a.f2 = f2;
    const int Lf = L < 6 ? L : 6;  // levels pooled inside one 32-wide MFMA tile
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) a.pyr.p[i] = i < L ? pyr[i] : nullptr;
    a.C = C; a.H = H; a.W1 = W1; a.W2 = W2; a.L = Lf;
    a.tiles_m = (W1 + 63) / 64;
    a.tiles_n = (W2 + 63) / 64;
    unsigned long long total = (unsigned long long)B * H * a.tiles_m * a.tiles_n;
    if (total > 0xFFFFFFF0ull) return DKT_E_SHAPE;
    a.total_tiles = (unsigned)total;
    a.divisor = divisor;
    unsigned blocks = (a.total_tiles + 3) / 4;
    hipLaunchKernelGGL(corr1d_build_kernel, dim3(blocks), dim3(256), 0, st, a);
    int rc = dkt_launch_status();
    if (rc) return rc;
    for (int i = Lf; i < L; ++i) {
        rc = launch_pool_w(pyr[i - 1], pyr[i], (long)B * H * W1, W2 >> (i - 1), st);
        if (rc) return rc;
    }
    return DKT_OK;
}

// ---------------------------------------------------------------------------
// Lookup: one thread per (pixel, level).  Lanes run along w1, so every one of
// the 2r+1 output stores is a contiguous 256-byte wave store into its own
// channel plane (the reference's permute(0,3,1,2).contiguous() for free).
// Each thread reads one contiguous (2r+2)-float window of its own pyramid row;
// tap k normally lands on window slots k,k+1, and the rare tap whose
// coordinate round trip floors differently is re-fetched exactly.
// ---------------------------------------------------------------------------
struct LookupArgs {
    DktPtrs pyr;
    const float *coords_x;
    long coords_bstride;
    float *out;
    long HW;
    int W2, L;
};

__device__ __forceinline__ float dkt_row_at(const float *row, int idx, int W) {
    return (idx >= 0 && idx < W) ? row[idx] : 0.0f;
}

__device__ __forceinline__ int dkt_clamp_idx(float fl, int W) {
    // both taps are outside the row beyond these bounds; keeps (int) defined
    return (int)fminf(fmaxf(fl, -2.0f), (float)W + 1.0f);
}

template <int R>
__global__ __launch_bounds__(256) void corr1d_lookup_kernel(LookupArgs a) {
    constexpr int K = 2 * R + 1;
    const long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= a.HW) return;
    const int lv = blockIdx.y;
    const int b = blockIdx.z;
    const int wi = a.W2 >> lv;
    const float *row = a.pyr.p[lv] + ((size_t)b * a.HW + p) * (size_t)wi;
    const float cx = a.coords_x[(size_t)b * a.coords_bstride + p];
    const float xc = __fdiv_rn(cx, (float)(1 << lv));
    const float wm1 = (float)(wi - 1);
    const float hwm1 = __fdiv_rn(wm1, 2.0f);

    DktTap taps[K];
#pragma unroll
    for (int k = 0; k < K; ++k) taps[k] = dkt_tap(__fadd_rn((float)(k - R), xc), wm1, hwm1);
    const int i0 = dkt_clamp_idx(taps[0].fl, wi);
    float win[K + 1];
#pragma unroll
    for (int j = 0; j <= K; ++j) win[j] = dkt_row_at(row, i0 + j, wi);

    float *o = a.out + ((size_t)b * a.L * K + (size_t)lv * K) * a.HW + p;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int ik = dkt_clamp_idx(taps[k].fl, wi);
        float v0 = win[k], v1 = win[k + 1];
        if (ik != i0 + k) {
            v0 = dkt_row_at(row, ik, wi);
            v1 = dkt_row_at(row, ik + 1, wi);
        }
        o[(size_t)k * a.HW] = dkt_blend(v0, v1, taps[k]);
    }
}

// ---------------------------------------------------------------------------
// Lookup, cooperative form: FOUR lanes per (pixel, level).  The (2r+2)-float
// window of a pixel sits inside 16 consecutive floats starting at the 16-byte
// boundary below it; each of the four lanes fetches one aligned float4 of that
// span (one 16-byte request instead of 2r+2 scalar requests that all hit the
// same one or two cache lines), the span is parked in LDS (pitch 20 floats:
// conflict-free for both the b128 writes and the scalar reads), and lane q then
// evaluates taps q, q+4, q+8 with exactly the scalar kernel's arithmetic.
// Row ends are masked by column index (zero padding), not by what was fetched.
// Needs 2r+2 <= 13, i.e. r <= 5; other radii use the scalar kernel.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void corr1d_lookup4_kernel(LookupArgs a) {
    constexpr int K = 2 * R + 1;
    constexpr int PITCH = 20;
    __shared__ __attribute__((aligned(16))) float span[64 * PITCH];
    const int q = threadIdx.x & 3;
    const int g = threadIdx.x >> 2;                 // pixel slot in this block
    const long p = blockIdx.x * 64L + g;
    const int lv = blockIdx.y;
    const int b = blockIdx.z;
    const int wi = a.W2 >> lv;
    const bool live = p < a.HW;
    const long pc = live ? p : a.HW - 1;
    const size_t n = (size_t)b * a.HW + pc;
    const float *lvl = a.pyr.p[lv];
    const float cx = a.coords_x[(size_t)b * a.coords_bstride + pc];
    const float xc = __fdiv_rn(cx, (float)(1 << lv));
    const float wm1 = (float)(wi - 1);
    const float hwm1 = __fdiv_rn(wm1, 2.0f);
    const DktTap t0 = dkt_tap(__fadd_rn((float)(-R), xc), wm1, hwm1);
    const int i0 = dkt_clamp_idx(t0.fl, wi);
    // element offsets inside this level's buffer
    const long row0 = (long)n * wi;
    const long e0 = row0 + i0;
    const long eal = e0 & ~3L;                      // may be negative by up to 3 (+2 from i0>=-2)
    const long nelem = ((long)gridDim.z * a.HW) * wi;
    {
        const long e = eal + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e >= 0 && e + 3 < nelem) {
            v = *(const float4 *)(lvl + e);
        } else {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (e + j >= 0 && e + j < nelem) ? lvl[e + j] : 0.0f;
            v = make_float4(t[0], t[1], t[2], t[3]);
        }
        // zero everything that is not a column of THIS row
        const long c0 = e - row0;
        if (c0 < 0 || c0 >= wi) v.x = 0.0f;
        if (c0 + 1 < 0 || c0 + 1 >= wi) v.y = 0.0f;
        if (c0 + 2 < 0 || c0 + 2 >= wi) v.z = 0.0f;
        if (c0 + 3 < 0 || c0 + 3 >= wi) v.w = 0.0f;
        *(float4 *)(span + g * PITCH + 4 * q) = v;
    }
    __syncthreads();
    if (!live) return;
    const int m = (int)(e0 - eal);                   // 0..3: where the window starts in the span
    const float *row = lvl + row0;
    float *o = a.out + ((size_t)b * a.L * K + (size_t)lv * K) * a.HW + p;
#pragma unroll
    for (int j = 0; j < (K + 3) / 4; ++j) {
        const int k = q + 4 * j;
        if (k >= K) break;
        const DktTap t = dkt_tap(__fadd_rn((float)(k - R), xc), wm1, hwm1);
        const int ik = dkt_clamp_idx(t.fl, wi);
        float v0, v1;
        if (ik == i0 + k) {
            v0 = span[g * PITCH + m + k];
            v1 = span[g * PITCH + m + k + 1];
        } else {
            v0 = dkt_row_at(row, ik, wi);
            v1 = dkt_row_at(row, ik + 1, wi);
        }
        o[(size_t)k * a.HW] = dkt_blend(v0, v1, t);
    }
}


template <int R>
static void launch_lookup(const LookupArgs &a, int B, hipStream_t st) {
    if constexpr (2 * R + 2 <= 13) {
        {
            dim3 grid((unsigned)((a.HW + 63) / 64), (unsigned)a.L, (unsigned)B);
            hipLaunchKernelGGL(corr1d_lookup4_kernel<R>, grid, dim3(256), 0, st, a);
            return;
        }
    }
    dim3 grid((unsigned)((a.HW + 255) / 256), (unsigned)a.L, (unsigned)B);
    hipLaunchKernelGGL(corr1d_lookup_kernel<R>, grid, dim3(256), 0, st, a);
}

extern "C" int dkt_corr1d_lookup(const float *const *pyr, const float *coords_x, long coords_bstride,
                                 float *out, int B, int H, int W1, int W2, int L, int r,
                                 int device, void *stream) {
    if (!pyr || !coords_x || !out) return DKT_E_NULL;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    LookupArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.pyr.p[i] = i < L ? pyr[i] : nullptr;
        if (i < L && !pyr[i]) return DKT_E_NULL;
    }
    DKT_ENTER(device);
    a.coords_x = coords_x;
    a.coords_bstride = coords_bstride;
    a.out = out;
    a.HW = (long)H * W1;
    a.W2 = W2;
    a.L = L;
    hipStream_t st = (hipStream_t)stream;
    switch (r) {
        case 0: launch_lookup<0>(a, B, st); break;
        case 1: launch_lookup<1>(a, B, st); break;
        case 2: launch_lookup<2>(a, B, st); break;
        case 3: launch_lookup<3>(a, B, st); break;
        case 4: launch_lookup<4>(a, B, st); break;
        case 5: launch_lookup<5>(a, B, st); break;
        case 6: launch_lookup<6>(a, B, st); break;
        case 7: launch_lookup<7>(a, B, st); break;
        default: launch_lookup<8>(a, B, st); break;
    }
    return dkt_launch_status();
}

// ---------------------------------------------------------------------------
// L2 normalisation over channels (CorrBlock1D_Cosine prologue, corr.py:201-202)
// torch: x / x.norm(dim=1, keepdim=True); norm = sqrt(sum x^2)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(const float *__restrict__ src,
                                                     float *__restrict__ dst, int C, long HW) {
    const long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= HW) return;
    const size_t base = (size_t)blockIdx.y * C * HW + p;
    float ss = 0.0f;
    for (int c = 0; c < C; ++c) {
        float v = src[base + (size_t)c * HW];
        ss = __fmaf_rn(v, v, ss);
    }
    const float nrm = sqrtf(ss);
    for (int c = 0; c < C; ++c) dst[base + (size_t)c * HW] = __fdiv_rn(src[base + (size_t)c * HW], nrm);
}

extern "C" int dkt_l2norm_channels(const float *src, float *dst, int B, int C, long HW,
                                   int device, void *stream) {
    if (!src || !dst) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || HW <= 0 || B > 65535) return DKT_E_SHAPE;
    DKT_ENTER(device);
    dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(l2norm_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, HW);
    return dkt_launch_status();
}
