// corr1d_skew.hip -- diagonal-major ("skewed") correlation pyramid and its lookup.
//
// In the reference's layout (core/corr.py:117) the volume row of left pixel w1 is W2 floats
// and a lookup touches a (2r+2)-float window of it around x = w1 - disparity: 40 useful
// bytes out of one or two 128-byte lines PER PIXEL AND LEVEL, and neighbouring pixels'
// windows are a whole row (1248 B at 1/4 KITTI) apart.  Measured: the row-layout lookup
// moves ~46 MB of lines for 17.7 MB of algorithmic traffic and sits at ~4 TB/s of line
// traffic = 20 % of the HBM roofline in algorithmic terms.
//
// Disparity is piecewise smooth, so the windows of neighbouring pixels lie along the
// DIAGONAL of the (w1, w2) plane.  The skewed pyramid stores that diagonal contiguously:
//     S_i[row][s][w1] = P_i[row*W1 + w1][(s + (w1 >> i)) mod W2_i],   row = b*H + h
// (a bijection of every volume row).  The w1 axis is padded to a multiple of 32 floats
// (dkt_corr1d_skew_pitch) so that every (row, s) line starts on a 128-byte boundary and a
// wave's 64-pixel segment is two whole cache lines.  A tap at integer column c of pixel
// w1 is S_i[row][(c - (w1>>i)) mod W2_i][w1]: for a wave of 64 consecutive pixels with a
// locally constant disparity every tap is ONE contiguous 256-byte read, and the bytes
// fetched equal the algorithmic bytes.  Arbitrary (non-smooth) coordinates stay correct --
// each lane then reads its own line per tap, which is slower than the row layout; the
// Python wrapper keeps the row-layout pyramid (it is the reference-visible attribute) and
// can be switched back to it.
//
// Arithmetic is identical to corr1d_lookup_kernel (same taps, same blend): results are
// bit-identical to the row-layout lookup.
#include "dkt_common.h"

struct SkewArgs {
    DktPtrs src;       // row layout, level i: (rows*W1, W2>>i)
    DktMutPtrs dst;    // skew layout, level i: (rows, W2>>i, pitch)
    int W1, W2, L, pitch;
    long rows;
};

// 32(s) x 32(w1) tile through LDS: reads are contiguous along the column index c (lanes
// along s), writes contiguous along w1.
__global__ __launch_bounds__(256) void corr1d_skew_kernel(SkewArgs a) {
    __shared__ float tile[32][33];
    const int lv = blockIdx.z;
    const int wi = a.W2 >> lv;
    const int tiles_s = (wi + 31) / 32, tiles_w = (a.W1 + 31) / 32;
    const long per_row = (long)tiles_s * tiles_w;
    const long t = blockIdx.x;
    const long row = t / per_row;
    if (row >= a.rows) return;
    const int tt = (int)(t - row * per_row);
    const int s0 = (tt % tiles_s) * 32, w0 = (tt / tiles_s) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float *src = a.src.p[lv] + row * a.W1 * (long)wi;
    float *dst = a.dst.p[lv] + row * (long)wi * a.pitch;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int w1 = w0 + ty + 8 * j, s = s0 + tx;
        float v = 0.0f;
        if (w1 < a.W1 && s < wi) {
            int c = (s + (w1 >> lv)) % wi;
            v = src[(long)w1 * wi + c];
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = s0 + ty + 8 * j, w1 = w0 + tx;
        if (w1 < a.W1 && s < wi) dst[(long)s * a.pitch + w1] = tile[tx][ty + 8 * j];
    }
}

static inline int skw_pitch(int W1) { return (W1 + 31) & ~31; }

extern "C" int dkt_corr1d_skew_pitch(int W1) { return W1 > 0 ? skw_pitch(W1) : DKT_E_SHAPE; }

extern "C" int dkt_corr1d_skew(const float *const *pyr, float *const *skew, int B, int H, int W1, int W2, int L,
                               int device, void *stream) {
    if (!pyr || !skew) return DKT_E_NULL;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    SkewArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.src.p[i] = i < L ? pyr[i] : nullptr;
        a.dst.p[i] = i < L ? skew[i] : nullptr;
        if (i < L && (!pyr[i] || !skew[i])) return DKT_E_NULL;
    }
    a.W1 = W1; a.W2 = W2; a.L = L; a.pitch = skw_pitch(W1);
    a.rows = (long)B * H;
    DKT_ENTER(device);
    // grid.x sized for level 0 (the widest); coarser levels exit early on surplus blocks
    const long per_row0 = (long)((W2 + 31) / 32) * ((W1 + 31) / 32);
    const long blocks = a.rows * per_row0;
    if (blocks > 0x7FFFFFFFL) return DKT_E_SHAPE;
    hipLaunchKernelGGL(corr1d_skew_kernel, dim3((unsigned)blocks, 1, (unsigned)L), dim3(256), 0,
                       (hipStream_t)stream, a);
    return dkt_launch_status();
}

struct SkewLookupArgs {
    DktPtrs skew;
    const float *coords_x;
    long coords_bstride;
    float *out;
    long HW;
    int H, W1, W2, L, pitch, nseg;
    float inv_wm1[DKT_MAX_LEVELS];   // RN(1 / (W2_i - 1)), computed on the host
};

__device__ __forceinline__ int skw_clamp_idx(float fl, int W) {
    return (int)fminf(fmaxf(fl, -2.0f), (float)W + 1.0f);
}

// dkt_tap (the bilinear_sampler round trip, core/utils/utils.py:59-74) with the IEEE division
// 2x / (W-1) evaluated as two Newton corrections on the host-rounded reciprocal y = RN(1/b):
//   q0 = a*y;  q1 = q0 + (a - b*q0)*y  (faithful);  q2 = q1 + (a - b*q1)*y  (correctly rounded,
// Markstein) -- 5 instructions instead of the ~10 of v_div_scale/v_div_fmas/v_div_fixup, nine
// times per thread.  tests/test_gpu_parity.py compares it bit for bit with the __fdiv_rn form
// (the row-layout kernel) over millions of coordinates and every width in use.
__device__ __forceinline__ DktTap skw_tap(float x, float wm1, float inv, float half_wm1) {
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

// One wave = one 64-pixel segment of one image row (lanes = consecutive w1, segment start
// 256-byte aligned in the padded skew row); grid.y = level, grid.z = batch.
template <int R>
__global__ __launch_bounds__(256) void corr1d_lookup_skew_kernel(SkewLookupArgs a) {
    constexpr int K = 2 * R + 1;
    const long gw = blockIdx.x * 4L + (threadIdx.x >> 6);
    const long hrow = gw / a.nseg;
    const int w1 = (int)(gw - hrow * a.nseg) * 64 + (threadIdx.x & 63);
    if (hrow >= a.H || w1 >= a.W1) return;
    const int lv = blockIdx.y;
    const int b = blockIdx.z;
    const int wi = a.W2 >> lv;
    const long p = hrow * a.W1 + w1;
    const int qm = (w1 >> lv) % wi;
    // S[row][s][w1] with row = b*H + h  ->  base + s*pitch
    const float *base = a.skew.p[lv] + (((long)b * a.H + hrow) * wi) * (long)a.pitch + w1;
    const float cx = a.coords_x[(size_t)b * a.coords_bstride + p];
    const float xc = __fdiv_rn(cx, (float)(1 << lv));
    const float wm1 = (float)(wi - 1);
    const float hwm1 = __fdiv_rn(wm1, 2.0f);
    const float inv = a.inv_wm1[lv];
    auto at = [&](int c) -> float {          // volume entry at integer column c, zero outside the row
        if (c < 0 || c >= wi) return 0.0f;
        int s = c - qm;
        if (s < 0) s += wi;
        return base[(long)s * a.pitch];
    };
    DktTap taps[K];
    if (wi > 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) taps[k] = skw_tap(__fadd_rn((float)(k - R), xc), wm1, inv, hwm1);
    } else {                                 // W-1 = 0: the reference divides by zero; keep its exact inf/NaN
#pragma unroll
        for (int k = 0; k < K; ++k) taps[k] = dkt_tap(__fadd_rn((float)(k - R), xc), wm1, hwm1);
    }
    const int i0 = skw_clamp_idx(taps[0].fl, wi);
    float win[K + 1];
#pragma unroll
    for (int j = 0; j <= K; ++j) win[j] = at(i0 + j);
    float *o = a.out + ((size_t)b * a.L * K + (size_t)lv * K) * a.HW + p;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int ik = skw_clamp_idx(taps[k].fl, wi);
        float v0 = win[k], v1 = win[k + 1];
        if (ik != i0 + k) {
            v0 = at(ik);
            v1 = at(ik + 1);
        }
        o[(size_t)k * a.HW] = dkt_blend(v0, v1, taps[k]);
    }
}

template <int R>
static void launch_skew_lookup(const SkewLookupArgs &a, int B, hipStream_t st) {
    const long waves = (long)a.H * a.nseg;
    dim3 grid((unsigned)((waves + 3) / 4), (unsigned)a.L, (unsigned)B);
    hipLaunchKernelGGL(corr1d_lookup_skew_kernel<R>, grid, dim3(256), 0, st, a);
}

extern "C" int dkt_corr1d_lookup_skew(const float *const *skew, const float *coords_x, long coords_bstride,
                                      float *out, int B, int H, int W1, int W2, int L, int r,
                                      int device, void *stream) {
    if (!skew || !coords_x || !out) return DKT_E_NULL;
    if (B <= 0 || H <= 0 || W1 <= 0 || W2 <= 0 || B > 65535) return DKT_E_SHAPE;
    if (L < 1 || L > DKT_MAX_LEVELS || (W2 >> (L - 1)) == 0) return DKT_E_LEVELS;
    if (r < 0 || r > DKT_MAX_RADIUS) return DKT_E_RADIUS;
    SkewLookupArgs a;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        a.skew.p[i] = i < L ? skew[i] : nullptr;
        if (i < L && !skew[i]) return DKT_E_NULL;
    }
    DKT_ENTER(device);
    a.coords_x = coords_x;
    a.coords_bstride = coords_bstride;
    a.out = out;
    a.HW = (long)H * W1;
    a.H = H; a.W1 = W1; a.W2 = W2; a.L = L;
    a.pitch = skw_pitch(W1);
    a.nseg = (W1 + 63) / 64;
    for (int i = 0; i < DKT_MAX_LEVELS; ++i) {
        const int wi = i < L ? (W2 >> i) : 0;
        a.inv_wm1[i] = wi > 1 ? (float)(1.0 / (double)(wi - 1)) : 0.0f;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (r) {
        case 0: launch_skew_lookup<0>(a, B, st); break;
        case 1: launch_skew_lookup<1>(a, B, st); break;
        case 2: launch_skew_lookup<2>(a, B, st); break;
        case 3: launch_skew_lookup<3>(a, B, st); break;
        case 4: launch_skew_lookup<4>(a, B, st); break;
        case 5: launch_skew_lookup<5>(a, B, st); break;
        case 6: launch_skew_lookup<6>(a, B, st); break;
        case 7: launch_skew_lookup<7>(a, B, st); break;
        default: launch_skew_lookup<8>(a, B, st); break;
    }
    return dkt_launch_status();
}
