// gwc_mfma.hip -- the group-wise correlation volume as a banded matrix product on the exact-fp32 matrix pipe
// (north_star: "MFMA only for the GwcNet grouped-correlation which is a true dense contraction").
// Reference: build_gwc_volume / groupwise_correlation, meta_arch/gwcnet/submodules.py:39-58
//            (== meta_arch/igev_stereo/submodule.py:152-170):
//     vol[b, g, d, h, w] = mean_{c in group g} ref[b, c, h, w] * tgt[b, c, h, w - d]     (0 for w < d)
//
// For one (b, g, h) row this is the band  d = w - w' in [0, D)  of  P[w][w'] = sum_c R[c][w] T[c][w'].
// A wave owns 64 consecutive w: four 16 x 16 tiles along w, each against the four 16-wide w' tiles its band touches
// (w' from w0 - 48 up), K = channels per group in steps of 4 on v_mfma_f32_16x16x4_f32 (an exact fp32 fma chain in
// ascending channel order -- within one rounding per product of the reference's sum of rounded products).  The product
// tiles go through LDS transposed to [w][d] (pitch 49: conflict-free writes), are read back as rows of the D planes and
// leave as 16-byte stores of 256 contiguous bytes per plane row -- the volume's 130-330 MB of stores are the kernel's
// real bound, the MFMA work is ~10 us.
//
// The bit-exact VALU kernel (volumes.hip) stays selectable for the C-oracle comparison (dkt_gwc_volume).
#include "dkt_common.h"

typedef float gm_f32x4 __attribute__((ext_vector_type(4)));

#define GM_D 48                   // disparities (band width): 3 tiles of 16
#define GM_PITCH 49               // LDS [w][d] pitch in floats

template <int KS, bool NT>   // channels per group = 4 * KS; NT: streaming (non-temporal) plane stores
__global__ __launch_bounds__(256) void gwc_mfma_kernel(const float *__restrict__ ref, const float *__restrict__ tgt,
                                                       float *__restrict__ vol, int C, int H, int W, int G, long vol_bstride,
                                                       int nwb, long items) {
    __shared__ float lds[4][64 * GM_PITCH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = blockIdx.x * 4L + wave;
    if (item >= items) return;                    // wave-uniform; no block barrier below
    // item = ((b * G + g) * H + h) * nwb + wblock : w blocks of a row are neighbours (one plane row)
    const int wb = (int)(item % nwb);
    long r = item / nwb;
    const int h = (int)(r % H); r /= H;
    const int g = (int)(r % G);
    const int b = (int)(r / G);
    const int w0 = wb * 64;
    constexpr int cpg = 4 * KS;
    const long HW = (long)H * W;
    const float *rrow = ref + ((long)b * C + (long)g * cpg) * HW + (long)h * W;
    const float *trow = tgt + ((long)b * C + (long)g * cpg) * HW + (long)h * W;
    const int i = lane & 15, k = lane >> 4;       // A: A[i][k] = R[c = 4 ks + k][w = tile + i];  B: B[k][j = i] = T[c][w' = tile + i]

    float Rf[4][KS], Tf[7][KS];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int w = w0 + 16 * t + i;
            const float v = rrow[(long)(4 * ks + k) * HW + (w < W ? w : 0)];
            Rf[t][ks] = w < W ? v : 0.0f;
        }
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int wp = w0 - 48 + 16 * t + i;
            const bool ok = wp >= 0 && wp < W;
            const float v = trow[(long)(4 * ks + k) * HW + (ok ? wp : 0)];
            Tf[t][ks] = ok ? v : 0.0f;
        }

    float *my = lds[wave];
    // tile (wt, p): rows w = w0 + 16 wt + (4 q + e), columns w' = w0 - 48 + 16 (wt + p) + j  ->  d = 48 - 16 p + row - col
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int wt = 0; wt < 4; ++wt)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            gm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Rf[wt][ks], Tf[wt + p][ks], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * q + e;
                const int d = 48 - 16 * p + row - j;
                if (d >= 0 && d < GM_D) {
                    // the reference's mean: sum / cpg (a power of two for cpg = 8 / 16: exact either way)
                    my[(16 * wt + row) * GM_PITCH + d] = __fdiv_rn(acc[e], (float)cpg);
                }
            }
        }
    // (wave-private LDS region: program order + the compiler's lgkmcnt waits are the only synchronisation needed)
    float *vrow = vol + (long)b * vol_bstride + (long)g * GM_D * HW + (long)h * W + w0;
#pragma unroll
    for (int it = 0; it < GM_D * 16 / 64; ++it) {
        const int idx = it * 64 + lane;
        const int d = idx >> 4, w4 = (idx & 15) * 4;
        gm_f32x4 v;
        v[0] = my[(w4 + 0) * GM_PITCH + d];
        v[1] = my[(w4 + 1) * GM_PITCH + d];
        v[2] = my[(w4 + 2) * GM_PITCH + d];
        v[3] = my[(w4 + 3) * GM_PITCH + d];
        // W % 4 == 0: a quad is inside or outside.  Volumes larger than the Infinity Cache can hold are streamed past it
        // (GwcNet 251 MB: 70 -> 61 us); smaller ones (IGEV 88 MB) are faster with plain stores (33 vs 36 us)
        if (w0 + w4 < W) {
            if (NT) __builtin_nontemporal_store(v, (gm_f32x4 *)(vrow + (long)d * HW + w4));
            else *(gm_f32x4 *)(vrow + (long)d * HW + w4) = v;
        }
    }
}

// MFMA form: D = 48, channels per group 4 / 8 / 12 / 16, W % 4 == 0, 16-byte aligned volume rows
extern "C" int dkt_gwc_volume_mfma(const float *ref, const float *tgt, float *vol,
                                   int B, int C, int H, int W, int D, int G, long vol_bstride,
                                   int device, void *stream) {
    if (!ref || !tgt || !vol) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0 || G <= 0) return DKT_E_SHAPE;
    if (C % G != 0) return DKT_E_GROUPS;
    const int cpg = C / G;
    if (vol_bstride < (long)G * D * H * W) return DKT_E_SHAPE;
    if (D != GM_D || cpg % 4 != 0 || cpg > 16 || W % 4 != 0 || vol_bstride % 4 != 0 || (((uintptr_t)vol) & 15) != 0)
        return DKT_E_UNSUPPORTED;
    const int nwb = (W + 63) / 64;
    const long items = (long)B * G * H * nwb;
    const long blocks = (items + 3) / 4;
    if (blocks > 0x7fffffffL) return DKT_E_SHAPE;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = (double)B * G * D * H * W * 4.0 > 128.0 * 1024 * 1024;
#define GM_LAUNCH(KS_)                                                                                                          \
    if (nt) hipLaunchKernelGGL((gwc_mfma_kernel<KS_, true>), dim3((unsigned)blocks), dim3(256), 0, st, ref, tgt, vol, C, H, W, G, vol_bstride, nwb, items); \
    else hipLaunchKernelGGL((gwc_mfma_kernel<KS_, false>), dim3((unsigned)blocks), dim3(256), 0, st, ref, tgt, vol, C, H, W, G, vol_bstride, nwb, items);
    switch (cpg / 4) {
    case 1: GM_LAUNCH(1) break;
    case 2: GM_LAUNCH(2) break;
    case 3: GM_LAUNCH(3) break;
    default: GM_LAUNCH(4) break;
    }
#undef GM_LAUNCH
    return dkt_launch_status();
}
