// volumes.hip -- cost-volume builders for gfx950.
//   dkt_gwc_volume     group-wise correlation volume
//   dkt_concat_volume  shifted-copy concatenation volume (GwcNet / IGEV variants)
// Reference: meta_arch/igev_stereo/submodule.py:152-170,207-218,
//            meta_arch/gwcnet/submodules.py:25-58.
// Both are HBM-write-bound (AI ~3 FLOP/B for gwc, 0 for concat): the kernels
// are organised around full-width coalesced plane writes; the target row is
// staged once in LDS so the D shifted re-reads never leave the CU.
#include "dkt_common.h"
#include <cstdlib>

// One block per (b, g, h) row.  LDS: tgt[cpg][W].  Each thread owns pixels
// w = tid, tid+256, ... and for each d forms the group mean
//   sum_j ref[j][w]*tgt[j][w-d] / cpg  (sequential fp32 sum of rounded products,
// the order torch's mean over the strided group axis uses).
#define GWC_MAX_CPG 16
extern __shared__ __attribute__((aligned(16))) float gwc_lds[];

__global__ __launch_bounds__(256) void gwc_volume_kernel(const float *__restrict__ ref,
                                                         const float *__restrict__ tgt,
                                                         float *__restrict__ vol, int C, int H, int W,
                                                         int D, int G, long vol_bstride) {
    const int cpg = C / G;
    const int h = blockIdx.x % H;
    const int g = (blockIdx.x / H) % G;
    const int b = blockIdx.x / (H * G);
    const size_t HW = (size_t)H * W;
    const size_t chan0 = ((size_t)b * C + (size_t)g * cpg) * HW + (size_t)h * W;
    for (int i = threadIdx.x; i < cpg * W; i += 256) {
        int j = i / W, w = i - j * W;
        gwc_lds[i] = tgt[chan0 + (size_t)j * HW + w];
    }
    __syncthreads();
    const float fcpg = (float)cpg;
    float *vrow = vol + (size_t)b * vol_bstride + (size_t)g * D * HW + (size_t)h * W;
    for (int w = threadIdx.x; w < W; w += 256) {
        float r[GWC_MAX_CPG];
#pragma unroll
        for (int j = 0; j < GWC_MAX_CPG; ++j) r[j] = j < cpg ? ref[chan0 + (size_t)j * HW + w] : 0.0f;
        for (int d = 0; d < D; ++d) {
            float out = 0.0f;
            if (w >= d) {
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < GWC_MAX_CPG; ++j)
                    if (j < cpg) s = __fadd_rn(s, __fmul_rn(r[j], gwc_lds[j * W + w - d]));
                out = __fdiv_rn(s, fcpg);
            }
            vrow[(size_t)d * HW + w] = out;
        }
    }
}

// Quad form (the default): a work item is (four consecutive disparities, four consecutive pixels) -> four
// float4 stores.  Both the target and the reference row set of the group sit in LDS (target rows with `lpad`
// floats of slack on the left so that w - d needs no clamp); per channel an item reads ONE aligned quad of
// the reference and TWO aligned quads of the target (the 7 columns its 16 outputs touch) -- three 16-byte
// LDS reads per 16 outputs where the one-pixel form issued one 4-byte read per output.  A thread walks the
// row's D/4 * W/4 items with a stride of 256: every lane is busy whatever W is (the one-pixel form kept 61 %
// of them busy at W = 312 and issued one scalar store per output), the footprint is ~60 registers (8 waves
// per SIMD hide the LDS and store latency), and the sum stays the reference's: s = s + r_j * t_j over the group's channels in
// ascending order, products rounded before they are added, mean by the same division.  Bit-identical.
typedef float gwc_f4 __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) gwc_f4 *gwc_lds_f4p;
template <bool VEC>
__global__ __launch_bounds__(256) void gwc_volume_quad_kernel(const float *__restrict__ ref,
                                                              const float *__restrict__ tgt,
                                                              float *__restrict__ vol, int C, int H, int W,
                                                              int D, int G, long vol_bstride, int lpad) {
    const int cpg = C / G;
    const int h = blockIdx.x % H;
    const int g = (blockIdx.x / H) % G;
    const int b = blockIdx.x / (H * G);
    const size_t HW = (size_t)H * W;
    const size_t chan0 = ((size_t)b * C + (size_t)g * cpg) * HW + (size_t)h * W;
    const int rpitch = (W + 7) & ~3;                                 // reference rows: 16-byte aligned quads
    const int pitch = lpad + rpitch;
    float *rlds = gwc_lds + cpg * pitch;
    if (VEC) {
        // rows are 16-byte aligned (W % 4 == 0): wave k stages channels k, k+4, ...; its lanes walk the row's quads.
        // All of a thread's loads are issued before the first LDS write (one round trip per block).
        const int nq4 = W >> 2, wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        for (int j0 = wv; j0 < cpg; j0 += 16) {
            float4 v[4][2], u[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // clamped (always valid) addresses: the loads stay unconditional, so all 16 are in flight at once
                    const int j = min(j0 + 4 * k, cpg - 1), wq = min(ln + 64 * q, nq4 - 1);
                    const size_t o = chan0 + (size_t)j * HW + 4 * wq;
                    v[k][q] = *(const float4 *)(tgt + o);
                    u[k][q] = *(const float4 *)(ref + o);
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = j0 + 4 * k, wq = ln + 64 * q;
                    if (j < cpg && wq < nq4) {
                        *(float4 *)(gwc_lds + j * pitch + lpad + 4 * wq) = v[k][q];
                        *(float4 *)(rlds + j * rpitch + 4 * wq) = u[k][q];
                    }
                }
            for (int wq = ln + 128; wq < nq4; wq += 64)            // rows wider than 512 pixels: the rest, plainly
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 4 * k;
                    if (j < cpg) {
                        *(float4 *)(gwc_lds + j * pitch + lpad + 4 * wq) = *(const float4 *)(tgt + chan0 + (size_t)j * HW + 4 * wq);
                        *(float4 *)(rlds + j * rpitch + 4 * wq) = *(const float4 *)(ref + chan0 + (size_t)j * HW + 4 * wq);
                    }
                }
        }
    } else {
    const int n_el = cpg * W;
    for (int i0 = threadIdx.x; i0 < n_el; i0 += 256 * 4) {          // both row sets -> LDS, 8 loads in flight
        float v[4], u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            const int j = i / W, w = i - j * W;
            v[k] = (i < n_el) ? tgt[chan0 + (size_t)j * HW + w] : 0.0f;
            u[k] = (i < n_el) ? ref[chan0 + (size_t)j * HW + w] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            const int j = i / W, w = i - j * W;
            if (i < n_el) {
                gwc_lds[j * pitch + lpad + w] = v[k];
                rlds[j * rpitch + w] = u[k];
            }
        }
    }
    }
    __syncthreads();
    const float fcpg = (float)cpg;
    const bool pow2 = (cpg & (cpg - 1)) == 0;                        // such a mean divides exactly by multiplication
    const float rcp = 1.0f / fcpg;
    // (Splitting the disparity quads over grid.y -- more, smaller blocks -- measured slower: 56 / 69 / 116 us with
    // 4 / 6 / 12 blocks per row against 50 us: the row loads are repeated per block.)
    const int nq = (W + 3) / 4, ndq = (D + 3) / 4;
    float *vrow = vol + (size_t)b * vol_bstride + (size_t)g * D * HW + (size_t)h * W;
    for (int item = threadIdx.x; item < nq * ndq; item += 256) {
        const int dq = item / nq, w = 4 * (item - dq * nq), d0 = 4 * dq;
        // target columns (w + i) - (d0 + dd), i, dd in 0..3, span [w - d0 - 3, w - d0 + 3]: read as the two
        // ALIGNED quads starting at w - d0 - 4 (w, d0, lpad multiples of 4) -> t8[0..7], column c at t8[c - (w-d0-4)]
        const float *tp = gwc_lds + lpad + w - d0 - 4;               // >= gwc_lds: d0 + 4 <= lpad
        const float *rp = rlds + w;
        float s[4][4];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) s[dd][0] = s[dd][1] = s[dd][2] = s[dd][3] = 0.0f;
#pragma unroll 4
        for (int j = 0; j < cpg; ++j) {
            // every term of these addresses is a multiple of 4 floats: tell the compiler (ds_read_b128, not 2 x ds_read2_b32)
            const float4 rq = *(const float4 *)__builtin_assume_aligned(rp + j * rpitch, 16);    // (entries past W feed unstored outputs only)
            // (volatile: the compiler would otherwise narrow the two quads to the 7 floats used and read them as 4 pieces)
            // (LDS address space kept explicit: a volatile access through a generic pointer becomes a flat load)
            gwc_lds_f4p tq = (gwc_lds_f4p)(tp + j * pitch);
            const gwc_f4 ta = tq[0], tb = tq[1];
            const float t8[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
            const float rv[4] = {rq.x, rq.y, rq.z, rq.w};
            // scalar multiply then add (never fused: the reference rounds the product first; never packed: DESIGN 3.4)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                for (int i = 0; i < 4; ++i) s[dd][i] = __fadd_rn(s[dd][i], __fmul_rn(rv[i], t8[4 - dd + i]));
        }
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = d0 + dd;
            if (d >= D) break;
            const float sv[4] = {s[dd][0], s[dd][1], s[dd][2], s[dd][3]};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (w + i >= d) ? (pow2 ? __fmul_rn(sv[i], rcp) : __fdiv_rn(sv[i], fcpg)) : 0.0f;
            float *dst = vrow + (size_t)d * HW + w;
            if (VEC && w + 3 < W) {
                *(float4 *)dst = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (w + i < W) dst[i] = o[i];
            }
        }
    }
}

// Wide groups (CGI's single-group normalised correlation: 96 channels per group) on the quad form: the group's rows
// do not fit the LDS at once, so they pass through it in chunks of GWC_CH channels while every thread keeps the
// 16 partial sums of each of its (up to GWC_NI) work items in registers across the chunks.  Same ascending-channel
// summation as the reference.  Needs 16-byte aligned rows (W % 4 == 0) and ceil(W/4) * ceil(D/4) <= 256 * GWC_NI.
#define GWC_CH 16
#define GWC_NI 4
__global__ __launch_bounds__(256) void gwc_volume_quad_chunked_kernel(const float *__restrict__ ref,
                                                                      const float *__restrict__ tgt,
                                                                      float *__restrict__ vol, int C, int H, int W,
                                                                      int D, int G, long vol_bstride, int lpad) {
    const int cpg = C / G;
    const int h = blockIdx.x % H;
    const int g = (blockIdx.x / H) % G;
    const int b = blockIdx.x / (H * G);
    const size_t HW = (size_t)H * W;
    const size_t chan0 = ((size_t)b * C + (size_t)g * cpg) * HW + (size_t)h * W;
    const int rpitch = (W + 7) & ~3;
    const int pitch = lpad + rpitch;
    float *rlds = gwc_lds + GWC_CH * pitch;
    const int nq4 = W >> 2, wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int nq = nq4, ndq = (D + 3) / 4;
    float s[GWC_NI][4][4];
#pragma unroll
    for (int i = 0; i < GWC_NI; ++i)
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) s[i][dd][0] = s[i][dd][1] = s[i][dd][2] = s[i][dd][3] = 0.0f;
    int iw[GWC_NI], id0[GWC_NI];
    bool iok[GWC_NI];
#pragma unroll
    for (int i = 0; i < GWC_NI; ++i) {
        const int item = threadIdx.x + 256 * i;
        iok[i] = item < nq * ndq;
        const int dq = (iok[i] ? item : 0) / nq;
        iw[i] = 4 * ((iok[i] ? item : 0) - dq * nq);
        id0[i] = 4 * dq;
    }
    for (int c0 = 0; c0 < cpg; c0 += GWC_CH) {
        const int nc = min(GWC_CH, cpg - c0);
        // stage channels c0 .. c0+nc-1: wave k takes k, k+4, k+8, k+12; all 16 loads of a thread in flight together
        float4 v[4][2], u[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = min(wv + 4 * k, nc - 1), wq = min(ln + 64 * q, nq4 - 1);
                const size_t o = chan0 + (size_t)(c0 + j) * HW + 4 * wq;
                v[k][q] = *(const float4 *)(tgt + o);
                u[k][q] = *(const float4 *)(ref + o);
            }
        __syncthreads();                                            // the previous chunk has been consumed
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = wv + 4 * k, wq = ln + 64 * q;
                if (j < nc && wq < nq4) {
                    *(float4 *)(gwc_lds + j * pitch + lpad + 4 * wq) = v[k][q];
                    *(float4 *)(rlds + j * rpitch + 4 * wq) = u[k][q];
                }
            }
        for (int wq = ln + 128; wq < nq4; wq += 64)                 // rows wider than 512 pixels
            for (int k = 0; k < 4; ++k) {
                const int j = wv + 4 * k;
                if (j < nc) {
                    *(float4 *)(gwc_lds + j * pitch + lpad + 4 * wq) = *(const float4 *)(tgt + chan0 + (size_t)(c0 + j) * HW + 4 * wq);
                    *(float4 *)(rlds + j * rpitch + 4 * wq) = *(const float4 *)(ref + chan0 + (size_t)(c0 + j) * HW + 4 * wq);
                }
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GWC_NI; ++i) {
            if (!iok[i]) continue;
            const float *tp = gwc_lds + lpad + iw[i] - id0[i] - 4;
            const float *rp = rlds + iw[i];
#pragma unroll 4
            for (int j = 0; j < nc; ++j) {
                const float4 rq = *(const float4 *)__builtin_assume_aligned(rp + j * rpitch, 16);
                gwc_lds_f4p tq = (gwc_lds_f4p)(tp + j * pitch);
                const gwc_f4 ta = tq[0], tb = tq[1];
                const float t8[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
                const float rv[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
                for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[i][dd][e] = __fadd_rn(s[i][dd][e], __fmul_rn(rv[e], t8[4 - dd + e]));
            }
        }
    }
    const float fcpg = (float)cpg;
    const bool pow2 = (cpg & (cpg - 1)) == 0;
    const float rcp = 1.0f / fcpg;
    float *vrow = vol + (size_t)b * vol_bstride + (size_t)g * D * HW + (size_t)h * W;
#pragma unroll
    for (int i = 0; i < GWC_NI; ++i) {
        if (!iok[i]) continue;
        const int w = iw[i];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = id0[i] + dd;
            if (d >= D) break;
            const float sv[4] = {s[i][dd][0], s[i][dd][1], s[i][dd][2], s[i][dd][3]};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (w + k >= d) ? (pow2 ? __fmul_rn(sv[k], rcp) : __fdiv_rn(sv[k], fcpg)) : 0.0f;
            *(float4 *)(vrow + (size_t)d * HW + w) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Groups wider than GWC_MAX_CPG channels (CGI's single-group normalised correlation: cpg = C):
// same block mapping and the same sequential sum order, the reference values pass through
// registers in chunks of 16 channels and the partial sums of a pixel stay in registers.
// grid.y splits the disparities into chunks of GWC_BIG_DCHUNK planes (a single-group volume
// has only B*H rows: one block per row would leave CUs idle).
#define GWC_BIG_DCHUNK 16
__global__ __launch_bounds__(256) void gwc_volume_big_kernel(const float *__restrict__ ref,
                                                             const float *__restrict__ tgt,
                                                             float *__restrict__ vol, int C, int H, int W,
                                                             int D, int G, long vol_bstride) {
    const int cpg = C / G;
    const int h = blockIdx.x % H;
    const int g = (blockIdx.x / H) % G;
    const int b = blockIdx.x / (H * G);
    const size_t HW = (size_t)H * W;
    const size_t chan0 = ((size_t)b * C + (size_t)g * cpg) * HW + (size_t)h * W;
    // one block per CU (the row set fills most of the LDS): the staging loads must overlap each
    // other, so they are issued in batches of 8 independent loads (a plain loop waits per load)
    const int n_el = cpg * W;
    for (int i0 = threadIdx.x; i0 < n_el; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + 256 * k;
            const int j = i / W, w = i - j * W;
            v[k] = i < n_el ? tgt[chan0 + (size_t)j * HW + w] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + 256 * k < n_el) gwc_lds[i0 + 256 * k] = v[k];
    }
    __syncthreads();
    const float fcpg = (float)cpg;
    float *vrow = vol + (size_t)b * vol_bstride + (size_t)g * D * HW + (size_t)h * W;
    const int d0 = blockIdx.y * GWC_BIG_DCHUNK;
    for (int w = threadIdx.x; w < W; w += 256) {
        float s[GWC_BIG_DCHUNK];
#pragma unroll
        for (int d = 0; d < GWC_BIG_DCHUNK; ++d) s[d] = 0.0f;
        // Branch-free inner loops (a branch per product serialises the LDS latency: 420 us
        // instead of ~60): disparities past D or left of the image border are computed on
        // clamped indices and never stored; a partial last channel chunk takes the generic loop.
        for (int j0 = 0; j0 < cpg; j0 += 16) {
            const int nj = min(16, cpg - j0);
            float r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = ref[chan0 + (size_t)(j0 + min(j, nj - 1)) * HW + w];
            if (nj == 16) {
#pragma unroll
                for (int dd = 0; dd < GWC_BIG_DCHUNK; ++dd) {
                    const float *lp = gwc_lds + j0 * W + max(w - (d0 + dd), 0);
#pragma unroll
                    for (int j = 0; j < 16; ++j) s[dd] = __fadd_rn(s[dd], __fmul_rn(r[j], lp[j * W]));
                }
            } else {
#pragma unroll
                for (int dd = 0; dd < GWC_BIG_DCHUNK; ++dd) {
                    const float *lp = gwc_lds + j0 * W + max(w - (d0 + dd), 0);
                    for (int j = 0; j < nj; ++j) s[dd] = __fadd_rn(s[dd], __fmul_rn(r[j], lp[j * W]));
                }
            }
        }
#pragma unroll
        for (int dd = 0; dd < GWC_BIG_DCHUNK; ++dd) {
            const int d = d0 + dd;
            if (d < D) vrow[(size_t)d * HW + w] = w >= d ? __fdiv_rn(s[dd], fcpg) : 0.0f;
        }
    }
}

extern "C" int dkt_gwc_volume(const float *ref, const float *tgt, float *vol,
                              int B, int C, int H, int W, int D, int G, long vol_bstride,
                              int device, void *stream) {
    if (!ref || !tgt || !vol) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0 || G <= 0) return DKT_E_SHAPE;
    if (C % G != 0) return DKT_E_GROUPS;
    const int cpg = C / G;
    const bool big = cpg > GWC_MAX_CPG;
    if (vol_bstride < (long)G * D * H * W) return DKT_E_SHAPE;
    size_t lds = (size_t)cpg * W * sizeof(float);
    if (lds > 160 * 1024) return DKT_E_UNSUPPORTED;
    unsigned long long blocks = (unsigned long long)B * G * H;
    if (blocks > 0x7FFFFFFFull) return DKT_E_SHAPE;
    DKT_ENTER(device);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(big ? (const void *)gwc_volume_big_kernel : (const void *)gwc_volume_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const bool vec16 = W % 4 == 0 && vol_bstride % 4 == 0 && ((((uintptr_t)vol) | ((uintptr_t)ref) | ((uintptr_t)tgt)) & 15) == 0;
    if (big && vec16 && (long)((W + 3) / 4) * ((D + 3) / 4) <= 256L * GWC_NI) {
        const int lpad_c = ((D + 3) & ~3) + 4;
        const size_t lds_c = (size_t)GWC_CH * (lpad_c + 2 * ((W + 7) & ~3)) * sizeof(float);
        if (lds_c <= 64 * 1024) {
            hipLaunchKernelGGL(gwc_volume_quad_chunked_kernel, dim3((unsigned)blocks), dim3(256), lds_c, (hipStream_t)stream,
                               ref, tgt, vol, C, H, W, D, G, vol_bstride, lpad_c);
            return dkt_launch_status();
        }
    }
    if (big)
        hipLaunchKernelGGL(gwc_volume_big_kernel,
                           dim3((unsigned)blocks, (unsigned)((D + GWC_BIG_DCHUNK - 1) / GWC_BIG_DCHUNK)), dim3(256), lds, (hipStream_t)stream,
                           ref, tgt, vol, C, H, W, D, G, vol_bstride);
    else {
        // float4 plane stores need 16-byte aligned rows: W and the batch stride multiples of 4, aligned base
        const bool vec = W % 4 == 0 && vol_bstride % 4 == 0 && ((uintptr_t)vol & 15) == 0;
        const int lpad = ((D + 3) & ~3) + 4;
        const int rp = (W + 7) & ~3;
        const size_t lds_q = (size_t)cpg * (lpad + 2 * rp) * sizeof(float);
        dim3 grid((unsigned)blocks), blk(256);
        hipStream_t st = (hipStream_t)stream;

        if (lds_q > 64 * 1024)
            hipLaunchKernelGGL(gwc_volume_kernel, grid, blk, lds, st, ref, tgt, vol, C, H, W, D, G, vol_bstride);
        else if (vec)
            hipLaunchKernelGGL(gwc_volume_quad_kernel<true>, grid, blk, lds_q, st, ref, tgt, vol, C, H, W, D, G, vol_bstride, lpad);
        else
            hipLaunchKernelGGL(gwc_volume_quad_kernel<false>, grid, blk, lds_q, st, ref, tgt, vol, C, H, W, D, G, vol_bstride, lpad);
    }
    return dkt_launch_status();
}

// One block per (b, c, h) source row; writes the D reference-half rows and the
// D target-half rows that depend on it.  The source rows are read once.
__global__ __launch_bounds__(256) void concat_volume_kernel(const float *__restrict__ ref,
                                                            const float *__restrict__ tgt,
                                                            float *__restrict__ vol, int C, int H, int W,
                                                            int D, int ref_masked, long vol_bstride) {
    const int h = blockIdx.x % H;
    const int c = (blockIdx.x / H) % C;
    const int b = blockIdx.x / (H * C);
    const size_t HW = (size_t)H * W;
    const size_t src = ((size_t)b * C + c) * HW + (size_t)h * W;
    float *oref = vol + (size_t)b * vol_bstride + (size_t)c * D * HW + (size_t)h * W;
    float *otgt = vol + (size_t)b * vol_bstride + (size_t)(C + c) * D * HW + (size_t)h * W;
    for (int w = threadIdx.x; w < W; w += 256) {
        const float rv = ref[src + w];
        for (int d = 0; d < D; ++d) {
            const bool in = w >= d;
            // d >= W: python slices [d:] and [:-d] are empty -> plane stays zero,
            // except the IGEV variant which assigns the whole reference plane.
            oref[(size_t)d * HW + w] = (in || !ref_masked) ? rv : 0.0f;
            otgt[(size_t)d * HW + w] = in ? tgt[src + w - d] : 0.0f;
        }
    }
}

extern "C" int dkt_concat_volume(const float *ref, const float *tgt, float *vol,
                                 int B, int C, int H, int W, int D, int ref_masked, long vol_bstride,
                                 int device, void *stream) {
    if (!ref || !tgt || !vol) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0) return DKT_E_SHAPE;
    if (vol_bstride < 2L * C * D * H * W) return DKT_E_SHAPE;
    unsigned long long blocks = (unsigned long long)B * C * H;
    if (blocks > 0x7FFFFFFFull) return DKT_E_SHAPE;
    DKT_ENTER(device);
    hipLaunchKernelGGL(concat_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       ref, tgt, vol, C, H, W, D, ref_masked ? 1 : 0, vol_bstride);
    return dkt_launch_status();
}
