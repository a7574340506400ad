// conv_direct.hip -- direct fp32 convolution for the two extreme layer shapes of the update
// block that a 64-channel-wide MFMA tile wastes:
//   * very few OUTPUT channels: flow_head.conv2 256->2 / disp_head.conv2 256->1, 3x3
//     (core/update.py:10, igev_stereo/update.py:20) -- the MFMA kernel pads 2 -> 64 channels;
//   * very few INPUT channels: convf1 2->64 / convd1 1->64, 7x7 (core/update.py:75,
//     igev_stereo/update.py:81) -- K = 2*49 would be padded to 49 chunks of 32.
// Exact fp32 FMAs on the VALU (no operand splitting), accumulated in (channel, row, column) tap
// order; the input patch of a channel chunk and its weights are staged in LDS.  Optional fused ReLU.
#include "dkt_common.h"
#include <cstdlib>

struct DirectArgs {
    const float *x;
    long x_bs;
    const float *w;      // (Cout, Cin, KS, KS)
    const float *bias;
    float *y;
    long y_bs;
    int Cin, Cout, H, W, tiles_w;
    int relu;
    int accumulate;      // few-output 3x3 form only: y += result (the loop's coords1 += delta in the head's epilogue)
    const float *diff_ref;   // ... and, optionally, diff_out = y_new - diff_ref (the loop's flow = coords1 - coords0)
    float *diff_out;
    long diff_ref_bs, diff_out_bs;
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

// ---------------------------------------------------------------------------------------------------------
// 3x3 convolution with very few output channels (flow_head.conv2 256 -> 2, disp_head.conv2 256 -> 1): the
// layer is a 58.8 MB read for 0.5 GFLOP -- HBM-bound -- and the per-iteration tail of the update operator.
//   block  = one 4-row x 64-column output tile (230 blocks at 184 x 312: every CU gets one), FEW_NW = 8 waves;
//   wave w = input channels [w*Cin/8, (w+1)*Cin/8) in chunks of FEW_CC, lane = 4 adjacent pixels of one row;
//   patch  = (4+2) rows x 72 columns per channel (image columns w0-4 .. w0+67: 16-byte groups), copied global ->
//            LDS by the DMA path (global_load_lds_dwordx4: no registers, no VALU; a patch row is 18 lanes, so two
//            instructions move a channel's 6 rows; W % 4 != 0 falls back to 4-byte pieces), double-buffered and
//            WAVE-PRIVATE: no block barrier in the channel loop.  Cells outside the image are zero-filled once
//            and never written again (their lanes are masked off);
//   math   = exact fp32 FMAs; the wave's weights sit in LDS (staged once) and are read as broadcasts; a patch row
//            is one 16-byte and two 4-byte LDS reads per lane;
//   the eight channel slices are summed through LDS in wave order, bias / ReLU in the epilogue.
// ---------------------------------------------------------------------------------------------------------
#if defined(FEW_DEBUG) || defined(FEW_DEBUG2)
// tools/stress_lds_dma.py instrumentation (never in the product build): [0] DMA'd 16-byte groups that differ from
// global memory, [1..5] LDS_ALLOC / wave / chunk / piece / lane of the first one, [6] weight words that differ,
// [7] masked groups that are not zero, [8] blocks with LDS base != 0, [9] blocks, [10] mismatching groups found
// intact 1 KiB further / [11] elsewhere in the wave's buffers
__device__ int few_dbg[16];
extern "C" int dkt_debug_few_counters(int *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(few_dbg), sizeof(int) * 16) != hipSuccess) return -1;
    if (reset) { int z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(few_dbg), z, sizeof z); }
    return 0;
}
#endif
#define FEW_CC 4
#define FEW_NW 8                                    // waves per block = channel slices
#define FEW_PR 6
#define FEW_PCP 72                                   // patch row: image columns w0-4 .. w0+67 (16-byte aligned groups)
template <int TO>
__global__ __launch_bounds__(64 * FEW_NW) void conv3x3_few_kernel(DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) float few_lds[];   // [FEW_NW waves][2][FEW_CC][FEW_PR][FEW_PCP] + weights
    constexpr int PLANE = FEW_PR * FEW_PCP, BUF = FEW_CC * PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w0 = (blockIdx.x % a.tiles_w) * 64, h0 = (blockIdx.x / a.tiles_w) * 4;
    const int b = blockIdx.z;
    const long HW = (long)a.H * a.W;
    const float *xb = a.x + (long)b * a.x_bs;
    float *mybuf = few_lds + wave * 2 * BUF;
    // zero both buffers once: cells outside the image are never written by the DMA below
    for (int i = lane; i < 2 * BUF / 4; i += 64) ((float4 *)mybuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int cq = (a.Cin + FEW_NW - 1) / FEW_NW;
    const int c_lo = wave * cq, c_hi = min(a.Cin, c_lo + cq);
    // this wave's weights -> LDS once ([channel][o][12]: 9 taps padded to three 16-byte groups), read back as broadcasts
    float *wl = few_lds + FEW_NW * 2 * BUF + wave * cq * TO * 12;
    for (int i = lane; i < (c_hi - c_lo) * TO * 9; i += 64) {
        const int t = i % 9, o = (i / 9) % TO, c = i / (9 * TO);
        wl[(c * TO + o) * 12 + t] = o < a.Cout ? a.w[((long)o * a.Cin + c_lo + c) * 9 + t] : 0.0f;
    }
    const int nchunk = (c_hi - c_lo + FEW_CC - 1) / FEW_CC;
    // DMA of one chunk.  LDS destination of a DMA instruction = wave-uniform base + lane * size.
    //   VEC (W % 4 == 0, 16-byte aligned planes): lane l of instruction j covers the 16-byte group g = 64 j + l of
    //   the channel's 6 x 18 groups: patch row g / 18, columns 4 (g % 18) .. +3  -- two instructions per channel;
    //   otherwise: per patch row one 64-lane and one 8-lane instruction of 4 bytes per lane.
    const bool vec = (a.W & 3) == 0 && (((uintptr_t)a.x | (uintptr_t)(a.x_bs * 4)) & 15) == 0;
    int g_row[2], g_off[2];
    bool g_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = 64 * j + lane;
        const int r = g / 18, c4 = 4 * (g - 18 * r);
        const int ih = h0 - 1 + r, iw = w0 - 4 + c4;
        g_ok[j] = g < FEW_PR * 18 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;      // W % 4 == 0: a group is all in or all out
        g_row[j] = r;
        g_off[j] = g_ok[j] ? ih * a.W + iw : 0;
    }
    const int iw_a = w0 - 4 + lane, iw_b = w0 + 60 + lane;
    const bool col_a = iw_a >= 0 && iw_a < a.W, col_b = lane < 8 && iw_b < a.W;
    auto stage = [&](int chunk, int which) {
        float *dst = mybuf + which * BUF;
#pragma unroll
        for (int c = 0; c < FEW_CC; ++c) {
            const int ch = c_lo + chunk * FEW_CC + c;
            if (ch >= c_hi) break;                                   // wave-uniform
            const float *xc = xb + (long)ch * HW;
            if (vec) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (g_ok[j])
                        __builtin_amdgcn_global_load_lds(xc + g_off[j], (__attribute__((address_space(3))) void *)(dst + c * PLANE + 256 * j), 16, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < FEW_PR; ++r) {
                    const int ih = h0 - 1 + r;
                    if (ih < 0 || ih >= a.H) continue;               // wave-uniform: the row stays zero
                    const float *row = xc + (long)ih * a.W;
                    float *lrow = dst + c * PLANE + r * FEW_PCP;
                    if (col_a) __builtin_amdgcn_global_load_lds(row + iw_a, (__attribute__((address_space(3))) void *)lrow, 4, 0, 0);
                    if (col_b) __builtin_amdgcn_global_load_lds(row + iw_b, (__attribute__((address_space(3))) void *)(lrow + 64), 4, 0, 0);
                }
            }
        }
    };
    const int prow = lane >> 4, q4 = 4 * (lane & 15);                // this lane's output row and first column
    float acc[4][TO];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < TO; ++o) acc[p][o] = 0.0f;
#ifdef FEW_DEBUG2
    unsigned dbg_bad = 0;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the zero fill has landed before the first DMA
    if (nchunk > 0) stage(0, 0);
    for (int k = 0; k < nchunk; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // chunk k is in LDS
#ifdef FEW_DEBUG
        if (vec) {
            const float *chk = mybuf + (k & 1) * BUF;
            for (int c = 0; c < FEW_CC; ++c) {
                const int ch = c_lo + k * FEW_CC + c;
                if (ch >= c_hi) break;
                const float *xc = xb + (long)ch * HW;
                for (int j = 0; j < 2; ++j) {
                    const int g = 64 * j + lane;
                    if (g >= FEW_PR * 18) continue;
                    const float4 l4 = *(const float4 *)(chk + c * PLANE + 256 * j + lane * 4);
                    if (g_ok[j]) {
                        const float4 g4 = *(const float4 *)(xc + g_off[j]);
                        if (l4.x != g4.x || l4.y != g4.y || l4.z != g4.z || l4.w != g4.w) {
                            if (atomicAdd(&few_dbg[0], 1) == 0) {
                                few_dbg[1] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
                                few_dbg[2] = wave; few_dbg[3] = k; few_dbg[4] = c * 2 + j; few_dbg[5] = lane;
                            }
                            // did the group land somewhere else in this wave's two buffers?
                            bool found = false;
                            for (int q = 0; q < 2 * BUF / 4 && !found; ++q) {
                                const float4 t = ((const float4 *)mybuf)[q];
                                if (t.x == g4.x && t.y == g4.y && t.z == g4.z && t.w == g4.w && (g4.x != 0.f || g4.y != 0.f)) found = true;
                            }
                            atomicAdd(&few_dbg[found ? 11 : 10], 1);
                        }
                    } else if (l4.x != 0.f || l4.y != 0.f || l4.z != 0.f || l4.w != 0.f) atomicAdd(&few_dbg[7], 1);
                }
            }
        }
        if (k == 0) {
            for (int i = lane; i < (c_hi - c_lo) * TO * 9; i += 64) {
                const int t = i % 9, o = (i / 9) % TO, c = i / (9 * TO);
                const float e = o < a.Cout ? a.w[((long)o * a.Cin + c_lo + c) * 9 + t] : 0.0f;
                if (wl[(c * TO + o) * 12 + t] != e) atomicAdd(&few_dbg[6], 1);
            }
            if (tid == 0) {
                atomicAdd(&few_dbg[9], 1);
                if (__builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11)) & 0xfff) atomicAdd(&few_dbg[8], 1);
            }
        }
#endif
        if (k + 1 < nchunk) stage(k + 1, (k + 1) & 1);             // chunk k+1 streams in under the FMAs of chunk k
        const float *src = mybuf + (k & 1) * BUF;
#pragma unroll
        for (int c = 0; c < FEW_CC; ++c) {
            const int ch = c_lo + k * FEW_CC + c;
            if (ch >= c_hi) break;                                   // wave-uniform
            float wv[TO][12];                                        // same address in every lane: LDS broadcast reads
#pragma unroll
            for (int o = 0; o < TO; ++o) {
                const float4 *wp = (const float4 *)(wl + ((ch - c_lo) * TO + o) * 12);
                const float4 w0_ = wp[0], w1_ = wp[1], w2_ = wp[2];
                wv[o][0] = w0_.x; wv[o][1] = w0_.y; wv[o][2] = w0_.z; wv[o][3] = w0_.w;
                wv[o][4] = w1_.x; wv[o][5] = w1_.y; wv[o][6] = w1_.z; wv[o][7] = w1_.w;
                wv[o][8] = w2_.x;
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // image columns q4-1 .. q4+4 of the tile = patch columns q4+3 .. q4+8
                const float *pr = src + c * PLANE + (prow + dy) * FEW_PCP + q4 + 4;
                const float4 v4 = *(const float4 *)pr;
                const float v[6] = {pr[-1], v4.x, v4.y, v4.z, v4.w, pr[4]};
#ifdef FEW_DEBUG2
                if (dy == 2) {   // light check (keeps the VGPR count of the product): patch rows 2..5 as read vs global memory
                    const int ih = h0 - 1 + prow + dy;
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        const int iw = w0 + q4 - 1 + e;
                        const float g = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? xb[(long)ch * HW + (long)ih * a.W + iw] : 0.0f;
                        dbg_bad |= (g != v[e]) ? (1u << e) : 0u;
                    }
                }
#endif
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int o = 0; o < TO; ++o) acc[p][o] = __fmaf_rn(v[p + dx], wv[o][dy * 3 + dx], acc[p][o]);
            }
        }
    }
#ifdef FEW_DEBUG2
    if (dbg_bad) {
        if (atomicAdd(&few_dbg[12], 1) == 0) { few_dbg[13] = (lane << 16) | (dbg_bad << 4) | wave; few_dbg[14] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11)); }
    }
    if (tid == 0) { atomicAdd(&few_dbg[9], 1); if (__builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11)) & 0xfff) atomicAdd(&few_dbg[8], 1); }
#endif
    // ---- sum the four channel quarters (wave order), bias, ReLU, store
    __syncthreads();
    float *part = few_lds;                                           // [FEW_NW waves][TO][4 px][64 lanes]
#pragma unroll
    for (int o = 0; o < TO; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) part[((wave * TO + o) * 4 + p) * 64 + lane] = acc[p][o];
    __syncthreads();
    if (wave != 0) return;
    const int oh = h0 + prow;
    if (oh >= a.H) return;
#pragma unroll
    for (int o = 0; o < TO; ++o) {
        if (o >= a.Cout) break;
        const float bias = a.bias ? a.bias[o] : 0.0f;
        float *yr = a.y + (long)b * a.y_bs + (long)o * HW + (long)oh * a.W + w0 + q4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (w0 + q4 + p >= a.W) continue;
            float v = part[((0 * TO + o) * 4 + p) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < FEW_NW; ++ww) v = __fadd_rn(v, part[((ww * TO + o) * 4 + p) * 64 + lane]);
            v = __fadd_rn(v, bias);
            if (a.relu) v = dkt_relu(v);
            v = a.accumulate ? __fadd_rn(yr[p], v) : v;
            yr[p] = v;
            if (a.diff_out) {
                const long off = (long)o * HW + (long)oh * a.W + w0 + q4 + p;
                a.diff_out[(long)b * a.diff_out_bs + off] = __fsub_rn(v, a.diff_ref[(long)b * a.diff_ref_bs + off]);
            }
        }
    }
}

template <int TO>
static int launch_few(DirectArgs a, int B, hipStream_t st) {
    const int cq = (a.Cin + FEW_NW - 1) / FEW_NW;
    const size_t need = ((size_t)FEW_NW * 2 * FEW_CC * FEW_PR * FEW_PCP + (size_t)FEW_NW * cq * TO * 12) * sizeof(float);
    if (need > 160 * 1024) return DKT_E_UNSUPPORTED;
    // The block asks for the CU's WHOLE LDS (160 KB), whatever it needs, so that no LDS-holding block of another kernel
    // shares its CU.  History: the TO = 1 form needs 120 KB, which leaves room for a 33 KB block of the
    // 1/16-resolution GRU convolution on the second stream, and in that situation the SLP-vectorised build of this
    // kernel (v_pk_fma_f32 chains) returned wrong sums -- 1000 of 1200 launches in tools/stress_lds_dma.py.  Round 3
    // established that the LDS-DMA copies are NOT at fault (every copied 16-byte group verified in place under the
    // failing co-residency; tools/lds_dma_probe*.hip: exact for every size / offset / allocation base); the wrong
    // values are the low halves of the packed accumulator pairs in lanes 48..63, and the scalar-FMA build
    // (-fno-slp-vectorize, dkt_stereo_amd/build.py) is exact in 1200 of 1200 launches beside 5000+ co-resident blocks.
    // The file is therefore built without SLP vectorisation AND the exclusive LDS claim stays as a second protection
    // (kernels on different streams do not overlap usefully on this part anyway, tools/concurrency_probe.py).
    // DKT_FEW_LDS_EXACT=1 (tools/stress_lds_dma.py only): request just `need`, i.e. allow LDS-holding co-residents
    static const bool exact = [] { const char *e = getenv("DKT_FEW_LDS_EXACT"); return e && atoi(e) != 0; }();
    const size_t lds = exact ? need : 160 * 1024;
    auto kern = conv3x3_few_kernel<TO>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    a.tiles_w = (a.W + 63) / 64;
    dim3 grid((unsigned)(a.tiles_w * ((a.H + 3) / 4)), 1, (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * FEW_NW), lds, st, a);
    return dkt_launch_status();
}

template <int KS, int TO, int CC>
static int launch_direct(const DirectArgs &a, int B, hipStream_t st) {
    const int tiles_h = (a.H + 15) / 16;
    dim3 grid((unsigned)(a.tiles_w * tiles_h), (unsigned)((a.Cout + TO - 1) / TO), (unsigned)B);
    hipLaunchKernelGGL((conv2d_direct_kernel<KS, TO, CC>), grid, dim3(256), 0, st, a);
    return dkt_launch_status();
}

static int conv2d_direct_impl(const float *x, long x_bstride, const float *w, const float *bias,
                              float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                              int KH, int KW, int relu, int accumulate, int device, void *stream,
                              const float *diff_ref = nullptr, long diff_ref_bs = 0, float *diff_out = nullptr, long diff_out_bs = 0);

extern "C" int dkt_conv2d_direct(const float *x, long x_bstride, const float *w, const float *bias,
                                 float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                                 int KH, int KW, int relu, int device, void *stream) {
    return conv2d_direct_impl(x, x_bstride, w, bias, y, y_bstride, B, Cin, Cout, H, W, KH, KW, relu, 0, device, stream);
}

extern "C" int dkt_conv2d_direct_accumulate(const float *x, long x_bstride, const float *w, const float *bias,
                                            float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                                            int KH, int KW, int device, void *stream) {
    if (KH != 3 || Cout > 4) return DKT_E_UNSUPPORTED;
    return conv2d_direct_impl(x, x_bstride, w, bias, y, y_bstride, B, Cin, Cout, H, W, KH, KW, 0, 1, device, stream);
}

extern "C" int dkt_conv2d_direct_accumulate_diff(const float *x, long x_bstride, const float *w, const float *bias,
                                                 float *y, long y_bstride, const float *diff_ref, long diff_ref_bstride,
                                                 float *diff_out, long diff_out_bstride, int B, int Cin, int Cout,
                                                 int H, int W, int KH, int KW, int device, void *stream) {
    if (KH != 3 || Cout > 4) return DKT_E_UNSUPPORTED;
    if (!diff_ref || !diff_out) return DKT_E_NULL;
    return conv2d_direct_impl(x, x_bstride, w, bias, y, y_bstride, B, Cin, Cout, H, W, KH, KW, 0, 1, device, stream,
                              diff_ref, diff_ref_bstride, diff_out, diff_out_bstride);
}

static int conv2d_direct_impl(const float *x, long x_bstride, const float *w, const float *bias,
                              float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                              int KH, int KW, int relu, int accumulate, int device, void *stream,
                              const float *diff_ref, long diff_ref_bs, float *diff_out, long diff_out_bs) {
    if (!x || !w || !y) return DKT_E_NULL;
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || B > 65535) return DKT_E_SHAPE;
    if (KH != KW || (KH != 3 && KH != 7)) return DKT_E_UNSUPPORTED;
    DirectArgs a;
    a.x = x; a.x_bs = x_bstride; a.w = w; a.bias = bias; a.y = y; a.y_bs = y_bstride;
    a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.tiles_w = (W + 63) / 64;
    a.relu = relu ? 1 : 0;
    a.accumulate = accumulate ? 1 : 0;
    a.diff_ref = diff_ref; a.diff_out = diff_out; a.diff_ref_bs = diff_ref_bs; a.diff_out_bs = diff_out_bs;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    if (KH == 3) {
        if (Cout > 4) return DKT_E_UNSUPPORTED;       // wide layers belong to dkt_conv2d_f16s
        if (Cout <= 1) return launch_few<1>(a, B, st);
        if (Cout <= 2) return launch_few<2>(a, B, st);
        return launch_few<4>(a, B, st);
    }
    if (Cin > 4) return DKT_E_UNSUPPORTED;
    return launch_direct<7, 16, 2>(a, B, st);
}
