// conv2d.hip -- implicit-GEMM 2-D convolution for the update operator on gfx950.
//
// The ConvGRU / motion-encoder / head convolutions are ~98 % of a RAFT-Stereo
// iteration (core/update.py:19-21, 72-76, 9-10, 111-113).  fp32 MFMA peaks at
// 157 TFLOP/s on MI355X, 1/16 of the fp16 rate, so this kernel evaluates the
// fp32 convolution on the fp16 matrix cores with SPLIT operands:
//     w*s = w_hi + w_lo,  x = x_hi + x_lo   (each part fp16, s a power of two)
//     w*x*s ~= w_hi*x_hi + w_lo*x_hi + w_hi*x_lo      (fp32 accumulation)
// which keeps ~22 significant bits per operand (fp32 has 24) at three
// v_mfma_f32_32x32x16_f16 per product block: 833 TFLOP/s-equivalent peak, 5.3x
// the fp32 pipe.  PASSES = 3 is that form (measured 1.5e-6 relative to an fp64
// convolution, the vendor fp32 path measures 0.9e-6); PASSES = 2 drops
// w_hi*x_lo (activations rounded to fp16, weights still split); PASSES = 1 is
// plain fp16.
//
// GEMM view   D[co][px] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][px + off(tap)]
//   A = weights, pre-packed once per layer to [tap][ci/16][co][16] fp16 so a
//       fragment (lane l: A[co = l&31][k = 8*(l>>5)..+7]) is one coalesced
//       16-byte load per lane straight from L2 -- no LDS for weights; loads run
//       two (tap, k16) steps ahead of the MFMAs that consume them;
//   B = activations: NCHW fp32 in HBM (possibly several tensors = the
//       reference's torch.cat operands, read in place), staged per 32-channel
//       chunk as a (rows+halo) x (32+halo) pixel patch in LDS, converted to
//       fp16 hi/lo and transposed to [pixel][channel] (pitch 80 B: conflict-
//       free ds_read_b128 for the fragment lane l: B[k = 8*(l>>5)..+7][px = l&31]);
//       each staged patch is reused by all taps and ALL output channels of the
//       block (a block spans up to 256 output channels, so a patch is staged once);
//   D accumulators keep pixels along lanes (C/D map col = l&31), so output
//       stores are 128-byte NCHW rows.
// Block = 4 waves, one per SIMD (the kernel wants the whole register file: up to
// 128 accumulators + three weight-fragment sets in flight).  Wave tile = 64 output
// channels x NF rows x 32 columns; WM x WN waves along (channels, rows).  LDS is
// double buffered: the fp32 loads of chunk c+1 are in flight under the MFMAs of
// chunk c; one barrier per chunk.
#include "dkt_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CONV_MAX_SRC 4
// scheduling tunables (swept with tools/sweep_conv_variants.sh)
#ifndef CONV_AR
#define CONV_AR 3
#endif
#ifndef CONV_SGB_MFMA
#define CONV_SGB_MFMA 1
#endif
#ifndef CONV_SGB_MEM
#define CONV_SGB_MEM 1
#endif
#ifndef CONV_SGB_VALU
#define CONV_SGB_VALU 3
#endif

struct ConvArgs {
    const float *src[CONV_MAX_SRC];
    long src_bs[CONV_MAX_SRC];
    int src_ch[CONV_MAX_SRC];
    int nsrc;
    const _Float16 *whi;
    const _Float16 *wlo;
    const float *bias;
    float out_scale;  // 1/s
    float *out;
    long out_bs;
    int H, W, Cout, CoutPad, nch16, tiles_w;
    int relu;
    // fused ConvGRU gate epilogues (core/update.py:27-31), epi = 0 none;
    //   1: merged z|r convolution, Cout = 2*Ch:  co <  Ch: z  = sigmoid(v + cz)        -> out  (z)
    //                                            co >= Ch: rh = sigmoid(v + cr) * h    -> out2 (r*h)
    //   2: q convolution, Cout = Ch:             h' = (1 - z)*h + z*tanh(v + cq)        -> out  (may alias h)
    int epi;
    const float *e_c0;   // cz (epi 1) / cq (epi 2)
    const float *e_c1;   // cr (epi 1) / z  (epi 2)
    const float *e_h;    // h
    long e_c0_bs, e_c1_bs, e_h_bs;
    float *out2;
    long out2_bs;
};

// Gate non-linearities for the fused epilogues.  The epilogue runs on the VALU after the
// MFMA loop with nothing to overlap it, so it uses the hardware transcendentals
// (v_exp_f32 / v_rcp_f32, ~1 ulp each: absolute error <= 3e-7 on (0,1) / (-1,1)) rather than
// the ~45-instruction correctly-rounded forms the streaming gate kernels afford
// (measured: precise forms cost +7.7 ms per pair, these -x ms).
__device__ __forceinline__ float conv_sigmoid(float x) {
    return __frcp_rn(1.0f + __expf(-x));
}
__device__ __forceinline__ float conv_tanh(float x) {
    const float t = __expf(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
    return (t - 1.0f) * __frcp_rn(t + 1.0f);
}

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; unsigned u; } v;
    v.h[0] = a;
    v.h[1] = b;
    return v.u;
}

// ABL: timing-only ablation mask (results are wrong when non-zero): 1 = no weight loads in
// the loop, 2 = no LDS fragment reads, 4 = no staging, 8 = no MFMAs.  See tools/bench_kernels.py.
template <int KS, int WM, int WN, int NF, int PASSES, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv2d_f16s_kernel(ConvArgs a) {
    constexpr int HALO = KS / 2;
    constexpr int TR = NF * WN;              // output rows per block
    constexpr int PR = TR + 2 * HALO;        // patch rows
    constexpr int PC = 32 + 2 * HALO;        // patch cols
    constexpr int NPP = PR * PC;             // patch pixels
    constexpr int PITCH = 20;                // 32-bit words per pixel: 16 (32 fp16) + 4 pad
    constexpr int PLANE = NPP * PITCH;       // words per (hi or lo) plane
    constexpr int NPLANES = PASSES == 3 ? 2 : 1;   // x_lo is only needed for the w_hi*x_lo pass
    constexpr int STAGE = PLANE * NPLANES;
    constexpr int NITEMS = NPP * 16;         // (pixel, channel pair) items per chunk
    constexpr int IT = (NITEMS + 255) / 256;
    constexpr int NSTEP = KS * KS * 2;       // (tap, 16-channel half) steps per chunk
    constexpr int AR = (NSTEP % 3 == 0 && CONV_AR == 3) ? 3 : 2;   // weight-fragment ring (prefetch distance AR-1)
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // 2 * STAGE words

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int tile = blockIdx.x;
    const int w0 = (tile % a.tiles_w) * 32;
    const int h0 = (tile / a.tiles_w) * TR;
    const int co_blk = blockIdx.y * (64 * WM);
    const int b = blockIdx.z;
    const long HW = (long)a.H * a.W;

    // ---- staging: item = (patch pixel pp, channel pair cp), item = tid + 256*it.
    // The (pp, cp) -> address arithmetic is recomputed per chunk (a few VALU ops per
    // item against ~14k MFMA cycles per chunk) rather than held in 2*IT registers.
    // Both halves are BRANCH-FREE (clamped addresses + selects, surplus items write a
    // dummy LDS word) and take an item range, so that slices of them can sit inside
    // the MFMA steps of the previous chunk and be interleaved with the MFMAs.
    float2 sreg[IT];
    const float *sbase = nullptr;   // source plane of the chunk being staged (wave-uniform)
    int snch = 0;
    auto stage_select = [&](int chunk) {
        int s = 0, c0 = chunk * 32;
        while (s + 1 < a.nsrc && c0 >= ((a.src_ch[s] + 31) & ~31)) {
            c0 -= (a.src_ch[s] + 31) & ~31;
            ++s;
        }
        sbase = a.src[s] + (long)b * a.src_bs[s] + (long)c0 * HW;
        snch = a.src_ch[s] - c0;    // valid channels from c0 on (may exceed 32)
    };
    auto stage_load = [&](int it0, int it1) {
        int t = tid;
        asm volatile("" : "+v"(t));          // opaque: stops the plan being hoisted into live registers
#pragma unroll
        for (int it = it0; it < it1; ++it) {
            if (it >= IT) break;
            const int item = t + 256 * it;
            const int cp = item / NPP, pp = item - cp * NPP;
            const int pr = pp / PC, pc = pp - pr * PC;
            const int ih = h0 - HALO + pr, iw = w0 - HALO + pc;
            const bool ok = item < NITEMS && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            const int ch = 2 * cp;
            const bool ok0 = ok && ch < snch, ok1 = ok && ch + 1 < snch;
            const int off = ok ? ih * a.W + iw : 0;
            const float l0 = sbase[(long)(ok0 ? ch : 0) * HW + off];
            const float l1 = sbase[(long)(ok1 ? ch + 1 : 0) * HW + off];
            sreg[it] = make_float2(ok0 ? l0 : 0.0f, ok1 ? l1 : 0.0f);
        }
    };
    auto stage_store = [&](unsigned *buf, int it0, int it1) {
        int t = tid;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int it = it0; it < it1; ++it) {
            if (it >= IT) break;
            const int item = t + 256 * it;
            const int cp = item / NPP, pp = item - cp * NPP;
            unsigned *dst = item < NITEMS ? buf + pp * PITCH + cp : lds + 2 * STAGE;   // surplus -> dummy
            float x0 = fminf(fmaxf(sreg[it].x, -65504.0f), 65504.0f);
            float x1 = fminf(fmaxf(sreg[it].y, -65504.0f), 65504.0f);
            const _Float16 h0_ = (_Float16)x0, h1_ = (_Float16)x1;
            dst[0] = pack_h2(h0_, h1_);
            if (NPLANES == 2) {
                const _Float16 l0 = (_Float16)(x0 - (float)h0_), l1 = (_Float16)(x1 - (float)h1_);
                dst[item < NITEMS ? PLANE : 1] = pack_h2(l0, l1);
            }
        }
    };

    f32x16 acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int li = lane & 31, kg = lane >> 5;
    const int co_w = co_blk + wm * 64;                 // this wave's first output channel
    const bool wave_on = co_w < a.CoutPad;             // waves past the (64-padded) channel count idle
    // weight fragment base (halves): ((tap*nch16 + ch16)*CoutPad + co)*16 + kg*8
    const long wlane = ((long)((wave_on ? co_w : 0) + li)) * 16 + kg * 8;
    const long wstep = (long)a.CoutPad * 16;           // one 16-channel slab
    const int nchunks = a.nch16 / 2;

    f16x8 Ahi[AR][2], Alo[AR][2];
    f16x8 Bhi[2][NF], Blo[2][NF];
    auto loadA = [&](int slot, int chunk, int step) {
        const int tap = step >> 1, kh = step & 1;
        const long wbase = ((long)tap * a.nch16 + (chunk * 2 + kh)) * wstep + wlane;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            Ahi[slot][m] = *(const f16x8 *)(a.whi + wbase + m * 32 * 16);
            if (PASSES >= 2) Alo[slot][m] = *(const f16x8 *)(a.wlo + wbase + m * 32 * 16);
        }
    };
    auto loadB = [&](int slot, const unsigned *buf, int step) {
        const int tap = step >> 1, kh = step & 1;
        const int dy = tap / KS, dx = tap % KS;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int pp = (wn * NF + n + dy) * PC + li + dx;
            const unsigned *pb = buf + pp * PITCH + kh * 8 + kg * 4;
            Bhi[slot][n] = *(const f16x8 *)pb;
            if (NPLANES == 2) Blo[slot][n] = *(const f16x8 *)(pb + PLANE);
        }
    };
    // Pass-major order: every accumulator is touched once per pass, so two MFMAs on the
    // same accumulator are 2*NF issues apart.  (Pass-minor order interleaves two
    // dependent chains one MFMA apart: measured 41 % issue-stall, MFMA pipe 46 % busy.)
    auto mma = [&](int as, int bs) {
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[as][m], Bhi[bs][n], acc[m][n], 0, 0, 0);
        if (PASSES >= 2) {
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[as][m], Bhi[bs][n], acc[m][n], 0, 0, 0);
        }
        if (PASSES == 3) {
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[as][m], Blo[bs][n], acc[m][n], 0, 0, 0);
        }
    };

    // One wave per SIMD issues in order, so MFMA time and everything else ADD unless the
    // other instructions sit between MFMAs in program order (measured on the 384->256
    // layer: MFMAs alone 237 us, data movement alone 122 us, serial form 392 us).  Each
    // (tap, k16) step therefore carries, besides its 8*NF*PASSES MFMAs: the weight loads of
    // step s+AR-1, the LDS fragment reads of step s+1 and one slice of the NEXT chunk's
    // staging (fp32 loads in the first half of the steps, convert + LDS writes in the
    // second half); a sched_group_barrier pattern asks for ~1 memory op + 3 VALU per MFMA.
    // Idle waves (co >= CoutPad) run the same stream on clamped weights and store nothing:
    // a wave-uniform branch here would split the scheduling region.
    constexpr int HALF = NSTEP / 2;
    constexpr int PER = (IT + HALF - 1) / HALF;
    constexpr int NMMA = 2 * NF * PASSES;
    stage_select(0);
    stage_load(0, IT);
    stage_store(lds, 0, IT);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < AR - 1; ++s) loadA(s, 0, s);
    loadB(0, lds, 0);
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        const unsigned *cur = lds + (c & 1) * STAGE;
        unsigned *nxt = lds + ((c + 1) & 1) * STAGE;
        stage_select(more ? c + 1 : c);      // last chunk re-stages itself into the idle buffer (harmless)
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            // weights for step s+AR-1 (possibly the next chunk's first steps)
            const int sa = s + AR - 1;
            if (!(ABL & 1)) {
                if (sa < NSTEP) loadA(sa % AR, c, sa);
                else loadA(sa % AR, more ? c + 1 : c, sa - NSTEP);
            }
            if (s + 1 < NSTEP && !(ABL & 2)) loadB((s + 1) & 1, cur, s + 1);
            if (!(ABL & 4)) {
                if (s < HALF) stage_load(s * PER, (s + 1) * PER);
                else stage_store(nxt, (s - HALF) * PER, (s - HALF + 1) * PER);
            }
            if (!(ABL & 8)) mma(s % AR, s & 1);
            else {
#pragma unroll
                for (int m = 0; m < 2; ++m) asm volatile("" ::"v"(Ahi[s % AR][m]), "v"(Alo[s % AR][m]));
#pragma unroll
                for (int n = 0; n < NF; ++n) asm volatile("" ::"v"(Bhi[s & 1][n]), "v"(Blo[s & 1][n]));
            }
            if (!(ABL & 16)) {
#pragma unroll
                for (int i = 0; i < NMMA / CONV_SGB_MFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, CONV_SGB_MFMA, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x320, CONV_SGB_MEM, 0);    // VMEM read / DS read / DS write
                    __builtin_amdgcn_sched_group_barrier(0x002, CONV_SGB_VALU, 0);   // VALU
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more && !(ABL & 2)) loadB(0, nxt, 0);      // NSTEP is even: step 0 always uses B slot 0
    }

    // ---- epilogue: un-scale, bias, optional ReLU, NCHW stores (128-byte rows) ----
    if (!wave_on) return;
    // this lane's 32 output channels: co = co_lane + m*32 + (r&3) + 8*(r>>2); their biases
    // are fetched as one batch of independent loads (clamped index, no per-element branch)
    const int co_lane = co_w + 4 * kg;
    float bv[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_lane + m * 32 + (r & 3) + 8 * (r >> 2);
            bv[m][r] = a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f;
        }
    const bool all_co = co_w + 64 <= a.Cout;            // wave-uniform: no channel guard needed
    const int iHW = (int)HW;
    if (a.epi == 0) {
        float *ob = a.out + (long)b * a.out_bs + (long)co_lane * HW;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int oh = h0 + wn * NF + n, ow = w0 + li;
            if (oh >= a.H || ow >= a.W) continue;
            float *op = ob + (long)oh * a.W + ow;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dco = m * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[m][n][r] * a.out_scale + bv[m][r];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    if (all_co || co_lane + dco < a.Cout) op[dco * iHW] = v;
                }
        }
        return;
    }
    // ---- fused GRU gates (gru_gates.hip arithmetic with hardware exp/rcp) ----
    const int Ch = a.epi == 1 ? a.Cout / 2 : a.Cout;
    const bool second = a.epi == 1 && co_w >= Ch;       // wave-uniform: this wave owns r channels
    const int cg = co_lane - (second ? Ch : 0);         // channel inside the Ch-wide gate tensors
    const float *pc = (second ? a.e_c1 : a.e_c0) + (long)b * (second ? a.e_c1_bs : a.e_c0_bs) + (long)cg * HW;
    const float *pz = a.e_c1 + (long)b * a.e_c1_bs + (long)cg * HW;       // epi 2 only
    const float *ph = a.e_h + (long)b * a.e_h_bs + (long)cg * HW;
    float *po = (second ? a.out2 + (long)b * a.out2_bs : a.out + (long)b * a.out_bs) + (long)cg * HW;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int oh = h0 + wn * NF + n, ow = w0 + li;
        if (oh >= a.H || ow >= a.W) continue;
        const long px = (long)oh * a.W + ow;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // all gate operands of these 16 channels are fetched as one batch BEFORE any store
            // (hout may alias h, so the compiler cannot hoist the loads over stores itself)
            float gc[16], gh[16], gz[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = m * 32 + (r & 3) + 8 * (r >> 2);
                const bool ok = all_co || co_lane + dco < a.Cout;
                const long o = (long)(ok ? dco : 0) * iHW + px;
                gc[r] = pc[o];
                gh[r] = (a.epi == 2 || second) ? ph[o] : 0.0f;
                gz[r] = a.epi == 2 ? pz[o] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = m * 32 + (r & 3) + 8 * (r >> 2);
                if (!(all_co || co_lane + dco < a.Cout)) continue;
                const long o = (long)dco * iHW + px;
                const float v = acc[m][n][r] * a.out_scale + bv[m][r];
                if (a.epi == 1) {
                    const float g = conv_sigmoid(__fadd_rn(v, gc[r]));
                    po[o] = second ? __fmul_rn(g, gh[r]) : g;
                } else {
                    const float q = conv_tanh(__fadd_rn(v, gc[r]));
                    po[o] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, gz[r]), gh[r]), __fmul_rn(gz[r], q));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Weight pre-pack: (Cout, Cin, KH, KW) fp32 -> hi/lo fp16, [tap][ci16][coPad][16],
// with the input-channel axis re-chunked to match the per-source 32-channel
// padding of the activation operands.  One thread per packed element.
// ---------------------------------------------------------------------------
struct PackArgs {
    const float *w;
    _Float16 *whi, *wlo;
    int Cout, Cin, taps, CoutPad, nch16;
    int src_ch[CONV_MAX_SRC];
    int nsrc;
    float scale;
};

__global__ __launch_bounds__(256) void conv2d_pack_kernel(PackArgs a) {
    const long total = (long)a.taps * a.nch16 * a.CoutPad * 16;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % 16);
    const int co = (int)((i / 16) % a.CoutPad);
    const int c16 = (int)((i / (16L * a.CoutPad)) % a.nch16);
    const int tap = (int)(i / (16L * a.CoutPad * a.nch16));
    // padded channel index -> (source, channel in source) -> original input channel
    int pc = c16 * 16 + k, s = 0, cbase = 0;
    while (s + 1 < a.nsrc && pc >= ((a.src_ch[s] + 31) & ~31)) {
        pc -= (a.src_ch[s] + 31) & ~31;
        cbase += a.src_ch[s];
        ++s;
    }
    float v = 0.0f;
    if (co < a.Cout && pc < a.src_ch[s]) v = a.w[((long)co * a.Cin + cbase + pc) * a.taps + tap] * a.scale;
    const _Float16 hi = (_Float16)v;
    a.whi[i] = hi;
    a.wlo[i] = (_Float16)(v - (float)hi);
}

// output channels are owned 64 per wave; the packed image is padded to that granule
static int conv_cout_pad(int Cout) { return (Cout + 63) & ~63; }

static int conv_padded_channels(const int *src_ch, int nsrc) {
    int t = 0;
    for (int s = 0; s < nsrc; ++s) t += (src_ch[s] + 31) & ~31;
    return t;
}

extern "C" long dkt_conv2d_packed_elems(const int *src_channels, int nsrc, int Cout, int KH, int KW) {
    if (!src_channels || nsrc < 1 || nsrc > CONV_MAX_SRC || Cout <= 0 || KH <= 0 || KW <= 0) return DKT_E_SHAPE;
    return (long)KH * KW * (conv_padded_channels(src_channels, nsrc) / 16) * conv_cout_pad(Cout) * 16;
}

extern "C" int dkt_conv2d_pack_weights(const float *w, const int *src_channels, int nsrc,
                                       int Cout, int KH, int KW, float scale,
                                       void *w_hi, void *w_lo, int device, void *stream) {
    if (!w || !src_channels || !w_hi || !w_lo) return DKT_E_NULL;
    if (nsrc < 1 || nsrc > CONV_MAX_SRC || Cout <= 0 || KH <= 0 || KW != KH || !(scale > 0.0f)) return DKT_E_SHAPE;
    PackArgs a;
    a.w = w;
    a.whi = (_Float16 *)w_hi;
    a.wlo = (_Float16 *)w_lo;
    a.Cout = Cout;
    a.Cin = 0;
    for (int s = 0; s < CONV_MAX_SRC; ++s) {
        a.src_ch[s] = s < nsrc ? src_channels[s] : 0;
        if (s < nsrc && src_channels[s] <= 0) return DKT_E_SHAPE;
        a.Cin += a.src_ch[s];
    }
    a.nsrc = nsrc;
    a.taps = KH * KW;
    a.CoutPad = conv_cout_pad(Cout);
    a.nch16 = conv_padded_channels(src_channels, nsrc) / 16;
    a.scale = scale;
    DKT_ENTER(device);
    const long total = (long)a.taps * a.nch16 * a.CoutPad * 16;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return dkt_launch_status();
}

template <int KS, int WM, int WN, int NF, int PASSES, int ABL = 0>
static int launch_conv(const ConvArgs &a, int B, hipStream_t st) {
    constexpr int HALO = KS / 2;
    constexpr int NPP = (NF * WN + 2 * HALO) * (32 + 2 * HALO);
    constexpr int STAGE = NPP * 20 * (PASSES == 3 ? 2 : 1);
    const size_t lds = ((size_t)2 * STAGE + 4) * sizeof(unsigned);   // + dummy words for surplus staging items
    auto kern = conv2d_f16s_kernel<KS, WM, WN, NF, PASSES, ABL>;
    if (lds > 64 * 1024) {
        // once per device and instantiation (and never inside a stream capture after warm-up)
        static unsigned long long done_mask = 0;       // benign race: worst case it is set twice
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(done_mask >> (dev & 63) & 1ull)) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            done_mask |= 1ull << (dev & 63);
        }
    }
    const int tiles_h = (a.H + NF * WN - 1) / (NF * WN);
    dim3 grid((unsigned)(a.tiles_w * tiles_h), (unsigned)((a.Cout + 64 * WM - 1) / (64 * WM)), (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return dkt_launch_status();
}

// Tile shape by layer width and image size.  Wide layers put all four waves on the
// channel axis (one block = up to 256 channels: each patch is staged once); narrow
// layers put them on rows.  Images too small to give every CU a block with 4-row
// wave tiles use 2-row wave tiles.
template <int KS, int PASSES>
static int launch_conv_shape(const ConvArgs &a, int B, hipStream_t st) {
    const long tiles4 = (long)a.tiles_w * ((a.H + 3) / 4) * B;    // blocks if a block covers 4 rows
    if (a.Cout <= 64) {
        if (tiles4 / 2 >= 256) return launch_conv<KS, 1, 4, 2, PASSES>(a, B, st);   // 64 co x 8 rows
        return launch_conv<KS, 1, 4, 1, PASSES>(a, B, st);                          // 64 co x 4 rows
    }
    if (a.Cout <= 128) {
        if (tiles4 / 2 >= 256) return launch_conv<KS, 2, 2, 4, PASSES>(a, B, st);   // 128 co x 8 rows
        return launch_conv<KS, 2, 2, 2, PASSES>(a, B, st);                          // 128 co x 4 rows
    }
    // wide layers: 256 co x 2 rows per block, two blocks per CU.  (Measured on 384->256 @184x312:
    // 324 us vs 374 us for the 256 co x 4 rows / one-block-per-CU form.)
    return launch_conv<KS, 4, 1, 2, PASSES>(a, B, st);                              // 256 co x 2 rows
}

struct ConvEpilogue {
    int kind;
    const float *c0, *c1, *h;
    long c0_bs, c1_bs, h_bs;
    float *out2;
    long out2_bs;
};

static int conv2d_f16s_impl(const float *const *src, const int *src_channels, const long *src_bstride,
                            int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                            float out_scale, float *out, long out_bstride,
                            int B, int H, int W, int Cout, int KH, int KW, int relu, int passes,
                            const ConvEpilogue *epi, int device, void *stream) {
    if (!src || !src_channels || !src_bstride || !w_hi || !w_lo || !out) return DKT_E_NULL;
    if (nsrc < 1 || nsrc > CONV_MAX_SRC || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || B > 65535) return DKT_E_SHAPE;
    if (KH != KW || (KH != 1 && KH != 3)) return DKT_E_UNSUPPORTED;
    if (passes < 1 || passes > 3) return DKT_E_UNSUPPORTED;
    ConvArgs a;
    for (int s = 0; s < CONV_MAX_SRC; ++s) {
        a.src[s] = s < nsrc ? src[s] : nullptr;
        a.src_bs[s] = s < nsrc ? src_bstride[s] : 0;
        a.src_ch[s] = s < nsrc ? src_channels[s] : 0;
        if (s < nsrc && (!src[s] || src_channels[s] <= 0)) return DKT_E_NULL;
    }
    a.nsrc = nsrc;
    a.whi = (const _Float16 *)w_hi;
    a.wlo = (const _Float16 *)w_lo;
    a.bias = bias;
    a.out_scale = out_scale;
    a.out = out;
    a.out_bs = out_bstride;
    a.H = H; a.W = W; a.Cout = Cout;
    a.CoutPad = conv_cout_pad(Cout);
    a.nch16 = conv_padded_channels(src_channels, nsrc) / 16;
    a.tiles_w = (W + 31) / 32;
    a.relu = relu ? 1 : 0;
    a.epi = 0;
    a.e_c0 = a.e_c1 = a.e_h = nullptr;
    a.e_c0_bs = a.e_c1_bs = a.e_h_bs = 0;
    a.out2 = nullptr;
    a.out2_bs = 0;
    if (epi) {
        a.epi = epi->kind;
        a.e_c0 = epi->c0; a.e_c1 = epi->c1; a.e_h = epi->h;
        a.e_c0_bs = epi->c0_bs; a.e_c1_bs = epi->c1_bs; a.e_h_bs = epi->h_bs;
        a.out2 = epi->out2; a.out2_bs = epi->out2_bs;
    }
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    if (KH == 3) {
        if (passes == 3) return launch_conv_shape<3, 3>(a, B, st);
        if (passes == 2) return launch_conv_shape<3, 2>(a, B, st);
        return launch_conv_shape<3, 1>(a, B, st);
    }
    if (passes == 3) return launch_conv_shape<1, 3>(a, B, st);
    if (passes == 2) return launch_conv_shape<1, 2>(a, B, st);
    return launch_conv_shape<1, 1>(a, B, st);
}

extern "C" int dkt_conv2d_f16s(const float *const *src, const int *src_channels, const long *src_bstride,
                               int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                               float out_scale, float *out, long out_bstride,
                               int B, int H, int W, int Cout, int KH, int KW, int relu, int passes,
                               int device, void *stream) {
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, out, out_bstride,
                            B, H, W, Cout, KH, KW, relu, passes, nullptr, device, stream);
}

extern "C" int dkt_conv2d_f16s_gate_zr(const float *const *src, const int *src_channels, const long *src_bstride,
                                       int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                                       float out_scale, const float *cz, long cz_bstride,
                                       const float *cr, long cr_bstride, const float *h, long h_bstride,
                                       float *z, long z_bstride, float *rh, long rh_bstride,
                                       int B, int H, int W, int Ch, int KH, int KW, int passes,
                                       int device, void *stream) {
    if (!cz || !cr || !h || !z || !rh) return DKT_E_NULL;
    if (Ch <= 0 || Ch % 64 != 0) return DKT_E_UNSUPPORTED;   // z and r channels must not share a wave
    ConvEpilogue e = {1, cz, cr, h, cz_bstride, cr_bstride, h_bstride, rh, rh_bstride};
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, z, z_bstride,
                            B, H, W, 2 * Ch, KH, KW, 0, passes, &e, device, stream);
}

extern "C" int dkt_conv2d_f16s_gate_out(const float *const *src, const int *src_channels, const long *src_bstride,
                                        int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                                        float out_scale, const float *cq, long cq_bstride,
                                        const float *z, long z_bstride, const float *h, long h_bstride,
                                        float *hout, long hout_bstride,
                                        int B, int H, int W, int Ch, int KH, int KW, int passes,
                                        int device, void *stream) {
    if (!cq || !z || !h || !hout) return DKT_E_NULL;
    if (Ch <= 0) return DKT_E_SHAPE;
    ConvEpilogue e = {2, cq, z, h, cq_bstride, z_bstride, h_bstride, nullptr, 0};
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, hout, hout_bstride,
                            B, H, W, Ch, KH, KW, 0, passes, &e, device, stream);
}
