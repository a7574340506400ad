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
// Block = 4 waves inside a 256-VGPR budget (two blocks per CU).  Wave tile = 32*MF output
// channels (MF = 2; 1 for the 1-2 channel heads) x NF rows x 32 columns; WM x WN waves along (channels, rows).  Blocks are
// persistent (a stream of tiles per block); LDS is double buffered: the fp32 loads of
// chunk c+1 -- of this tile or the block's next one -- are in flight under the MFMAs of
// chunk c; one barrier per chunk.
#include "dkt_common.h"
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CONV_MAX_SRC 4
// scheduling tunables (swept with tools/build_conv_variant.sh + tools/conv_ablation.py)
#ifndef CONV_AR
#define CONV_AR 3
#endif
#ifndef CONV_MIN_BLOCKS
#define CONV_MIN_BLOCKS 2    // blocks per CU the register budget is capped for (256 VGPRs)
#endif
#ifndef CONV_FEW_TILES
#define CONV_FEW_TILES 200   // fewer tiles than this: half-height tile shapes
#endif
#ifndef CONV_AR_NF1
#define CONV_AR_NF1 3
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
    float out_scale;  // 1 / (weight scale * in_scale)
    float in_scale;   // power of two applied to the activations before the fp16 split (range control)
    float *out;
    long out_bs;
    int H, W, Cout, CoutPad, nch16, tiles_w;
    int Ho, Wo;                        // output size (== H, W for stride 1)
    int tiles_xy, n_co, total_tiles;   // persistent tile stream: id = (b*n_co + co block)*tiles_xy + xy
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
    //   3: residual join (core/extractor.py:60 with the preceding norm folded into the weights):
    //      out = relu(e_c0 + [relu](v))
    // Instance-norm of the INPUT folded into the staging (kernel instantiations with NRM = 1; one source):
    // (mean, 1/std) per (batch, channel) plane; the kernel convolves relu((x - mean) * invstd).
    const float *in_norm;
    // Instance-norm statistics of the OUTPUT accumulated in the epilogue (epi 0): every wave writes the sums and sums of
    // squares of its channels over its rows x 32 columns to stats_ws[((b * entries + e) * Cout + co) * 2 + {0, 1}], entry
    // e = (tile index in the image) * WN + wave row; conv_stats_reduce (norm.hip) folds them into the fp64 partial-sum
    // workspace the instance-norm kernels read (the statistics pass over the 235 MB activation disappears).
    float *stats_ws;
    double *stats_part;
    long stats_hw;       // filled by launch_conv: plane size of the output
    // Sub-sampled sources (a 1x1 stride-2 projection run as a 1x1 stride-1 layer on every second pixel of every second row:
    // the tiled stride-2 form stages the odd rows and columns it never uses -- three quarters of its requests): H, W are the
    // OUTPUT size, src_px the pixel stride in the source, src_w / src_hw its row pitch and plane size.  Plain layers: 1, W, H*W.
    int src_px, src_w;
    long src_hw;
};
int conv_stats_reduce(const float *ws, double *part, int B, int C, long entries, long HW, hipStream_t st);      // norm.hip

// Gate non-linearities for the fused epilogues.  The epilogue runs on the VALU after the
// MFMA loop with nothing to overlap it, so it uses the hardware transcendentals
// (v_exp_f32 / v_rcp_f32, ~1 ulp each: absolute error <= 3e-7 on (0,1) / (-1,1)) rather than
// the ~45-instruction correctly-rounded forms the streaming gate kernels afford
// (measured: precise forms cost +7.7 ms per pair, these -x ms).
__device__ __forceinline__ float conv_sigmoid(float x) {
    return __frcp_rn(1.0f + __expf(-x));
}
__device__ __forceinline__ float conv_tanh(float x) {
    const float xc = x < -15.0f ? -15.0f : (x > 15.0f ? 15.0f : x);      // NaN passes through
    const float t = __expf(2.0f * xc);
    return (t - 1.0f) * __frcp_rn(t + 1.0f);
}

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; unsigned u; } v;
    v.h[0] = a;
    v.h[1] = b;
    return v.u;
}

struct ConvArgsPair {
    ConvArgs p[2];
};

template <int KS, int WM, int WN, int NF, int PASSES, int MF = 2, int ST = 1, int CK = 2, int NRM = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 4 ? 1 : CONV_MIN_BLOCKS)) void conv2d_f16s_kernel(ConvArgsPair ap, int nb0) {
    // Two independent convolutions may share one launch (dkt_conv2d_f16s_pair): blocks [0, nb0) stream
    // the tiles of problem 0, the others those of problem 1 -- a small layer (the coarsest GRU: 36
    // tiles) then rides in the tile-quantisation slack of a large one (the finest GRU: 460 tiles on
    // 256 resident blocks) instead of occupying the device for a launch of its own.  The selection
    // is block-uniform: every a.field below is a scalar load from the chosen argument block.
    const bool second = (int)blockIdx.x >= nb0;
    const ConvArgs &a = ap.p[second ? 1 : 0];      // uniform index into the kernel-argument segment: scalar loads
    const int blk_first = second ? nb0 : 0;
    const int blk_count = second ? (int)gridDim.x - nb0 : nb0;
    constexpr int HALO = KS / 2;
    constexpr int TR = NF * WN;              // output rows per block
    // ST = 2: stride-2 convolution (the encoders' down-sampling layers).  Tiles are OUTPUT tiles;
    // the input patch is (TR-1)*ST + KS rows x 31*ST + KS columns, and the fragment of tap
    // (dy,dx) reads patch pixel (ST*row + dy, ST*col + dx): a lane stride of ST*80 B, still
    // conflict-free for ds_read_b128 (8 consecutive lanes hit 8 disjoint 4-bank groups).
    constexpr int PR = (TR - 1) * ST + KS;   // patch rows
    constexpr int PC = 31 * ST + KS;         // patch cols
    constexpr int NPP = PR * PC;             // patch pixels
    // CK = k16 steps per tap and chunk: a chunk is 16*CK input channels.  CK = 1 halves the LDS
    // stage, which lets the 64-channel layers use 8-row tiles (NF = 2) at two blocks per CU.
    constexpr int PITCH = 8 * CK + 4;        // 32-bit words per pixel: 8*CK (16*CK fp16) + 4 pad
    constexpr int PLANE = NPP * PITCH;       // words per (hi or lo) plane
    constexpr int NPLANES = PASSES == 3 ? 2 : 1;   // x_lo is only needed for the w_hi*x_lo pass
    constexpr int STAGE = PLANE * NPLANES;
    constexpr int NSTEP = KS * KS * CK;      // (tap, 16-channel group) steps per chunk
    // weight-fragment ring (prefetch distance AR-1 steps).  A step of an NF=1 tile is only
    // 6 MFMAs (~80 ns): its ring is deeper so that the L2 latency of the weights stays covered.
    constexpr int AR_WANT = NF == 1 ? CONV_AR_NF1 : CONV_AR;
    constexpr int AR = (NSTEP % AR_WANT == 0) ? AR_WANT : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // 2 * STAGE words (+ 8 dummy)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const long HW = (long)a.H * a.W;
    const long SHW = a.src_hw;              // plane size of the sources (== HW unless they are sub-sampled)
    // Persistent block: tiles t = (block index within its problem), + (blocks of that problem), ...; the (tile, chunk)
    // pairs of a block form ONE software-pipelined stream, so that only the block's first
    // chunk is staged with its global-load latency exposed (for a 64->64 layer -- two chunks
    // per tile -- prologue + epilogue used to be 52 of 113 us).
    // tile id = (b * n_co + co block) * tiles_xy + spatial tile.
    auto decode = [&](int t, int &th0, int &tw0, int &tco, int &tb) {
        const int xy = t % a.tiles_xy, r = t / a.tiles_xy;
        tw0 = (xy % a.tiles_w) * 32;
        th0 = (xy / a.tiles_w) * TR;
        tco = (r % a.n_co) * (32 * MF * WM);
        tb = r / a.n_co;
    };
    int h0, w0, co_blk, b;            // tile being computed
    int tile = (int)blockIdx.x - blk_first;
    decode(tile, h0, w0, co_blk, b);

    // ---- staging.  Wave w stages channels 8w..8w+7 of every 32-channel chunk; its lanes walk
    // the patch pixels (pp = lane + 64*it).  The channel is wave-uniform, so a channel plane's
    // base address lives in SGPRs and the per-lane part of the address -- the pixel's byte
    // offset in a plane -- is computed once per TILE (SIT registers), not per chunk: staging
    // costs ~7 VALU instructions per value (select, clamp, 2 cvt + sub + cvt for the hi/lo
    // split, half a pack) instead of ~30.  (PMC on a 64->64 layer before this form: 18 VALU
    // instructions per MFMA, VALU port 50 % busy, MFMA pipe 22 %.)  Each lane writes its
    // pixel's 8 channels as one ds_write_b128 per plane.  Both halves take a SLICE range
    // (slice = pixel slot x channel pair) so that they can be spread over the MFMA steps of the
    // previous chunk.
    constexpr int SG = 2 * CK;                // waves along channels (8 channels each)
    constexpr int PG = WM * WN / SG;          // waves along pixels
    constexpr int SIT = (NPP + 64 * PG - 1) / (64 * PG);   // pixel slots per lane
    constexpr int NSL = SIT * 4;              // slices per chunk
    const int swave = __builtin_amdgcn_readfirstlane(wave) % SG;     // channel group this wave stages
    const int spgrp = __builtin_amdgcn_readfirstlane(wave) / SG;     // pixel group
    unsigned spix[SIT];                       // byte offset of the slot's pixel inside a channel plane
    bool sok[SIT];                            // the slot's pixel lies inside the image
    float sreg[SIT][8];
    unsigned shw[SIT][4], slw[SIT][4];        // converted (hi, lo) fp16 pairs awaiting the LDS write
    const float *sbase = nullptr;             // plane of the first channel this wave stages (wave-uniform)
    int snch = 0;                             // valid channels from there on (<= 0: all padding)
    float nmean[NRM ? 8 : 1], ninv[NRM ? 8 : 1];   // NRM: (mean, 1/std) of the 8 planes this wave stages (wave-uniform)
    auto stage_tile = [&](int th0, int tw0) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int pp = lane + 64 * (it * PG + spgrp);
            const int pr = pp / PC, pc = pp - pr * PC;
            const int ih = th0 * ST - HALO + pr, iw = tw0 * ST - HALO + pc;
            sok[it] = pp < NPP && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            spix[it] = sok[it] ? (unsigned)((ih * a.src_w + iw) * a.src_px) * 4u : 0u;
        }
    };
    auto stage_select = [&](int tb, int chunk) {
        int s = 0, c0 = chunk * 16 * CK;
        while (s + 1 < a.nsrc && c0 >= ((a.src_ch[s] + 31) & ~31)) {
            c0 -= (a.src_ch[s] + 31) & ~31;
            ++s;
        }
        const int cb = c0 + 8 * swave;
        snch = a.src_ch[s] - cb;
        sbase = a.src[s] + (long)tb * a.src_bs[s] + (long)(snch > 0 ? cb : 0) * SHW;
        if (NRM) {
            const float *np = a.in_norm + 2 * ((long)tb * a.src_ch[0] + (snch > 0 ? cb : 0));
#pragma unroll
            for (int j = 0; j < (NRM ? 8 : 1); ++j) {
                const int jc = min(j, max(snch, 1) - 1);
                // wave-uniform values: pinned into SGPRs (the loads themselves are vector loads -- the compiler
                // cannot prove the buffer is not written by this kernel -- and would otherwise hold 16 VGPRs)
                nmean[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(np[2 * jc])));
                ninv[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(np[2 * jc + 1])));
            }
        }
    };
    auto stage_load = [&](int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            if (k >= NSL) break;
            const int it = k >> 2, q = k & 3;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                // unconditional load from a clamped (valid) plane; padding channels and pixels
                // outside the image are zeroed when the value is converted (stage_store) -- a
                // select here would tie an s_waitcnt vmcnt(0) to every load
                const int j = 2 * q + jj;
                const int jc = min(j, max(snch, 1) - 1);                     // wave-uniform
                const char *pj = (const char *)(sbase + (long)jc * SHW);
                sreg[it][j] = *(const float *)(pj + spix[it]);
            }
        }
    };
    auto stage_store = [&](unsigned *buf, int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            if (k >= NSL) break;
            const int it = k >> 2, q = k & 3;
            float u0 = sreg[it][2 * q], u1 = sreg[it][2 * q + 1];
            if (NRM) {     // the arithmetic of instnorm_apply_kernel (norm.hip), so that the fusion is bit-identical
                u0 = dkt_relu(__fmul_rn(__fsub_rn(u0, nmean[(2 * q) % (NRM ? 8 : 1)]), ninv[(2 * q) % (NRM ? 8 : 1)]));
                u1 = dkt_relu(__fmul_rn(__fsub_rn(u1, nmean[(2 * q + 1) % (NRM ? 8 : 1)]), ninv[(2 * q + 1) % (NRM ? 8 : 1)]));
            }
            const float v0 = (2 * q < snch && sok[it]) ? u0 : 0.0f;
            const float v1 = (2 * q + 1 < snch && sok[it]) ? u1 : 0.0f;
            // range: |x * in_scale| must stay below 65520 (the fp16 hi part); beyond that -- and for
            // NaN / Inf inputs -- the result is non-finite, never a silently saturated value
            const float x0 = v0 * a.in_scale;
            const float x1 = v1 * a.in_scale;
            const _Float16 h0_ = (_Float16)x0, h1_ = (_Float16)x1;
            shw[it][q] = pack_h2(h0_, h1_);
            if (NPLANES == 2) slw[it][q] = pack_h2((_Float16)(x0 - (float)h0_), (_Float16)(x1 - (float)h1_));
            if (q == 3) {
                const int pp = lane + 64 * (it * PG + spgrp);
                unsigned *dst = pp < NPP ? buf + pp * PITCH + 4 * swave : lds + 2 * STAGE;   // surplus lanes -> dummy
                *(uint4 *)dst = make_uint4(shw[it][0], shw[it][1], shw[it][2], shw[it][3]);
                if (NPLANES == 2)
                    *(uint4 *)(dst + (pp < NPP ? PLANE : 4)) = make_uint4(slw[it][0], slw[it][1], slw[it][2], slw[it][3]);
            }
        }
    };

    f32x16 acc[MF][NF];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    };
    zero_acc();

    const int li = lane & 31, kg = lane >> 5;
    // weight fragment base (halves): ((tap*nch16 + ch16)*CoutPad + co)*16 + kg*8; waves past
    // the (64-padded) channel count idle on clamped weights
    auto wlane_of = [&](int tco) {
        const int cw = tco + wm * 32 * MF;
        return ((long)((cw < a.CoutPad ? cw : 0) + li)) * 16 + kg * 8;
    };
    long wl_cur = wlane_of(co_blk), wl_nxt = wl_cur;   // weight bases of the computed / the following tile
    const long wstep = (long)a.CoutPad * 16;           // one 16-channel slab
    const int nchunks = a.nch16 / CK;

    f16x8 Ahi[AR][MF], Alo[AR][MF];
    f16x8 Bhi[2][NF], Blo[2][NF];
    auto loadA = [&](int slot, long wlane, int chunk, int step) {
        const int tap = step / CK, kh = step % CK;
        const long wbase = ((long)tap * a.nch16 + (chunk * CK + kh)) * wstep + wlane;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            Ahi[slot][m] = *(const f16x8 *)(a.whi + wbase + m * 32 * 16);
            if (PASSES >= 2) Alo[slot][m] = *(const f16x8 *)(a.wlo + wbase + m * 32 * 16);
        }
    };
    auto loadB = [&](int slot, const unsigned *buf, int step) {
        const int tap = step / CK, kh = step % CK;
        const int dy = tap / KS, dx = tap % KS;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int pp = ((wn * NF + n) * ST + dy) * PC + li * ST + dx;
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
            for (int m = 0; m < MF; ++m)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[as][m], Bhi[bs][n], acc[m][n], 0, 0, 0);
        if (PASSES >= 2) {
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int m = 0; m < MF; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[as][m], Bhi[bs][n], acc[m][n], 0, 0, 0);
        }
        if (PASSES == 3) {
#pragma unroll
            for (int n = 0; n < NF; ++n)
#pragma unroll
                for (int m = 0; m < MF; ++m)
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
    auto epilogue = [&]() {
        // ---- epilogue: un-scale, bias, optional ReLU, NCHW stores (128-byte rows) ----
        const int co_w = co_blk + wm * 32 * MF;            // this wave's first output channel
        if (co_w >= a.CoutPad) return;                     // idle wave
        // this lane's 32 output channels: co = co_lane + m*32 + (r&3) + 8*(r>>2); their biases
        // are fetched as one batch of independent loads (clamped index, no per-element branch)
        const int co_lane = co_w + 4 * kg;
        float bv[MF][16];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_lane + m * 32 + (r & 3) + 8 * (r >> 2);
                bv[m][r] = a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f;
            }
        const bool all_co = co_w + 32 * MF <= a.Cout;            // wave-uniform: no channel guard needed
        const int iHW = (int)HW;
        if (a.epi == 0) {
            const int oHW = a.Ho * a.Wo;                    // output plane (== HW when ST == 1)
            float *ob = a.out + (long)b * a.out_bs + (long)co_lane * oHW;
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int oh = h0 + wn * NF + n, ow = w0 + li;
                if (oh >= a.Ho || ow >= a.Wo) continue;
                float *op = ob + (long)oh * a.Wo + ow;
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dco = m * 32 + (r & 3) + 8 * (r >> 2);
                        float v = acc[m][n][r] * a.out_scale + bv[m][r];
                        if (a.relu) v = dkt_relu(v);
                        if (all_co || co_lane + dco < a.Cout) op[dco * oHW] = v;
                    }
            }
            if (a.stats_ws) {
                // sums / sums of squares of the values just stored: per lane over its NF pixels, then a reduce-scatter over the
                // 32 pixel lanes of the half-wave (31 exchanges per statistic instead of 160): lane li ends with value index li.
                // Channel blocks go in groups of two (32 values per lane) and, for an odd MF, a last single one (16 values: the
                // first exchange only adds, lane li & 15 ends with value li & 15).
                auto stats_group = [&](const int m0, const int nm) {
                    float s1[32], s2[32];
#pragma unroll
                    for (int i = 0; i < 32; ++i) s1[i] = s2[i] = 0.0f;
#pragma unroll
                    for (int n = 0; n < NF; ++n) {
                        const int oh = h0 + wn * NF + n, ow = w0 + li;
                        const bool ok = oh < a.Ho && ow < a.Wo;
#pragma unroll
                        for (int mm = 0; mm < nm; ++mm)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float v = acc[m0 + mm][n][r] * a.out_scale + bv[m0 + mm][r];
                                if (a.relu) v = dkt_relu(v);
                                v = ok ? v : 0.0f;
                                s1[mm * 16 + r] = __fadd_rn(s1[mm * 16 + r], v);
                                s2[mm * 16 + r] = __fadd_rn(s2[mm * 16 + r], __fmul_rn(v, v));
                            }
                    }
                    if (nm == 1) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            s1[i] = __fadd_rn(s1[i], __shfl_xor(s1[i], 16, 32));
                            s2[i] = __fadd_rn(s2[i], __shfl_xor(s2[i], 16, 32));
                        }
                    }
#pragma unroll
                    for (int d = 16; d >= 1; d >>= 1) {
                        if (nm == 1 && d == 16) continue;
                        const bool up = (li & d) != 0;
#pragma unroll
                        for (int j = 0; j < d; ++j) {
                            const float k1 = up ? s1[j + d] : s1[j], g1 = up ? s1[j] : s1[j + d];
                            const float k2 = up ? s2[j + d] : s2[j], g2 = up ? s2[j] : s2[j + d];
                            s1[j] = __fadd_rn(k1, __shfl_xor(g1, d, 32));
                            s2[j] = __fadd_rn(k2, __shfl_xor(g2, d, 32));
                        }
                    }
                    const int vi = nm == 1 ? (li & 15) : li;                 // value index this lane holds
                    const int dco = (m0 + (vi >> 4)) * 32 + (vi & 3) + 8 * ((vi & 15) >> 2);
                    const int co = co_lane + dco;
                    const long e = (long)((h0 / (NF * WN)) * a.tiles_w + w0 / 32) * WN + wn;
                    if (co < a.Cout && (nm == 2 || li < 16)) {
                        float *p = a.stats_ws + (((long)b * a.tiles_xy * WN + e) * a.Cout + co) * 2;
                        p[0] = s1[0];
                        p[1] = s2[0];
                    }
                };
#pragma unroll
                for (int m0 = 0; m0 + 1 < MF; m0 += 2) stats_group(m0, 2);
                if (MF & 1) stats_group(MF - 1, 1);
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
            for (int mh = 0; mh < 2 * MF; ++mh) {
                // the gate operands of 8 channels are fetched as one batch BEFORE any store (hout
                // may alias h, so the compiler cannot hoist the loads over stores itself); batches
                // of 8 rather than 16 keep the kernel inside the 256-VGPR budget of two blocks/CU
                const int m = mh >> 1, r0 = (mh & 1) * 8;
                float gc[8], gh[8], gz[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + i;
                    const int dco = m * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = all_co || co_lane + dco < a.Cout;
                    const long o = (long)(ok ? dco : 0) * iHW + px;
                    gc[i] = pc[o];
                    gh[i] = (a.epi == 2 || second) ? ph[o] : 0.0f;
                    gz[i] = a.epi == 2 ? pz[o] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + i;
                    const int dco = m * 32 + (r & 3) + 8 * (r >> 2);
                    if (!(all_co || co_lane + dco < a.Cout)) continue;
                    const long o = (long)dco * iHW + px;
                    const float v = acc[m][n][r] * a.out_scale + bv[m][r];
                    if (a.epi == 1) {
                        const float g = conv_sigmoid(__fadd_rn(v, gc[i]));
                        po[o] = second ? __fmul_rn(g, gh[i]) : g;
                    } else if (a.epi == 3) {
                        po[o] = dkt_relu(__fadd_rn(gc[i], a.relu ? dkt_relu(v) : v));
                    } else {
                        const float q = conv_tanh(__fadd_rn(v, gc[i]));
                        po[o] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, gz[i]), gh[i]), __fmul_rn(gz[i], q));
                    }
                }
            }
        }
    };

    constexpr int HALF = NSTEP / 2;
    constexpr int PER = (NSL + HALF - 1) / HALF;
    constexpr int NMMA = MF * NF * PASSES;
    stage_tile(h0, w0);
    stage_select(b, 0);
    stage_load(0, NSL);
    stage_store(lds, 0, NSL);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < AR - 1; ++s) loadA(s, wl_cur, 0, s);
    loadB(0, lds, 0);
    int g = 0;                               // chunks consumed by this block: LDS buffer parity
    for (;;) {
        const int tn = tile + blk_count;
        const bool have_next = tn < a.total_tiles;
        int nh0 = h0, nw0 = w0, nco = co_blk, nb = b;
        if (have_next) decode(tn, nh0, nw0, nco, nb);
        wl_nxt = wlane_of(nco);
        for (int c = 0; c < nchunks; ++c, ++g) {
            const bool in_tile = c + 1 < nchunks;
            const bool more = in_tile || have_next;
            const unsigned *cur = lds + (g & 1) * STAGE;
            unsigned *nxt = lds + ((g + 1) & 1) * STAGE;
            // what this chunk's steps stage and prefetch: the tile's next chunk, else the next
            // tile's first chunk, else (the block's very last chunk) itself again, harmlessly,
            // into the idle buffer
            const int c_f = in_tile ? c + 1 : (have_next ? 0 : c);
            const long wl_f = in_tile ? wl_cur : wl_nxt;
            if (in_tile || !have_next) stage_select(b, c_f);
            else {
                stage_select(nb, 0);
                stage_tile(nh0, nw0);
            }
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                // weights for step s+AR-1 (possibly the following chunk's first steps)
                const int sa = s + AR - 1;
                if (sa < NSTEP) loadA(sa % AR, wl_cur, c, sa);
                else loadA(sa % AR, wl_f, c_f, sa - NSTEP);
                if (s + 1 < NSTEP) loadB((s + 1) & 1, cur, s + 1);
                if (s < HALF) stage_load(s * PER, (s + 1) * PER);
                else stage_store(nxt, (s - HALF) * PER, (s - HALF + 1) * PER);
                mma(s % AR, s & 1);
#pragma unroll
                for (int i = 0; i < NMMA / CONV_SGB_MFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, CONV_SGB_MFMA, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x320, CONV_SGB_MEM, 0);    // VMEM read / DS read / DS write
                    __builtin_amdgcn_sched_group_barrier(0x002, CONV_SGB_VALU, 0);   // VALU
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (more) loadB(0, nxt, 0);      // step 0 always uses B slot 0
        }
        epilogue();
        if (!have_next) break;
        tile = tn; h0 = nh0; w0 = nw0; co_blk = nco; b = nb;
        wl_cur = wl_nxt;
        zero_acc();
    }
}

// ---------------------------------------------------------------------------
// Weight pre-pack: (Cout, Cin, KH, KW) fp32 -> hi/lo fp16, [tap][ci16][coPad][16],
// with the input-channel axis re-chunked to match the per-source 32-channel
// padding of the activation operands.  One thread per packed element.
// ---------------------------------------------------------------------------
#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 0
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
#endif  // CONV_TU_PASSES == 0

static int conv_cout_pad(int Cout) { return (Cout + 63) & ~63; }

static int conv_padded_channels(const int *src_ch, int nsrc) {
    int t = 0;
    for (int s = 0; s < nsrc; ++s) t += (src_ch[s] + 31) & ~31;
    return t;
}

#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 0
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

#endif  // CONV_TU_PASSES == 0 (weight packer)

// Resident blocks the device holds of one instantiation (occupancy x CUs), cached per device.
static int conv_slots(const void *kern, size_t lds, int dev, int threads) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    return per_cu * cus;
}

// A second, independent convolution that shares the launch (see the kernel's head comment).
struct ConvSecond {
    ConvArgs a;
    int B;
};

template <int KS, int WM, int WN, int NF, int PASSES, int MF = 2, int ST = 1, int CK = 2, int NRM = 0>
static int launch_conv(ConvArgs a, int B, hipStream_t st, const ConvSecond *sec = nullptr) {
    constexpr int NPP = ((NF * WN - 1) * ST + KS) * (31 * ST + KS);
    constexpr int STAGE = NPP * (8 * CK + 4) * (PASSES == 3 ? 2 : 1);
    const size_t lds = ((size_t)2 * STAGE + 8) * sizeof(unsigned);   // + dummy words for surplus staging lanes
    auto kern = conv2d_f16s_kernel<KS, WM, WN, NF, PASSES, MF, ST, CK, NRM>;
    // once per device and instantiation (and never inside a stream capture after warm-up)
    static int slots[64] = {0};                        // benign race: worst case computed twice
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!slots[dev & 63]) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        slots[dev & 63] = conv_slots((const void *)kern, lds, dev, 64 * WM * WN);
    }
    const int tiles_h = (a.Ho + NF * WN - 1) / (NF * WN);
    a.tiles_xy = a.tiles_w * tiles_h;
    a.n_co = (a.Cout + 32 * MF * WM - 1) / (32 * MF * WM);
    const long total = (long)a.tiles_xy * a.n_co * B;
    if (total > 0x7fffffffL) return DKT_E_SHAPE;
    a.total_tiles = (int)total;
    constexpr bool persist = true;        // blocks stream their tiles (cross-tile pipelining)
    const long cap = slots[dev & 63];
    if (!sec) {
        const long nblk = persist && total > cap ? cap : total;
        ConvArgsPair ap;
        ap.p[0] = a;
        ap.p[1] = a;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), lds, st, ap, (int)nblk);
        int rc = dkt_launch_status();
        if (rc == DKT_OK && a.stats_ws)
            rc = conv_stats_reduce(a.stats_ws, a.stats_part, B, a.Cout, (long)a.tiles_xy * WN, (long)a.Ho * a.Wo, st);
        return rc;
    }
    if (a.stats_ws || sec->a.stats_ws) return DKT_E_UNSUPPORTED;
    // ---- two problems: resident blocks are split in proportion to the work (tiles x chunks)
    ConvArgs b = sec->a;
    b.tiles_xy = b.tiles_w * ((b.Ho + NF * WN - 1) / (NF * WN));
    b.n_co = (b.Cout + 32 * MF * WM - 1) / (32 * MF * WM);
    const long total1 = (long)b.tiles_xy * b.n_co * sec->B;
    if (total1 > 0x7fffffffL) return DKT_E_SHAPE;
    b.total_tiles = (int)total1;
    long nb0 = total, nb1 = total1;
    if (persist && total + total1 > cap) {
        const double w0 = (double)total * a.nch16, w1 = (double)total1 * b.nch16;
        nb1 = (long)(cap * w1 / (w0 + w1) + 0.5);
        nb1 = nb1 < 1 ? 1 : (nb1 > total1 ? total1 : nb1);
        nb0 = cap - nb1;
        if (nb0 > total) nb0 = total;
        if (nb0 < 1) nb0 = 1;
    }
    ConvArgsPair ap;
    ap.p[0] = a;
    ap.p[1] = b;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nb0 + nb1)), dim3(64 * WM * WN), lds, st, ap, (int)nb0);
    return dkt_launch_status();
}

#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 3
#include "conv_ws.h"      // weights-stationary kernel of the 64 -> 64 layers (three-pass form only)
#define CONV_HAVE_WS 1
#endif

// Tile shape by layer width and image size.  Wide layers put all four waves on the
// channel axis (one block = up to 256 channels: each patch is staged once); narrow
// layers put them on rows.  Images too small to give every CU a block with 4-row
// wave tiles use 2-row wave tiles.
template <int KS, int PASSES>
static int launch_conv_shape(const ConvArgs &a, int B, hipStream_t st, const ConvSecond *sec = nullptr) {
    const long tiles4 = (long)a.tiles_w * ((a.H + 3) / 4) * B;    // blocks if a block covers 4 rows
#ifdef CONV_HAVE_WS
    if constexpr (KS == 3 && PASSES == 3) {
        if (!sec && conv_ws_eligible(a, B)) return launch_conv_ws(a, B, st);
    }
#endif
    if (a.in_norm) {
        // instance norm + ReLU of the input folded into the staging: the second 3x3 layer of the feature
        // encoder's residual blocks (core/extractor.py:46-50; 64, 96 and 128 channels) -- the same tile
        // shapes as below, instantiated with NRM = 1
        if constexpr (KS == 3) {
            if (sec || a.nsrc != 1 || a.epi == 1 || a.epi == 2) return DKT_E_UNSUPPORTED;
            if (a.Cout > 32 && a.Cout <= 64) {
                if (tiles4 / 2 >= 512) return launch_conv<KS, 1, 4, 2, PASSES, 2, 1, 1, 1>(a, B, st);
                return launch_conv<KS, 1, 4, 1, PASSES, 2, 1, 2, 1>(a, B, st);
            }
            if (a.Cout > 64 && a.Cout <= 96 && tiles4 >= CONV_FEW_TILES) return launch_conv<KS, 1, 4, 1, PASSES, 3, 1, 2, 1>(a, B, st);
            if (a.Cout > 64 && a.Cout <= 128) {
                if (tiles4 < CONV_FEW_TILES) return launch_conv<KS, 2, 2, 1, PASSES, 2, 1, 2, 1>(a, B, st);
                return launch_conv<KS, 2, 2, 2, PASSES, 2, 1, 2, 1>(a, B, st);
            }
        }
        return DKT_E_UNSUPPORTED;
    }
    // 4-row blocks keep the LDS stage at 65 KB, i.e. two blocks per CU; the 8-row forms
    // (one block per CU) measured 15-35 % slower on every encoder layer (tools/_exp_enc.py).
    // Small images (the 1/8 and 1/16 GRUs: 115 and 69 tiles of the default shape for 256 CUs)
    // take half-height tiles so that twice as many CUs work.
    const long few = CONV_FEW_TILES;
    // the flow / disparity heads (2 and 1 output channels): a 32-channel wave tile halves the padded MFMA work
    if (a.Cout <= 32) return launch_conv<KS, 1, 4, 1, PASSES, 1>(a, B, st, sec);  // 32 co x 4 rows
    if (a.Cout <= 64) {
        // 64 co x 8 rows with 16-channel chunks (33 KB LDS stage: still two blocks per CU): twice the
        // MFMAs per weight fragment of the 4-row form (64->64 @368x624: 90 -> 78 us; not for images
        // that give fewer 8-row tiles than resident blocks: @184x312 29 -> 32 us)
        if constexpr (KS == 3) {
            if (tiles4 / 2 >= 512) return launch_conv<KS, 1, 4, 2, PASSES, 2, 1, 1>(a, B, st, sec);   // >= one full wave of blocks
        }
        return launch_conv<KS, 1, 4, 1, PASSES>(a, B, st, sec);                          // 64 co x 4 rows
    }
    if (a.Cout <= 128) {
        // 65 .. 96 channels: a 96-channel wave tile (three 32-channel blocks per wave, four waves on rows) instead of padding
        // a quarter of the 128-channel tile's MFMAs and weight fragments: 96->96 @368x624 205 -> 181 us, two images 357 -> 330
        if constexpr (KS == 3) {
            if (a.Cout <= 96 && !sec && tiles4 >= few) return launch_conv<KS, 1, 4, 1, PASSES, 3>(a, B, st);
        }
        if (tiles4 < few) return launch_conv<KS, 2, 2, 1, PASSES>(a, B, st, sec);        // 128 co x 2 rows
        return launch_conv<KS, 2, 2, 2, PASSES>(a, B, st, sec);                          // 128 co x 4 rows
    }
    // 257 .. 384 channels (the context encoder's 128 -> 3 x 128 projection, raft_stereo.py:103-106) on large images: four waves of
    // 96 channels each and two rows -- the patch is staged once and no channel block is padding, where the 256-channel tile runs
    // a second, half-empty block: 253 -> 213 us @184x312 (slower below ~500 tiles: 65 -> 70 us @92x156)
    if constexpr (KS == 3) {
        if (a.Cout > 256 && a.Cout <= 384 && !sec && (long)a.tiles_w * ((a.H + 1) / 2) * B >= 512)
            return launch_conv<KS, 4, 1, 2, PASSES, 3>(a, B, st);
    }
    const long tiles2 = (long)a.tiles_w * ((a.H + 1) / 2) * B * ((a.Cout + 255) / 256);
    if (tiles2 < few) return launch_conv<KS, 4, 1, 1, PASSES>(a, B, st, sec);            // 256 co x 1 row
    // wide layers on large images: ONE block of 8 waves per CU, 256 co x 4 rows (WM x WN = 4 x 2): the two
    // waves of a SIMD share their weight fragments through L1 and the patch halo shrinks (6 rows
    // staged per 4 instead of 4 per 2).  Measured 384->256 @184x312: 326 -> 315 us; 128->256: 113 -> 109.
    if (tiles4 * ((a.Cout + 255) / 256) >= 256) {
        if constexpr (KS == 3) return launch_conv<KS, 4, 2, 2, PASSES>(a, B, st, sec);
    }
    // otherwise: 256 co x 2 rows per block, two blocks per CU.  (Measured on 384->256 @184x312:
    // 324 us vs 374 us for the 256 co x 4 rows / one-block-per-CU form.)
    return launch_conv<KS, 4, 1, 2, PASSES>(a, B, st, sec);                              // 256 co x 2 rows
}

// Stride-2 layers (the encoders' down-sampling convolutions, core/extractor.py:16,34,136-138):
// 2-row output tiles keep the (2*TR+1) x 65 input patch at 52 KB per LDS stage.
template <int KS, int PASSES>
static int launch_conv_stride2(const ConvArgs &a, int B, hipStream_t st) {
    if constexpr (KS == 3 && PASSES == 3) {
        // 128 co x 4 rows with 16-channel chunks (9 x 65 patch, 56 KB per stage, one block per CU): twice the MFMAs per weight
        // fragment of the 2-row form -- 64->96 @736x1248 281 -> 243 us, two images 518 -> 422; 96->128 @368x624 x 2: 174 -> 138
        if (a.Cout <= 128 && (long)a.tiles_w * ((a.Ho + 3) / 4) * B >= 512) {
            // (up to 96 channels: the 96-channel wave tile, as for stride 1 -- 243 -> 223 us, two images 422 -> 395)
            if (a.Cout <= 96) return launch_conv<KS, 1, 4, 1, PASSES, 3, 2, 1>(a, B, st);
            return launch_conv<KS, 2, 2, 2, PASSES, 2, 2, 1>(a, B, st);
        }
    }
    if (a.Cout <= 128) return launch_conv<KS, 2, 2, 1, PASSES, 2, 2>(a, B, st);   // 128 co x 2 rows
    return launch_conv<KS, 4, 1, 1, PASSES, 2, 2>(a, B, st);                      // 256 co x 1 row
}

// The kernel instantiations are split by number of MFMA passes so that the build can compile them as
// three parallel translation units (dkt_stereo_amd/build.py: -DCONV_TU_PASSES=1|2|3; =0 holds the ABI
// and the weight packer).  Compiled without the macro this file is one complete translation unit.
int conv2d_launch_p1(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec);
int conv2d_launch_p2(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec);
int conv2d_launch_p3(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec);

template <int PASSES>
static int conv2d_launch_passes(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec) {
    if (stride == 2) {
        if (sec || a.in_norm) return DKT_E_UNSUPPORTED;
        if (KH == 1) {
            // 1x1, stride 2 = the 1x1 stride-1 layer on the even pixels of the even rows (see ConvArgs::src_px)
            ConvArgs v = a;
            v.H = a.Ho; v.W = a.Wo;
            v.src_px = 2; v.src_w = a.W; v.src_hw = (long)a.H * a.W;
            return launch_conv_shape<1, PASSES>(v, B, st, nullptr);
        }

        if (KH == 3) return launch_conv_stride2<3, PASSES>(a, B, st);
        return launch_conv_stride2<1, PASSES>(a, B, st);
    }
    if (KH == 3) return launch_conv_shape<3, PASSES>(a, B, st, sec);
    return launch_conv_shape<1, PASSES>(a, B, st, sec);
}
#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 1
int conv2d_launch_p1(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec) { return conv2d_launch_passes<1>(a, B, KH, stride, st, sec); }
#endif
#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 2
int conv2d_launch_p2(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec) { return conv2d_launch_passes<2>(a, B, KH, stride, st, sec); }
#endif
#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 3
int conv2d_launch_p3(const ConvArgs &a, int B, int KH, int stride, hipStream_t st, const ConvSecond *sec) { return conv2d_launch_passes<3>(a, B, KH, stride, st, sec); }
#endif

#if !defined(CONV_TU_PASSES) || CONV_TU_PASSES == 0
struct ConvEpilogue {
    int kind;
    const float *c0, *c1, *h;
    long c0_bs, c1_bs, h_bs;
    float *out2;
    long out2_bs;
};

// validates one convolution's parameters and fills its kernel argument block
static int conv_fill(ConvArgs &a, const float *const *src, const int *src_channels, const long *src_bstride,
                     int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                     float out_scale, float in_scale, float *out, long out_bstride,
                     int B, int H, int W, int Cout, int KH, int KW, int relu, int passes,
                     const ConvEpilogue *epi, int stride) {
    if (!src || !src_channels || !src_bstride || !w_hi || !w_lo || !out) return DKT_E_NULL;
    if (stride != 1 && (stride != 2 || epi)) return DKT_E_UNSUPPORTED;
    if (nsrc < 1 || nsrc > CONV_MAX_SRC || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || B > 65535) return DKT_E_SHAPE;
    if (KH != KW || (KH != 1 && KH != 3)) return DKT_E_UNSUPPORTED;
    if (passes < 1 || passes > 3) return DKT_E_UNSUPPORTED;
    if (!(in_scale > 0.0f) || !(out_scale > 0.0f)) return DKT_E_SHAPE;
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
    a.in_scale = in_scale;
    a.out = out;
    a.out_bs = out_bstride;
    a.H = H; a.W = W; a.Cout = Cout;
    // "same"-style padding KH/2 (what every layer on the path uses): output = floor((H - 1) / stride) + 1
    a.Ho = (H - 1) / stride + 1;
    a.Wo = (W - 1) / stride + 1;
    a.CoutPad = conv_cout_pad(Cout);
    a.nch16 = conv_padded_channels(src_channels, nsrc) / 16;
    a.tiles_w = (a.Wo + 31) / 32;
    a.tiles_xy = a.n_co = a.total_tiles = 0;       // set by launch_conv (they depend on the tile shape)
    a.relu = relu ? 1 : 0;
    a.epi = 0;
    a.e_c0 = a.e_c1 = a.e_h = nullptr;
    a.e_c0_bs = a.e_c1_bs = a.e_h_bs = 0;
    a.out2 = nullptr;
    a.out2_bs = 0;
    a.in_norm = nullptr;
    a.stats_ws = nullptr;
    a.stats_part = nullptr;
    a.stats_hw = 0;
    a.src_px = 1; a.src_w = W; a.src_hw = (long)H * W;
    if (epi) {
        a.epi = epi->kind;
        a.e_c0 = epi->c0; a.e_c1 = epi->c1; a.e_h = epi->h;
        a.e_c0_bs = epi->c0_bs; a.e_c1_bs = epi->c1_bs; a.e_h_bs = epi->h_bs;
        a.out2 = epi->out2; a.out2_bs = epi->out2_bs;
    }
    return DKT_OK;
}

static int conv_dispatch(const ConvArgs &a, int B, int KH, int stride, int passes, hipStream_t st, const ConvSecond *sec) {
    if (passes == 3) return conv2d_launch_p3(a, B, KH, stride, st, sec);
    if (passes == 2) return conv2d_launch_p2(a, B, KH, stride, st, sec);
    return conv2d_launch_p1(a, B, KH, stride, st, sec);
}

static int conv2d_f16s_impl(const float *const *src, const int *src_channels, const long *src_bstride,
                            int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                            float out_scale, float in_scale, float *out, long out_bstride,
                            int B, int H, int W, int Cout, int KH, int KW, int relu, int passes,
                            const ConvEpilogue *epi, int device, void *stream, int stride = 1) {
    ConvArgs a;
    const int rc = conv_fill(a, src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, in_scale, out,
                             out_bstride, B, H, W, Cout, KH, KW, relu, passes, epi, stride);
    if (rc != DKT_OK) return rc;
    DKT_ENTER(device);
    return conv_dispatch(a, B, KH, stride, passes, (hipStream_t)stream, nullptr);
}

// ---- two convolutions in one launch (descriptor form) ----
static int conv_fill_desc(ConvArgs &a, const dkt_conv_desc *d, int passes) {
    if (!d) return DKT_E_NULL;
    ConvEpilogue e = {d->epilogue, d->e0, d->e1, d->h, d->e0_bstride, d->e1_bstride, d->h_bstride, d->out2, d->out2_bstride};
    if (d->epilogue < 0 || d->epilogue > 3) return DKT_E_UNSUPPORTED;
    if (d->epilogue == 1) {
        if (!d->e0 || !d->e1 || !d->h || !d->out2) return DKT_E_NULL;
        if (d->Cout % 128 != 0) return DKT_E_UNSUPPORTED;          // Ch multiple of 64: z and r never share a wave
    } else if (d->epilogue == 2) {
        if (!d->e0 || !d->e1 || !d->h) return DKT_E_NULL;
    } else if (d->epilogue == 3) {
        if (!d->e0) return DKT_E_NULL;
    }
    const int stride = d->stride == 2 ? 2 : 1;
    if (d->stride != 0 && d->stride != 1 && d->stride != 2) return DKT_E_UNSUPPORTED;
    const int rc = conv_fill(a, d->src, d->src_channels, d->src_bstride, d->nsrc, d->w_hi, d->w_lo, d->bias, d->out_scale,
                             d->in_scale, d->out, d->out_bstride, d->B, d->H, d->W, d->Cout, d->KH, d->KW,
                             d->relu, passes,
                             d->epilogue ? &e : nullptr, stride);
    if (rc != DKT_OK) return rc;
    if (d->stats_ws || d->stats_part) {
        if (!d->stats_ws || !d->stats_part) return DKT_E_NULL;
        if (d->epilogue != 0) return DKT_E_UNSUPPORTED;
        a.stats_ws = d->stats_ws;
        a.stats_part = (double *)d->stats_part;
    }
    if (d->in_norm) {
        if (d->nsrc != 1 || d->KH != 3 || d->Cout <= 32 || d->Cout > 128 || d->epilogue == 1 || d->epilogue == 2)
            return DKT_E_UNSUPPORTED;
        a.in_norm = d->in_norm;
    }
    return DKT_OK;
}

static int conv_width_class(int Cout) { return Cout <= 32 ? 0 : Cout <= 64 ? 1 : Cout <= 128 ? 2 : 3; }

extern "C" long dkt_conv2d_stats_ws_floats(int B, int Cout, int Ho, int Wo) {
    if (B <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0) return DKT_E_SHAPE;
    // One entry per (tile, wave row): tiles_w * ceil(Ho / TR) * WN per image with TR = WN * NF rows per tile, TR in {1, 2, 4, 8}
    // over the tile shapes -- waves whose rows lie past Ho write (zero) entries too, so the bound is Ho rounded UP to 8 rows per
    // 32-column strip (`Ho` entries were 9 short at Ho = 33 on the 4-row shapes)
    return (long)B * ((Wo + 31) / 32) * ((Ho + 7) / 8 * 8) * Cout * 2;
}

extern "C" int dkt_conv2d_f16s_desc(const dkt_conv_desc *p, int passes, int device, void *stream) {
    ConvArgs a;
    const int rc = conv_fill_desc(a, p, passes);
    if (rc != DKT_OK) return rc;
    DKT_ENTER(device);
    return conv_dispatch(a, p->B, p->KH, p->stride == 2 ? 2 : 1, passes, (hipStream_t)stream, nullptr);
}

extern "C" int dkt_conv2d_f16s_pair(const dkt_conv_desc *p0, const dkt_conv_desc *p1, int passes, int device, void *stream) {
    ConvArgs a;
    ConvSecond sec;
    int rc = conv_fill_desc(a, p0, passes);
    if (rc != DKT_OK) return rc;
    rc = conv_fill_desc(sec.a, p1, passes);
    if (rc != DKT_OK) return rc;
    if (p0->in_norm || p1->in_norm || p0->stride == 2 || p1->stride == 2) return DKT_E_UNSUPPORTED;
    sec.B = p1->B;
    // both problems run one kernel instantiation: same filter size and the same output-width class
    if (p0->KH != p1->KH || conv_width_class(p0->Cout) != conv_width_class(p1->Cout)) return DKT_E_UNSUPPORTED;
    DKT_ENTER(device);
    return conv_dispatch(a, p0->B, p0->KH, 1, passes, (hipStream_t)stream, &sec);
}

extern "C" int dkt_conv2d_f16s(const float *const *src, const int *src_channels, const long *src_bstride,
                               int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                               float out_scale, float in_scale, float *out, long out_bstride,
                               int B, int H, int W, int Cout, int KH, int KW, int relu, int passes,
                               int device, void *stream) {
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, in_scale, out, out_bstride,
                            B, H, W, Cout, KH, KW, relu, passes, nullptr, device, stream);
}

extern "C" int dkt_conv2d_f16s_strided(const float *const *src, const int *src_channels, const long *src_bstride,
                                       int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                                       float out_scale, float in_scale, float *out, long out_bstride,
                                       int B, int H, int W, int Cout, int KH, int KW, int stride, int relu,
                                       int passes, int device, void *stream) {
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, in_scale, out, out_bstride,
                            B, H, W, Cout, KH, KW, relu, passes, nullptr, device, stream, stride);
}

extern "C" int dkt_conv2d_f16s_gate_zr(const float *const *src, const int *src_channels, const long *src_bstride,
                                       int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                                       float out_scale, float in_scale, const float *cz, long cz_bstride,
                                       const float *cr, long cr_bstride, const float *h, long h_bstride,
                                       float *z, long z_bstride, float *rh, long rh_bstride,
                                       int B, int H, int W, int Ch, int KH, int KW, int passes,
                                       int device, void *stream) {
    if (!cz || !cr || !h || !z || !rh) return DKT_E_NULL;
    if (Ch <= 0 || Ch % 64 != 0) return DKT_E_UNSUPPORTED;   // z and r channels must not share a wave
    ConvEpilogue e = {1, cz, cr, h, cz_bstride, cr_bstride, h_bstride, rh, rh_bstride};
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, in_scale, z, z_bstride,
                            B, H, W, 2 * Ch, KH, KW, 0, passes, &e, device, stream);
}

extern "C" int dkt_conv2d_f16s_gate_out(const float *const *src, const int *src_channels, const long *src_bstride,
                                        int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                                        float out_scale, float in_scale, const float *cq, long cq_bstride,
                                        const float *z, long z_bstride, const float *h, long h_bstride,
                                        float *hout, long hout_bstride,
                                        int B, int H, int W, int Ch, int KH, int KW, int passes,
                                        int device, void *stream) {
    if (!cq || !z || !h || !hout) return DKT_E_NULL;
    if (Ch <= 0) return DKT_E_SHAPE;
    ConvEpilogue e = {2, cq, z, h, cq_bstride, z_bstride, h_bstride, nullptr, 0};
    return conv2d_f16s_impl(src, src_channels, src_bstride, nsrc, w_hi, w_lo, bias, out_scale, in_scale, hout, hout_bstride,
                            B, H, W, Ch, KH, KW, 0, passes, &e, device, stream);
}
#endif  // CONV_TU_PASSES == 0 (ABI entry points)
