// conv_ws.h -- weights-stationary 3x3 convolution for the encoders' 64 -> 64 layers (included by conv2d.hip, PASSES = 3 unit).
//
// Twelve full-resolution 64 -> 64 image-convolutions are half of an encoder pass (core/extractor.py:122-143: conv1 and
// layer1 of BasicEncoder / MultiBasicEncoder run at the input resolution when n_downsample = 2).  With 64 output channels a
// wave tile of the streaming kernels re-reads its weight fragments for every 2 x 32 pixels -- through the vector-memory path
// (conv2d_f16s_kernel) or through LDS behind a ring and a barrier per step (conv_c8_kernel) -- and both end at 0.09-0.12 of
// the MFMA peak.  A 64 -> 64 3x3 layer has only 36 864 weights: split into (hi, lo) fp16 parts they are 144 KB, i.e. 288
// registers per lane when each wave of a 4-wave block keeps the fragments of its 32 output channels x all 576 (tap, ci)
// products.  So:
//   * block = 4 waves, ONE wave per SIMD, the whole 512-register file (arch + acc VGPRs) per wave: 288 weight registers loaded
//     once per launch, 64 accumulators (32 co x 4 rows x 32 columns), the rest staging / fragments;
//   * wave (wm, wn): wm = 32-channel half, wn = row half of the block's 8 rows x 32 columns output tile; blocks are persistent,
//     tiles in an XCD-aware column-major order;
//   * the only operand stream is the activation patch: fp32 NCHW -> registers (buffer loads: the channel offset is a scalar)
//     -> [instance norm + ReLU of the producer] -> (hi, lo) fp16 (conv2d_f16s_kernel's split, bit for bit) -> LDS as
//     [pixel][16 channels], double-buffered per 16-channel chunk, one barrier per chunk of 108 MFMAs per wave;
//   * taps walk column by column: a B fragment (one patch row at one dx) feeds the three dy that use it -- 36 fragment-pair
//     reads per 108 MFMAs; no weight traffic at all inside the loop.
// Same packed weight image, argument block and epilogue contracts (bias / ReLU / output statistics / residual join) as the
// streaming kernel; the summation order per accumulator differs (chunk-major, then dx-major instead of dy-major taps), so the
// results are NOT bit-identical to it -- the same error bound against fp64 (tests/test_gpu_round4.py compares both).
// Measurements and what was tried: DESIGN.md 3.6b.

// (dx, patch row) units of a chunk are issued in 16 groups: single units, except that the last row of a column of taps
// (3 MFMAs on the last accumulator) shares a group with the first row of the next column (3 MFMAs on accumulator 0)
static constexpr int conv_ws_group_first(int g) { return g < 5 ? g : g == 5 ? 5 : g < 10 ? g + 1 : g == 10 ? 11 : g + 2; }
static constexpr int conv_ws_group_units(int g) { return (g == 5 || g == 10) ? 2 : 1; }

// output rows a patch row r feeds: dy = 0..2 with 0 <= r - dy < nf
static constexpr int conv_ws_rows_fed(int r, int nf) {
    int n = 0;
    for (int dy = 0; dy < 3; ++dy) n += (r - dy >= 0 && r - dy < nf) ? 1 : 0;
    return n;
}

typedef _Float16 ws_h2 __attribute__((ext_vector_type(2)));
typedef float ws_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ws_bits(ws_h2 h) {
    union { ws_h2 h; unsigned u; } v;
    v.h = h;
    return v.u;
}

template <int NRM, int EPI>
__global__ __launch_bounds__(256, 1) void conv64_ws_kernel(ConvArgs a) {
    constexpr int HALO = 1;
    constexpr int NF = 4, WN = 2;            // rows per wave, waves along rows
    constexpr int TR = NF * WN;              // 8 output rows per block
    constexpr int PR = TR + 2, PC = 34, NPP = PR * PC;
    constexpr int PITCH = 12;                // words per pixel: 8 (16 fp16) + 4 pad (conflict-free ds_read_b128 / ds_write_b128)
    constexpr int PLANE = NPP * PITCH;
    constexpr int STAGE = 2 * PLANE;         // hi + lo
    constexpr int NCH = 4;                   // 16-channel chunks of the 64 input channels
    constexpr int NG = 16;                   // issue groups per chunk (conv_ws_group_first)
    constexpr int RB = 4;                    // B-fragment ring: reads run two groups ahead
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];   // 2 * STAGE words (+ 8 dummy)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, kg = lane >> 5;
    const long HW = (long)a.H * a.W;
    const int tiles_h = a.tiles_xy / a.tiles_w;

    // XCD-aware persistent stream: the hardware deals consecutive block ids round-robin over the 8 XCDs; logical block
    // lb = (id % 8) * (n / 8) + id / 8 gives every XCD a contiguous range of the tile order, and tiles are ordered
    // column-major (vertical neighbours share 2 of 10 patch rows), so that halo rows are re-read from the XCD's own L2.
    const int nblk = (int)gridDim.x;
    const int lb = (nblk % 8 == 0) ? ((int)blockIdx.x % 8) * (nblk / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    auto decode = [&](int t, int &th0, int &tw0, int &tb) {
        const int xy = t % a.tiles_xy;
        tb = t / a.tiles_xy;
        th0 = (xy % tiles_h) * TR;
        tw0 = (xy / tiles_h) * 32;
    };
    int h0, w0, b;
    int tile = lb;
    decode(tile, h0, w0, b);

    f16x8 Ahi[NCH * 9], Alo[NCH * 9];
    // ---- stationary weights: fragment (chunk c, tap) of this wave's 32 channels, packed image [tap][ci16][64 co][16]
    auto load_weights = [&]() {
        const long wlane = (long)(wm * 32 + li) * 16 + kg * 8;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const long off = ((long)tap * NCH + c) * (64 * 16) + wlane;
                Ahi[c * 9 + tap] = *(const f16x8 *)(a.whi + off);
                Alo[c * 9 + tap] = *(const f16x8 *)(a.wlo + off);
            }
        // register classes by hand: accumulators (64) + 48 of the 72 fragments (4 registers each) fill the 256 acc VGPRs, the other
        // fragments stay in arch VGPRs beside everything the VALU touches (left to itself the allocator shuffles fragments
        // between the two files inside the loop and spills)
#pragma unroll
        for (int i = 0; i < NCH * 9; ++i) {
            if (i < 24) {
                asm volatile("" : "+a"(Ahi[i]));
                asm volatile("" : "+a"(Alo[i]));
            } else {
                asm volatile("" : "+v"(Ahi[i]));
                asm volatile("" : "+v"(Alo[i]));
            }
        }
    };

    // ---- staging (conv2d_f16s_kernel's, CK = 1): wave w stages channels 8*(w & 1).. of the chunk for pixel slots
    // pp = lane + 64 * (2 * it + (w >> 1))
    constexpr int SG = 2, PG = 2;
    constexpr int SIT = (NPP + 64 * PG - 1) / (64 * PG);   // 3
    constexpr int NSL = SIT * 4;                           // 12 slices (pixel slot x channel pair) per chunk
    const int swave = __builtin_amdgcn_readfirstlane(wave) % SG;
    const int spgrp = __builtin_amdgcn_readfirstlane(wave) / SG;
    unsigned spix[SIT];
    float sscale[SIT];                       // in_scale inside the image, 0 outside (the zero padding)
    float sreg[SIT][8];
    unsigned shw[SIT][4], slw[SIT][4];
    __amdgpu_buffer_rsrc_t srsrc;            // the 8 planes this wave stages (wave-uniform descriptor: no 64-bit address math)
    const unsigned plane_bytes = (unsigned)HW * 4u;
    float nmean[NRM ? 8 : 1], ninv[NRM ? 8 : 1];
    auto stage_tile = [&](int th0, int tw0) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int pp = lane + 64 * (it * PG + spgrp);
            const int pr = pp / PC, pc = pp - pr * PC;
            const int ih = th0 - HALO + pr, iw = tw0 - HALO + pc;
            const bool ok = pp < NPP && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            sscale[it] = ok ? a.in_scale : 0.0f;
            spix[it] = ok ? (unsigned)(ih * a.W + iw) * 4u : 0u;
        }
    };
    auto stage_select = [&](int tb, int chunk) {
        const int cb = chunk * 16 + 8 * swave;
        const float *sbase = a.src[0] + (long)tb * a.src_bs[0] + (long)cb * HW;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)sbase);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)sbase >> 32));
        srsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((size_t)hi32 << 32) | lo32), 0, (int)(8u * plane_bytes), 0x00020000);
        if (NRM) {
            // (mean, 1/std) of the 8 planes from the block's LDS copy: wave-uniform LDS reads instead of global loads, which
            // would sit in the in-order memory counter in front of the chunk's activation requests
            const float *np = (const float *)(lds + 2 * STAGE + 8) + 2 * (tb * 64 + cb);
#pragma unroll
            for (int j = 0; j < (NRM ? 8 : 1); ++j) {
                nmean[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(np[2 * j])));
                ninv[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(np[2 * j + 1])));
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
                const int j = 2 * q + jj;
                sreg[it][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srsrc, spix[it], j * plane_bytes, 0));
            }
        }
    };
    auto stage_store = [&](unsigned *buf, int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            if (k >= NSL) break;
            const int it = k >> 2, q = k & 3;
            float u0 = sreg[it][2 * q], u1 = sreg[it][2 * q + 1];
            if (NRM) {     // the arithmetic of instnorm_apply_kernel (norm.hip)
                u0 = dkt_relu(__fmul_rn(__fsub_rn(u0, nmean[(2 * q) % (NRM ? 8 : 1)]), ninv[(2 * q) % (NRM ? 8 : 1)]));
                u1 = dkt_relu(__fmul_rn(__fsub_rn(u1, nmean[(2 * q + 1) % (NRM ? 8 : 1)]), ninv[(2 * q + 1) % (NRM ? 8 : 1)]));
            }
            // (finite values only: a NaN / Inf at a clamped address would reach the padding through 0 * x; check_finite
            // rejects such inputs anyway).  Packed conversions: v_cvt_pk_f16_f32, the split of conv2d_f16s_kernel bit for bit
            const ws_f2 xv = {u0 * sscale[it], u1 * sscale[it]};
            const ws_h2 hv = __builtin_convertvector(xv, ws_h2);
            const ws_f2 rv = {xv[0] - (float)hv[0], xv[1] - (float)hv[1]};
            const ws_h2 lv = __builtin_convertvector(rv, ws_h2);
            shw[it][q] = ws_bits(hv);
            slw[it][q] = ws_bits(lv);
            if (q == 3) {
                const int pp = lane + 64 * (it * PG + spgrp);
                unsigned *dst = pp < NPP ? buf + pp * PITCH + 4 * swave : lds + 2 * STAGE;   // surplus lanes -> dummy
                *(uint4 *)dst = make_uint4(shw[it][0], shw[it][1], shw[it][2], shw[it][3]);
                *(uint4 *)(dst + (pp < NPP ? PLANE : 4)) = make_uint4(slw[it][0], slw[it][1], slw[it][2], slw[it][3]);
            }
        }
    };

    f32x16 acc[NF];
    auto zero_acc = [&]() {
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    };
    zero_acc();

    f16x8 Bhi[RB], Blo[RB];
    // step s = (dx, patch row r): the fragment of patch row wn*NF + r at column offset dx
    auto loadB = [&](int slot, const unsigned *buf, int s) {
        const int dx = s / (NF + 2), r = s % (NF + 2);
        const int pp = (wn * NF + r) * PC + li + dx;
        const unsigned *pb = buf + pp * PITCH + kg * 4;
        Bhi[slot] = *(const f16x8 *)pb;
        Blo[slot] = *(const f16x8 *)(pb + PLANE);
    };
    // patch row r at dx serves output rows n = r - dy, dy = 0..2; pass-major: an accumulator is touched every third MFMA
    auto mma = [&](const int c, const int s) {
        const int dx = s / (NF + 2), r = s % (NF + 2);
        const int slot = s % RB;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int n = r - dy;
            if (n >= 0 && n < NF)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[c * 9 + dy * 3 + dx], Bhi[slot], acc[n], 0, 0, 0);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int n = r - dy;
            if (n >= 0 && n < NF)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Alo[c * 9 + dy * 3 + dx], Bhi[slot], acc[n], 0, 0, 0);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int n = r - dy;
            if (n >= 0 && n < NF)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ahi[c * 9 + dy * 3 + dx], Blo[slot], acc[n], 0, 0, 0);
        }
    };

    // a group's MFMAs, pass-major over its units: the merged groups alternate between the last row of one column of taps
    // (accumulator NF - 1) and the first row of the next (accumulator 0)
    auto mma_group = [&](const int c, const int g) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int u = conv_ws_group_first(g); u < conv_ws_group_first(g) + conv_ws_group_units(g); ++u) {
                const int dx = u / (NF + 2), r = u % (NF + 2);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int n = r - dy;
                    if (n >= 0 && n < NF) {
                        const int t = c * 9 + dy * 3 + dx;
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pass == 1 ? Alo[t] : Ahi[t], pass == 2 ? Blo[u % RB] : Bhi[u % RB],
                                                                        acc[n], 0, 0, 0);
                    }
                }
            }
    };

    // Epilogue.  EPI = 0: bias [+ ReLU]; 1: the same + instance-norm statistics of the values stored (conv2d_f16s_kernel's
    // stats_ws contract, entry = (tile, wn)); 3: residual join out = relu(e_c0 + [relu](v)) (core/extractor.py:60 with the norm
    // folded into the weights).  Stores / residual loads go through buffer descriptors: the channel offset is a scalar, the
    // pixel offset one VGPR per row -- no 64-bit address arithmetic; a scheduling fence per row keeps one row's values live.
    // (Measured and not kept: the values transposed through a wave-private LDS area and stored / the residual fetched as 16-byte
    // accesses -- 16 instead of 64 per wave and tile: epilogue 6 550 against 4 810 cycles, the LDS round trip costs more than the
    // narrower store stream.)
    auto wave_rsrc = [&](const float *base) {
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((size_t)hi32 << 32) | lo32), 0, (int)(32u * plane_bytes), 0x00020000);
    };
    auto epilogue = [&]() {
        const int co_lane = wm * 32 + 4 * kg;           // co = co_lane + (r & 3) + 8 * (r >> 2)
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = a.bias ? a.bias[co_lane + (r & 3) + 8 * (r >> 2)] : 0.0f;
        const auto orsrc = wave_rsrc(a.out + (long)b * a.out_bs + (long)(wm * 32) * HW);
        auto row_off = [&](int n, bool &ok) {            // byte offset of (channel 4*kg, row n, column li) in the wave's 32 planes
            const int oh = h0 + wn * NF + n, ow = w0 + li;
            ok = oh < a.H && ow < a.W;
            return ok ? (unsigned)(4 * kg) * plane_bytes + (unsigned)(oh * a.W + ow) * 4u : 0x80000000u;   // outside: dropped
        };
        if (EPI == 3) {
            // The wave's memory counter is in order: a residual request issued behind a row's stores is not back before those
            // stores have reached memory (requests one row ahead measured ~2 500 cycles of waiting per row).  Rows 0 and 1 are
            // requested up front; row n + 2 is requested once row n's results are computed (its registers are free) and BEFORE
            // row n's stores are issued.  (All four rows up front need 64 registers: the allocator spills weights for them.)
            const auto crsrc = wave_rsrc(a.e_c0 + (long)b * a.e_c0_bs + (long)(wm * 32) * HW);
            float gc[2][16];
            unsigned vo[NF];
            auto request = [&](int n) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    gc[n & 1][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crsrc, vo[n], ((r & 3) + 8 * (r >> 2)) * plane_bytes, 0));
            };
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                bool ok;
                vo[n] = row_off(n, ok);
            }
            request(0);
            request(1);
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[n][r] * a.out_scale + bv[r];
                    o[r] = dkt_relu(__fadd_rn(gc[n & 1][r], a.relu ? dkt_relu(v) : v));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (n + 2 < NF) request(n + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), orsrc, vo[n], ((r & 3) + 8 * (r >> 2)) * plane_bytes, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        float s1[EPI == 1 ? 16 : 1], s2[EPI == 1 ? 16 : 1];
#pragma unroll
        for (int i = 0; i < (EPI == 1 ? 16 : 1); ++i) s1[i] = s2[i] = 0.0f;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            bool ok;
            const unsigned vo = row_off(n, ok);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[n][r] * a.out_scale + bv[r];
                if (a.relu) v = dkt_relu(v);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, vo, ((r & 3) + 8 * (r >> 2)) * plane_bytes, 0);
                if (EPI == 1) {
                    v = ok ? v : 0.0f;
                    s1[r] = __fadd_rn(s1[r], v);
                    s2[r] = __fadd_rn(s2[r], __fmul_rn(v, v));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (EPI == 1) {
            // per-lane sums over the NF pixels -> one add across the half-wave's two 16-lane groups -> reduce-scatter over 16
            // lanes (15 exchanges per statistic): lane li & 15 ends with value index li & 15
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s1[i] = __fadd_rn(s1[i], __shfl_xor(s1[i], 16, 32));
                s2[i] = __fadd_rn(s2[i], __shfl_xor(s2[i], 16, 32));
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {
                const bool up = (li & d) != 0;
#pragma unroll
                for (int j = 0; j < d; ++j) {
                    const float k1 = up ? s1[j + d] : s1[j], g1 = up ? s1[j] : s1[j + d];
                    const float k2 = up ? s2[j + d] : s2[j], g2 = up ? s2[j] : s2[j + d];
                    s1[j] = __fadd_rn(k1, __shfl_xor(g1, d, 32));
                    s2[j] = __fadd_rn(k2, __shfl_xor(g2, d, 32));
                }
            }
            const int vi = li & 15;
            const int co = co_lane + (vi & 3) + 8 * (vi >> 2);
            const long e = (long)((h0 / TR) * a.tiles_w + w0 / 32) * WN + wn;
            if (li < 16) {
                float *p = a.stats_ws + (((long)b * a.tiles_xy * WN + e) * 64 + co) * 2;
                p[0] = s1[0];
                p[1] = s2[0];
            }
        }
    };

    // ---- main stream.  A chunk's 18 (dx, patch row) units are issued as 16 groups; a group carries the fragment reads of the
    // group after next, its share of the NEXT chunk's staging (fp32 requests in groups 0..5, convert + LDS writes in groups
    // 9..15: a request has ~60 MFMAs to come back) and its 3 / 6 / 9 MFMAs.
    if (NRM) {
        // every image's 64 x (mean, 1/std) behind the two stages (launch_conv_ws sizes the allocation: B * 512 bytes)
        float *np = (float *)(lds + 2 * STAGE + 8);
        for (int i = tid; i < a.n_co * 128; i += 256) np[i] = a.in_norm[i];
        __syncthreads();
    }
    stage_tile(h0, w0);
    stage_select(b, 0);
    stage_load(0, NSL);
    stage_store(lds, 0, NSL);
    __builtin_amdgcn_sched_barrier(0);        // the prologue's staging registers are free before the 288 weight registers fill
    load_weights();
    __syncthreads();
    loadB(0, lds, 0);
    loadB(1, lds, 1);
    for (;;) {
        const int tn = tile + nblk;
        const bool have_next = tn < a.total_tiles;
        int nh0 = h0, nw0 = w0, nb = b;
        if (have_next) decode(tn, nh0, nw0, nb);
        // @trace(0)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned *cur = lds + (c & 1) * STAGE;
            unsigned *nxt = lds + ((c + 1) & 1) * STAGE;
            if (c + 1 < NCH) {
                stage_select(b, c + 1);
            } else {
                // the next tile's first chunk; the block's very last chunk stages its own tile's chunk 0 again, harmlessly
                stage_select(nb, 0);
                stage_tile(nh0, nw0);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                // fragment reads run two groups ahead
                if (g + 2 < NG) {
#pragma unroll
                    for (int u = conv_ws_group_first(g + 2); u < conv_ws_group_first(g + 2) + conv_ws_group_units(g + 2); ++u)
                        loadB(u % RB, cur, u);
                }
                if (g < 6) stage_load(2 * g, 2 * g + 2);                                   // slices 2g, 2g + 1
                else if (g == 9) stage_store(nxt, 0, 1);
                else if (g >= 10 && g < 15) stage_store(nxt, 2 * (g - 10) + 1, 2 * (g - 10) + 3);
                else if (g == 15) stage_store(nxt, 11, 12);
                mma_group(c, g);
                if (conv_ws_group_units(g) == 1 && conv_ws_rows_fed(conv_ws_group_first(g) % (NF + 2), NF) == 1) {
                    // three MFMAs on ONE accumulator: adjacent (an issue slot between two dependent MFMAs costs ~40 cycles)
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                } else {
                    int nm = 0;
#pragma unroll
                    for (int u = conv_ws_group_first(g); u < conv_ws_group_first(g) + conv_ws_group_units(g); ++u)
                        nm += 3 * conv_ws_rows_fed(u % (NF + 2), NF);
#pragma unroll
                    for (int i = 0; i < nm; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x320, 1, 0);                    // VMEM read / DS read / DS write
                        if (g >= 9) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);        // VALU (the conversion groups)
                        else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (a raw barrier behind the wave's own LDS traffic: __syncthreads() also drains the memory counter -- the activation
            // requests of the next chunk and the previous tile's stores -- at every chunk)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // @trace(1 + c)
            if (c + 1 < NCH) {           // (the next tile's first fragments are read behind the epilogue: 16 registers it needs)
                loadB(0, nxt, 0);
                loadB(1, nxt, 1);
            }
        }
        epilogue();
        // @trace(5)
        if (!have_next) break;
        tile = tn; h0 = nh0; w0 = nw0; b = nb;
        loadB(0, lds, 0);
        loadB(1, lds, 1);
        zero_acc();
    }
}

// 64 -> 64, 3x3, stride 1, one 64-channel source, epilogues 0 / 3: images with at least 192 tiles (8 x 32 pixels) take the
// weights-stationary kernel (DKT_CONV_WS=0 keeps the streaming kernel: A/B and bit-identity tests of the round-2 path)
static bool conv_ws_enabled() {
    const char *e = getenv("DKT_CONV_WS");      // read per launch: the tests compare both kernels in one process
    return !(e && e[0] == '0');
}

static bool conv_ws_eligible(const ConvArgs &a, int B) {
    if (!conv_ws_enabled()) return false;
    if (a.nsrc != 1 || a.src_ch[0] != 64 || a.Cout != 64 || a.nch16 != 4 || a.CoutPad != 64) return false;
    if (a.epi != 0 && a.epi != 3) return false;
    if (a.Ho != a.H || a.Wo != a.W || B > 64) return false;
    if ((long)a.H * a.W * 128 > 0x7fffffffL) return false;     // 32 output planes per buffer descriptor, byte offsets in 32 bits
    const long tiles = (long)a.tiles_w * ((a.H + 7) / 8) * B;
    return tiles >= 192;           // measured: 22 against 30 us at 230 tiles (184 x 312), equal at 128, 35 against 45-48 at 512
}

static int launch_conv_ws(ConvArgs a, int B, hipStream_t st) {
    constexpr int STAGE = 10 * 34 * 12 * 2;
    const size_t lds = ((size_t)2 * STAGE + 8) * sizeof(unsigned) + (a.in_norm ? (size_t)B * 512 : 0);
    static int ready[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int epi = a.epi == 3 ? 3 : (a.stats_ws ? 1 : 0);
    const void *kern = nullptr;
    if (a.in_norm) kern = epi == 3 ? (const void *)conv64_ws_kernel<1, 3> : epi == 1 ? (const void *)conv64_ws_kernel<1, 1> : (const void *)conv64_ws_kernel<1, 0>;
    else kern = epi == 3 ? (const void *)conv64_ws_kernel<0, 3> : epi == 1 ? (const void *)conv64_ws_kernel<0, 1> : (const void *)conv64_ws_kernel<0, 0>;
    if (!ready[dev & 63]) {
        const void *all[6] = {(const void *)conv64_ws_kernel<0, 0>, (const void *)conv64_ws_kernel<0, 1>, (const void *)conv64_ws_kernel<0, 3>,
                              (const void *)conv64_ws_kernel<1, 0>, (const void *)conv64_ws_kernel<1, 1>, (const void *)conv64_ws_kernel<1, 3>};
        for (int i = 0; i < 6; ++i) {
            hipError_t e = hipFuncSetAttribute(all[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)2 * STAGE + 8) * sizeof(unsigned)) + 64 * 512);
            if (e != hipSuccess) return (int)e;
        }
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        ready[dev & 63] = cus;
    }
    a.tiles_xy = a.tiles_w * ((a.H + 7) / 8);
    a.n_co = B;                       // (the kernel has one channel block; the field carries the batch size for the in_norm copy)
    const long total = (long)a.tiles_xy * B;
    if (total > 0x7fffffffL) return DKT_E_SHAPE;
    a.total_tiles = (int)total;
    const long cap = ready[dev & 63];
    // no more blocks than the rounds need (3588 tiles are 15 rounds on 256 CUs and on 240): the spare CUs stay free for the
    // other encoder's stream; a multiple of 8 keeps the XCD mapping
    long nblk = total > cap ? cap : total;
    if (total > cap) {
        const long rounds = (total + cap - 1) / cap;
        const long need = ((total + rounds - 1) / rounds + 7) & ~7L;
        if (need < nblk) nblk = need;
    }
    void *params[1] = {(void *)&a};
    (void)hipLaunchKernel(kern, dim3((unsigned)nblk), dim3(256), params, lds, st);
    int rc = dkt_launch_status();
    if (rc == DKT_OK && a.stats_ws)
        rc = conv_stats_reduce(a.stats_ws, a.stats_part, B, a.Cout, (long)a.tiles_xy * 2, (long)a.H * a.W, st);
    return rc;
}
