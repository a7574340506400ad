// conv_c8.hip -- round-3 3x3 convolution of the update operator (core/update.py:16-32, 64-85, 6-13) on gfx950:
// the same split-fp16 arithmetic as conv2d.hip (w*x ~= w_hi*x_hi + w_lo*x_hi + w_hi*x_lo, fp32 accumulation, three
// v_mfma_f32_32x32x16_f16 per product block), rebuilt around the measured limiter of that kernel (the
// weight stream through the vector-memory path, staging VALU work and registers, one drained barrier per chunk):
//
//   * ACTIVATIONS ARRIVE PRE-SPLIT ("C8S" layout): the producing kernel's epilogue writes fp16 (hi, lo) pairs,
//     [batch][channel group of 8][hi|lo][Hp][Wp][8 channels] with a zero border (Hp = roundup(H,8)+2,
//     Wp = roundup(W,32)+2) -- the same bytes as fp32.  A (rows+2) x 34 pixel patch of a 16-channel chunk is then
//     copied global -> LDS by the DMA path (global_load_lds, 16 B per lane, no registers, no VALU, no bounds
//     logic: the border is part of the tensor) into the exact image the B fragments are read from
//     ([plane][pixel] x 16 B: conflict-free ds_read_b128).
//   * WEIGHTS go through LDS too, shared by all waves of the block: per (chunk, tap) step one contiguous
//     [co64 block][hi|lo][k8][64 co][8] image (4 KB per 64 output channels) is DMA'd into a 3-slot ring two steps
//     ahead.  Per block and step the vector-memory path carries 16 KB instead of the 32 KB the per-wave weight loads
//     of conv2d.hip pull through L1 (and blocks are twice as large, so half as many of them stream the image).
//   * ONE raw s_barrier per step, counted vmcnt AND counted lgkmcnt, no drained waits: DMA and fragment reads stay
//     in flight across barriers.  Fragment registers are single-buffered: every fragment is re-read into its own
//     register as soon as its last MFMA of the step has issued, one read per MFMA issue shadow (see C8_STEP).
//   * wave tile 64 co x NF rows x 32 columns (NF = 4: 128 accumulators); steps walk the taps column by column so that
//     a B fragment (one patch row) serves three steps: 24 MFMAs per 8 fragment reads.
//
// Outputs: fp32 NCHW and / or C8S (the next convolution's operand), fused ConvGRU gate epilogues as conv2d.hip.
#include "dkt_common.h"
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define C8_MAX_SRC 4
#define C8_PC 34                 // patch columns: 32 + halo

struct C8Args {
    const char *src[C8_MAX_SRC];      // C8S tensors (all with the same Hp, Wp)
    long src_bs[C8_MAX_SRC];          // bytes per batch item
    int src_n16[C8_MAX_SRC];          // 16-channel chunks per source
    int nsrc, nchunks;
    long plane_bytes;                 // Hp * Wp * 16
    int Wp;
    const char *w;                    // [chunk][tap][co64][hi|lo][k8][64][8] fp16
    int n_co64;
    const float *bias;
    float out_scale;
    int H, W, Cout;
    int tiles_w, tiles_xy, n_co, total_tiles;
    int relu, epi;
    float *out;  long out_bs;          // fp32 NCHW destination (or null)
    char *out_c8; long out_c8_bs, out_c8_plane; int out_c8_Wp, out_c8_ch0;   // C8S destination (or null)
    float act_scale;                   // power of two applied before the fp16 split of C8S outputs
    const float *e_c0, *e_c1, *e_h;    // gate operands (fp32 NCHW), see conv2d.hip
    long e_c0_bs, e_c1_bs, e_h_bs;
    float *out2; long out2_bs;         // epi 1: r*h as fp32 NCHW (or null)
    char *out2_c8; long out2_c8_bs;    // epi 1: r*h as C8S (same Hp/Wp/plane as out_c8; channel offset out2_c8_ch0)
    int out2_c8_ch0;
    // epi 3 (flow / disparity head, core/update.py:9-13 with 1 or 2 outputs): the layer's ReLU output is never written;
    // its epilogue reduces it over the block's channels against the NEXT layer's 3x3 weights, one plane per (output, tap):
    // head_out[b][(o * n_co + co block) * 9 + tap][H][W] = sum_co head_w[o][co][tap] * relu(conv)[co]; dkt_head_finish sums
    // the shifted planes.
    const float *head_w;               // [n_out][Cout][12] (9 taps + 3 pad)
    float *head_out; long head_out_bs; int head_nout;
    int f32_c4;                        // the fp32 tensors above (out, out2, e_c0, e_c1, e_h) are [B][C/4][H][W][4] instead of NCHW
    const float *tail; long tail_bs; int tail_ch;   // epi 0: channels Cout .. Cout+tail_ch-1 of the C8S output are copied from here
    float tail_ratio;                  // ... times tail_scale / act_scale (the tail channels may carry their own power-of-two scale)
};

struct C8ArgsPair {
    C8Args p[2];
};

// Flag round between the two stages of conv_c8_chain_kernel (gru_c8.hip's protocol): a stage-0 tile publishes its flag word
// (launch count) behind an agent-scope release of its stores; a stage-1 tile waits, before its first patch is requested, for the
// stage-0 tiles that produce the patch rows and columns it reads (its own tile, the halo, every channel block, both problems).
// pub / dep null: plain launch.
struct C8Sync {
    unsigned *pub;              // [problem][tile] words of the stage that publishes
    const unsigned *dep;        // ... the same words, as the stage that waits sees them
    int dep_tr;                 // rows per tile of the publishing stage
    int dep_tiles_w, dep_tiles_h, dep_nco, dep_nprob;      // its tile grid, channel blocks and problems
    int dep_tiles_per_prob;     // tiles_xy * n_co * B of one publishing problem
    unsigned *err;
    unsigned target;            // the launch count this launch publishes (set by the kernel from the block's own first word)
    int timing_only;            // 1: no waits at all (an upper bound of what the fusion can buy; results are wrong)
};

__device__ __forceinline__ float c8_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float c8_tanh(float x) {
    const float xc = x < -15.0f ? -15.0f : (x > 15.0f ? 15.0f : x);      // NaN passes through
    const float t = __expf(2.0f * xc);
    return (t - 1.0f) * __frcp_rn(t + 1.0f);
}
__device__ __forceinline__ unsigned c8_pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; unsigned u; } v;
    v.h[0] = a;
    v.h[1] = b;
    return v.u;
}

template <int N>
__device__ __forceinline__ void c8_wait_vm() {
    // vmcnt(N) only (expcnt / lgkmcnt fields at their no-wait maxima).  The builtin, not inline asm: hipcc keeps its own
    // LDS-read bookkeeping across it, so the first pass after the barrier waits for ITS fragments only (counted lgkmcnt)
    static_assert(N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// Fragment reads are inline asm with hand-counted lgkmcnt waits: hipcc answers every LDS read that is in flight
// across the step's barrier with lgkmcnt(0) at the first MFMA behind it, which exposed the latency of the six reads
// issued just before the barrier in EVERY step (ablation: 296 us with the reads, 200 us without, MFMAs alone 182 us).
template <int OFF>
__device__ __forceinline__ void c8_lds_read(f16x8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void c8_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// WM x WN waves along (output channels, rows); wave tile 64 co x NF rows x 32 columns.
// RING = weight ring slots: step s computes from slot s % RING while the images of steps s+1 .. s+RING-1 are in the ring or
// on their way (short steps -- small wave tiles -- need the deeper rings to cover the DMA latency).
// PASSES (round 5): 3 = the fp32-class product above; 2 = w_hi*x_hi + w_lo*x_hi (activations rounded to fp16: their lo planes are
// never read); 1 = w_hi*x_hi only (plain fp16 operands).  Reduced-pass launches are NOT parity paths by themselves: they serve the
// precision schedules of loop_c8 (early refinement iterations at fewer passes).  The DMA stream, the ring and the
// vmcnt waits are the same in all three; what changes is which fragments are read and multiplied (and the lgkmcnt counts).
// The kernel's body as a device function (round 5): `nblocks` persistent blocks, this one `bid`, walk the tiles of the launch's
// one or two problems; `lds` = the block's dynamic LDS.  conv_c8_kernel is this body; conv_c8_chain_kernel runs two of them
// back to back in one launch (dependent layers with a flag round instead of a kernel boundary, see C8Sync).
template <int WM, int WN, int NF, int RING, int PASSES>
__device__ __forceinline__ void conv_c8_body(const C8ArgsPair &ap, const int nb0, const int nblocks, const int bid, char *const lds,
                                             const C8Sync &sy) {
    static_assert(PASSES >= 1 && PASSES <= 3, "passes");
    constexpr int C8_RING = RING;
    constexpr int NW = WM * WN;
    constexpr int TR = WN * NF;                      // output rows per block
    constexpr int PR = TR + 2;
    constexpr int NPP = PR * C8_PC;                  // patch pixels
    constexpr int NU = NPP * 4;                      // 16-byte units per chunk: (2 k8 groups) x (hi, lo)
    constexpr int NPR = (NU + 63) / 64;              // 1-KiB DMA pieces that carry data
    constexpr int NIA = (NPR + NW - 1) / NW;         // activation DMA pieces per wave and chunk (every wave issues the same
                                                     // number -- the vmcnt waits count them; surplus pieces land in one slack KiB)
    constexpr int ACT_BYTES = (NIA * NW > NPR ? NPR + 1 : NPR) * 1024;
    constexpr int WSLOT = WM * 4096;
    constexpr int WPI = (WM * 4) / NW;               // weight DMA pieces per wave and step
    static_assert((WM * 4) % NW == 0, "weight image must split evenly over the waves");
    constexpr int MF = 2;
    char *const lds_act = lds;                       // act[2][ACT_BYTES] | wring[C8_RING][WSLOT]
    char *const lds_w = lds + 2 * ACT_BYTES;

    const bool second = bid >= nb0;
    const C8Args &a = ap.p[second ? 1 : 0];
    const int blk_first = second ? nb0 : 0;
    const int blk_count = second ? nblocks - nb0 : nb0;
    if (blk_count <= 0 || bid - blk_first >= a.total_tiles) return;      // (a chain stage with fewer tiles than the launch has blocks)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int li = lane & 31, kg = lane >> 5;
    const long HW = (long)a.H * a.W;

    auto decode = [&](int t, int &th0, int &tw0, int &tco, int &tb) {
        const int xy = t % a.tiles_xy, r = t / a.tiles_xy;
        tw0 = (xy % a.tiles_w) * 32;
        th0 = (xy / a.tiles_w) * TR;
        tco = (r % a.n_co) * (64 * WM);
        tb = r / a.n_co;
    };
    int h0, w0, co_blk, b;
    int tile = bid - blk_first;       // (the XCD-aware order of the fused ConvGRU launch measured -0.6 % here: r05_cfg_variants.txt)
    decode(tile, h0, w0, co_blk, b);

    // ---- chain launches (C8Sync): wave 0 polls the flags of the stage-0 tiles a stage-1 tile reads from
    typedef __attribute__((address_space(1))) unsigned c8_gu32;
    auto wait_deps = [&](int th0, int tw0, int tb) {
        if (!sy.dep || sy.timing_only) return;
        if (wave == 0) {
            const int r0 = (th0 > 0 ? th0 - 1 : 0) / sy.dep_tr;
            int r1 = (th0 + TR) / sy.dep_tr;
            r1 = r1 < sy.dep_tiles_h ? r1 : sy.dep_tiles_h - 1;
            const int c0 = tw0 / 32 > 0 ? tw0 / 32 - 1 : 0;
            int c1 = tw0 / 32 + 1;
            c1 = c1 < sy.dep_tiles_w ? c1 : sy.dep_tiles_w - 1;
            const int nr = r1 - r0 + 1, nc = c1 - c0 + 1;
            const int per = nr * nc, n = per * sy.dep_nco * sy.dep_nprob;          // <= 64 (checked by the host)
            int k = lane < n ? lane : 0;
            const int cc = k % nc; k /= nc;
            const int rr = k % nr; k /= nr;
            const int co = k % sy.dep_nco, pr = k / sy.dep_nco;
            const long idx = (long)pr * sy.dep_tiles_per_prob +
                             ((long)tb * sy.dep_nco + co) * (sy.dep_tiles_w * sy.dep_tiles_h) + (r0 + rr) * sy.dep_tiles_w + (c0 + cc);
            c8_gu32 *f = (c8_gu32 *)sy.dep + idx;
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = lane >= n || (int)(v - sy.target) >= 0;
                if (__all(ok)) break;
                if (spins > (1u << 21)) {
                    if (lane == 0 && sy.err) __hip_atomic_fetch_or((c8_gu32 *)sy.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bit 1: a chain launch
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    };
    auto publish = [&](int t) {
        if (!sy.pub || sy.timing_only) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // every wave: its stores of this tile are visible device-wide
        __builtin_amdgcn_s_barrier();
        if (tid == 0)
            __hip_atomic_store((c8_gu32 *)sy.pub + (long)(second ? sy.dep_tiles_per_prob : 0) + t, sy.target, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- activation DMA: piece p = j * NW + wave covers units u = 64 p + lane of the chunk image
    //      [q = 2 kg + hl][patch pixel]; the source offset of a lane inside the chunk's 4 planes is fixed per tile.
    unsigned aoff_cur[NIA], aoff_nxt[NIA];
    auto tile_offsets = [&](int th0, int tw0, unsigned (&off)[NIA]) {
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            int u = 64 * (j * NW + wave) + lane;
            u = u < NU ? u : 0;                              // slack lanes of the last piece re-read unit 0 into slack LDS
            const int q = u / NPP, pp = u - q * NPP;
            const int pr = pp / C8_PC, pc = pp - pr * C8_PC;
            off[j] = (unsigned)(q * a.plane_bytes + ((long)(th0 + pr) * a.Wp + (tw0 + pc)) * 16);
        }
    };
    auto chunk_base = [&](int tb, int chunk) -> const char * {       // wave-uniform
        int s = 0, c = chunk;
        while (s + 1 < a.nsrc && c >= a.src_n16[s]) {
            c -= a.src_n16[s];
            ++s;
        }
        return a.src[s] + (long)tb * a.src_bs[s] + (long)c * 4 * a.plane_bytes;
    };
    auto issue_act = [&](const char *base, const unsigned (&off)[NIA], int buf) {
        char *dst = lds_act + buf * ACT_BYTES;
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            __builtin_amdgcn_global_load_lds((const void *)(base + off[j]),
                                             (__attribute__((address_space(3))) void *)(dst + min(j * NW + wave, NPR) * 1024), 16, 0, 0);
    };
    // ---- weight DMA: the step image of this block's WM co64 blocks is contiguous (WM * 4 KB)
    // The images of consecutive steps of one tile are wstep bytes apart ((chunk, tap) is the outer index).
    auto w_tile = [&](int tco) -> const char * {
        int c64 = tco >> 6;
        c64 = c64 + WM <= a.n_co64 ? c64 : 0;                // blocks past the padded channel count idle on valid data
        return a.w + (long)c64 * 4096;
    };
    const long wstep = (long)a.n_co64 * 4096;
    auto issue_w = [&](const char *img, int slot) {
        char *dst = lds_w + slot * WSLOT;
#pragma unroll
        for (int j = 0; j < WPI; ++j) {
            const int p = wave * WPI + j;
            __builtin_amdgcn_global_load_lds((const void *)(img + p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
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

    // fragment addresses: A = slot + wm*4096 + hl*2048 + kg*1024 + (m*32 + li)*16
    //                     B = act + ((2 kg + hl) * NPP + (wn*NF + n + dy) * 34 + li + dx) * 16
    const int a_lane = wm * 4096 + kg * 1024 + li * 16;
    const int b_lane = (2 * kg * NPP + wn * NF * C8_PC + li) * 16;
    f16x8 Ahi[MF], Alo[MF], Bhi[NF + 2], Blo[NF + 2];      // B: one fragment per patch ROW of the current tap column (see C8_STEP)
    const unsigned lds_w_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds_w + a_lane;
    const unsigned lds_b_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds_act + b_lane;
    // One MFMA / one fragment read, each pinned in program order: the step below places at most one LDS read in the
    // issue shadow of each MFMA (clusters of six reads between passes cost ~90 us of the 290 on the 384 -> 256 layer --
    // not their waits, their issue).  Offsets must be literals for the asm immediates: macros, not loops.
#define C8_MM(A, m, B, r, n)                                                                       \
    {                                                                                              \
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[m], B[r], acc[m][n], 0, 0, 0);        \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#define C8_RD(dst, off, addr)                                  \
    {                                                          \
        c8_lds_read<(off)>(dst, addr);                         \
        __builtin_amdgcn_sched_barrier(0);                     \
    }
#define C8_RDL(dst, off, addr)                                            \
    {                                                                     \
        c8_lds_read<(off)>(dst, addr);                                    \
        __builtin_amdgcn_sched_barrier(0);                                \
    }
#define C8_ROW(r, dx, plane) ((plane) + ((r) * C8_PC + (dx)) * 16)      /* fragment of patch row r at tap column dx */

    // ------------------------------------------------------------------------------------------------
    // epilogue (after the tile's last step; the next tile's first DMA pieces are already in flight)
    // ------------------------------------------------------------------------------------------------
    // A lane holds, per accumulator block (m, n), 16 values v[r]: channel (r&3) + 8*(r>>2) + 4*kg of its block, pixel
    // (row n, column li) -- i.e. four groups j = r>>2 of four CONSECUTIVE channels.  fp32 tensors on the epilogue side
    // (gate operands, state, z) may therefore be kept in the "C4" layout [B][C/4][H][W][4]: one 16-byte access per
    // group instead of four strided 4-byte ones (a.f32_c4); NCHW remains for tensors other kernels read.
    // C8S outputs need 8 consecutive channels per 16 bytes: a pair of groups (j, j+1) is completed by exchanging
    // halves with lane ^ 32 (v_permlane32_swap), after which the lane stores group 2*jp + kg of its block.
    auto store_c8_pair = [&](char *dst_b, int ch_block, int g_end, int jp, int oh, int ow, bool inside, const float (&va)[4], const float (&vb)[4]) {
        unsigned ha[2], la[2], hb[2], lb[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float x0 = va[2 * d] * a.act_scale, x1 = va[2 * d + 1] * a.act_scale;
            const float y0 = vb[2 * d] * a.act_scale, y1 = vb[2 * d + 1] * a.act_scale;
            const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1, b0 = (_Float16)y0, b1 = (_Float16)y1;
            ha[d] = c8_pack_h2(a0, a1);
            la[d] = c8_pack_h2((_Float16)(x0 - (float)a0), (_Float16)(x1 - (float)a1));
            hb[d] = c8_pack_h2(b0, b1);
            lb[d] = c8_pack_h2((_Float16)(y0 - (float)b0), (_Float16)(y1 - (float)b1));
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            auto r = __builtin_amdgcn_permlane32_swap(ha[d], hb[d], false, false);
            ha[d] = r[0]; hb[d] = r[1];
            auto q = __builtin_amdgcn_permlane32_swap(la[d], lb[d], false, false);
            la[d] = q[0]; lb[d] = q[1];
        }
        const int g = (ch_block >> 3) + 2 * jp + kg;
        if (inside && g < g_end) {            // (groups past the destination's padded channel count do not exist)
            char *p = dst_b + (long)g * 2 * a.out_c8_plane + ((long)(oh + 1) * a.out_c8_Wp + (ow + 1)) * 16;
            *(uint4 *)p = make_uint4(ha[0], ha[1], hb[0], hb[1]);
            *(uint4 *)(p + a.out_c8_plane) = make_uint4(la[0], la[1], lb[0], lb[1]);
        }
    };
    // address of the 4 channels [c4, c4+4) of pixel px in an fp32 tensor (batch base already applied)
    auto f32_ptr = [&](const float *base, int c4, long px, int iHW) -> const float * {
        return a.f32_c4 ? base + ((long)(c4 >> 2) * iHW + px) * 4 : base + (long)c4 * iHW + px;
    };
    auto f32_load4 = [&](const float *p, int iHW, float (&v)[4]) {
        if (a.f32_c4) {
            const float4 t = *(const float4 *)p;
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = p[(long)i * iHW];
        }
    };
    auto f32_store4 = [&](float *p, int iHW, const float (&v)[4], int nvalid) {
        if (a.f32_c4) *(float4 *)p = make_float4(v[0], v[1], v[2], v[3]);      // (C4 tensors are padded to 4 channels)
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nvalid) p[(long)i * iHW] = v[i];
        }
    };
    char *const lds_head = lds + 2 * ACT_BYTES + C8_RING * WSLOT;      // epi 3 only: [wave][NF][9][32] floats
    auto epilogue_head = [&]() {
        const int co_w = co_blk + wm * 64;
        const bool idle = co_w >= a.n_co64 * 64;
        const int iHW = (int)HW;
        float *red = (float *)lds_head;
        int kgo = kg;                          // the lane's channel half, opaque from here on: nothing derived from it
        asm volatile("" : "+v"(kgo));          // can be pre-computed at kernel entry and kept live (spilled) through the main loop
        // ONE output (stereo: the x component / the disparity), two tile rows at a time: the accumulators, the next tile's
        // prefetched fragments and 18 partial sums fit the register file (all NF rows at once, or a loop over outputs that
        // keeps every accumulator live, spilled 250-400 registers)
        constexpr int NH = NF > 1 ? 2 : 1;
#pragma unroll
        for (int n0 = 0; n0 < NF; n0 += NH) {
            float P[NH][9];
#pragma unroll
            for (int n = 0; n < NH; ++n)
#pragma unroll
                for (int t = 0; t < 9; ++t) P[n][t] = 0.0f;
            if (!idle) {
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // channel co = cu + 4 kg: wave-uniform base pointers + one lane offset (Cout is a multiple of 64 here)
                        const int cu = __builtin_amdgcn_readfirstlane(co_w) + m * 32 + (r & 3) + 8 * (r >> 2);
                        const float *wp = a.head_w + (long)cu * 12 + kgo * 48;
                        const float4 w0 = *(const float4 *)wp, w1 = *(const float4 *)(wp + 4);
                        const float w8 = wp[8];
                        const float wt[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w8};
                        const float bias = a.bias[cu + 4 * kgo];                 // (bias is mandatory here: a branch per channel wrecks the allocation)
#pragma unroll
                        for (int n = 0; n < NH; ++n) {
                            const float v = dkt_relu(acc[m][n0 + n][r] * a.out_scale + bias);
#pragma unroll
                            for (int t = 0; t < 9; ++t) P[n][t] = __fmaf_rn(wt[t], v, P[n][t]);
                        }
                        __builtin_amdgcn_sched_barrier(0);       // one channel at a time: no hoisting of all the weight loads
                    }
            }
            // the two half-waves hold different channels of the same pixels
#pragma unroll
            for (int n = 0; n < NH; ++n)
#pragma unroll
                for (int t = 0; t < 9; ++t) P[n][t] = __fadd_rn(P[n][t], __shfl_xor(P[n][t], 32));
            if (kg == 0) {
#pragma unroll
                for (int n = 0; n < NH; ++n)
#pragma unroll
                    for (int t = 0; t < 9; ++t) red[((wave * NF + n0 + n) * 9 + t) * 32 + li] = P[n][t];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // wave (wm = 0, wn) sums the WM channel quarters of its rows in wave order and writes the planes
        if (wm == 0) {
            for (int item = lane; item < NF * 9 * 32; item += 64) {
                const int px = item & 31, t = (item >> 5) % 9, n = item / (9 * 32);
                float sum = red[(((0 + WM * wn) * NF + n) * 9 + t) * 32 + px];
#pragma unroll
                for (int q = 1; q < WM; ++q) sum = __fadd_rn(sum, red[(((q + WM * wn) * NF + n) * 9 + t) * 32 + px]);
                const int oh = h0 + wn * NF + n, ow = w0 + px;
                if (oh < a.H && ow < a.W)
                    a.head_out[(long)b * a.head_out_bs + ((long)(co_blk / (64 * WM)) * 9 + t) * iHW + (long)oh * a.W + ow] = sum;
            }
        }
        __builtin_amdgcn_s_barrier();                            // `red` is free again before the next tile's epilogue
    };
    auto epilogue = [&]() {
        if constexpr (NW == 8) {                 // (the two 8-wave shapes carry the head epilogue; round 4: in the NF = 4
            if (a.epi == 3) {                    //  shapes its code costs hundreds of spilled registers)
                epilogue_head();
                return;
            }
        }
        const int co_w = co_blk + wm * 64;
        if (co_w >= a.n_co64 * 64) return;                   // idle wave
        const int iHW = (int)HW;
        const int Ch = a.epi == 1 ? a.Cout / 2 : a.Cout;
        const bool rpart = a.epi == 1 && co_w >= Ch;         // wave-uniform: this wave owns r channels
        const int cw = co_w - (rpart ? Ch : 0);              // first channel of this wave inside the Ch-wide tensors
        // wave-uniform selections
        const float *pc = a.epi ? (rpart ? a.e_c1 + (long)b * a.e_c1_bs : a.e_c0 + (long)b * a.e_c0_bs) : nullptr;
        const float *pz = a.epi == 2 ? a.e_c1 + (long)b * a.e_c1_bs : nullptr;
        const float *ph = (a.epi == 2 || rpart) ? a.e_h + (long)b * a.e_h_bs : nullptr;
        float *po = rpart ? (a.out2 ? a.out2 + (long)b * a.out2_bs : nullptr) : (a.out ? a.out + (long)b * a.out_bs : nullptr);
        char *pc8 = rpart ? (a.out2_c8 ? a.out2_c8 + (long)b * a.out2_c8_bs : nullptr) : (a.out_c8 ? a.out_c8 + (long)b * a.out_c8_bs : nullptr);
        const int c8_ch0 = rpart ? a.out2_c8_ch0 : a.out_c8_ch0;
        const int cout_eff = a.epi == 1 ? Ch : a.Cout;      // channels of the destination tensors
        const int c8_gend = 2 * ((c8_ch0 + cout_eff + (a.epi == 0 ? a.tail_ch : 0) + 15) / 16);   // 8-channel groups the C8S destination holds
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            float bv[16];                                      // this lane's 16 biases of block m
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_w + m * 32 + 4 * kg + (r & 3) + 8 * (r >> 2);
                bv[r] = a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.0f;
            }
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int oh = h0 + wn * NF + n, ow = w0 + li;
                const bool inside = oh < a.H && ow < a.W;
                const long px = inside ? (long)oh * a.W + ow : 0;
                // The fp32 operands of the fused epilogues (gate terms, state, residual) of all four channel groups of this
                // accumulator block are fetched BEFORE the block's first store.  Interleaved with the stores, as they were, a
                // group's loads waited behind the previous group's stores (a load cannot move above a store that may alias
                // it): the epilogue was a chain of memory round trips -- counters of an epilogue-only launch: 57 % of the wave
                // cycles in s_waitcnt (tools/c8_epi_pmc.sh) -- 32 per wave, 8 now.  The registers come from the next tile's
                // prefetched fragments, which are re-read after the epilogue instead of being held across it.
                float gca[4][4], gha[4][4], gza[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = cw + m * 32 + 8 * j + 4 * kg;
                    if (a.epi != 0) f32_load4(f32_ptr(pc, cl < cout_eff ? cl : 0, px, iHW), iHW, gca[j]);
                    else gca[j][0] = gca[j][1] = gca[j][2] = gca[j][3] = 0.0f;
                    if (ph) f32_load4(f32_ptr(ph, cl, px, iHW), iHW, gha[j]);
                    else gha[j][0] = gha[j][1] = gha[j][2] = gha[j][3] = 0.0f;
                    if (pz) f32_load4(f32_ptr(pz, cl, px, iHW), iHW, gza[j]);
                    else gza[j][0] = gza[j][1] = gza[j][2] = gza[j][3] = 0.0f;
                }
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    float v[2][4];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * jp + jj;
                        const int cl = cw + m * 32 + 8 * j + 4 * kg;           // first of this lane's 4 channels (destination numbering)
                        float x[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = acc[m][n][4 * j + i] * a.out_scale + bv[4 * j + i];
                        if (a.epi == 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float y = a.relu ? dkt_relu(x[i]) : x[i];
                                if (cl + i >= a.Cout) {
                                    y = 0.0f;
                                    if (a.tail && cl + i < a.Cout + a.tail_ch && inside)
                                        y = a.tail[(long)b * a.tail_bs + (long)(cl + i - a.Cout) * iHW + px] * a.tail_ratio;
                                }
                                v[jj][i] = y;
                            }
                        } else {
                            float gc[4], gh[4], gz[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                gc[i] = gca[j][i];
                                gh[i] = gha[j][i];
                                gz[i] = gza[j][i];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (a.epi == 1) {
                                    const float g = c8_sigmoid(__fadd_rn(x[i], gc[i]));
                                    v[jj][i] = rpart ? __fmul_rn(g, gh[i]) : g;
                                } else if (a.epi == 4) {
                                    // residual join of core/extractor.py:52-60: relu(x + [relu](conv + bias)); padded channels stay 0
                                    const float t = a.relu ? dkt_relu(x[i]) : x[i];
                                    v[jj][i] = cl < a.Cout ? dkt_relu(__fadd_rn(gc[i], t)) : 0.0f;
                                } else {
                                    const float q = c8_tanh(__fadd_rn(x[i], gc[i]));
                                    v[jj][i] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, gz[i]), gh[i]), __fmul_rn(gz[i], q));
                                }
                            }
                        }
                        if (po && inside && cl < cout_eff)
                            f32_store4(const_cast<float *>(f32_ptr(po, cl, px, iHW)), iHW, v[jj], cout_eff - cl);
                    }
                    if (pc8) store_c8_pair(pc8, c8_ch0 + cw + m * 32, c8_gend, jp, oh, ow, inside, v[0], v[1]);
                }
            }
        }
    };

    // ------------------------------------------------------------------------------------------------
    // main stream
    // ------------------------------------------------------------------------------------------------
    tile_offsets(h0, w0, aoff_cur);
    if (sy.dep) {
        wait_deps(h0, w0, b);
        __builtin_amdgcn_s_barrier();
    }
    // prologue: chunk 0's patch and the weight images of steps 0..2
    issue_act(chunk_base(b, 0), aoff_cur, 0);
    const char *wptr = w_tile(co_blk);          // image of the next step to fetch
#pragma unroll
    for (int s = 0; s < C8_RING - 1; ++s) {
        issue_w(wptr, s);
        wptr += wstep;
    }
    c8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    // fragments of step 0 that the steps do not fetch themselves: Alo, rows 0 .. NF-1 (in the order the waits count).
    // PASSES == 1 keeps the A fragments double-buffered in the registers the other forms call Ahi / Alo: a chunk has nine steps,
    // so a tile's even chunks start from Ahi and its odd ones from Alo -- the chunk loop runs two chunks per trip (the host
    // refuses odd chunk counts at one pass; a run-time choice between the two step sequences sent the accumulators to scratch).
    // C8_PRIME(g_) fetches the first step's fragments of a tile from ring slot `sl` / activation buffer g_ & 1.
#define C8_PRIME(g_)                                                                                        \
    {                                                                                                       \
        const unsigned aw = lds_w_addr + sl * WSLOT, ab = lds_b_addr + ((g_) & 1) * ACT_BYTES;              \
        if constexpr (PASSES == 1) { C8_RD(Ahi[0], 0, aw) C8_RD(Ahi[1], 512, aw) }                         \
        else { C8_RD(Alo[0], 2048, aw) C8_RD(Alo[1], 2048 + 512, aw) }                                      \
        C8_RD(Bhi[0], C8_ROW(0, 0, 0), ab)                                                                  \
        if constexpr (NF > 1) C8_RD(Bhi[1], C8_ROW(1, 0, 0), ab)                                            \
        if constexpr (NF > 2) C8_RD(Bhi[2], C8_ROW(2, 0, 0), ab)                                            \
        if constexpr (NF > 3) C8_RD(Bhi[3], C8_ROW(3, 0, 0), ab)                                            \
        if constexpr (PASSES == 3) {                                                                        \
            C8_RDL(Blo[0], C8_ROW(0, 0, NPP * 16), ab)                                                      \
            if constexpr (NF > 1) C8_RDL(Blo[1], C8_ROW(1, 0, NPP * 16), ab)                                \
            if constexpr (NF > 2) C8_RDL(Blo[2], C8_ROW(2, 0, NPP * 16), ab)                                \
            if constexpr (NF > 3) C8_RDL(Blo[3], C8_ROW(3, 0, NPP * 16), ab)                                \
        }                                                                                                   \
        c8_wait_lgkm<0>();                                                                                  \
    }
    int g = 0;              // chunks consumed by this block: activation buffer parity
    int sl = 0;             // ring slot of the step being computed
    C8_PRIME(0)
    for (;;) {
        const int tn = tile + blk_count;
        const bool have_next = tn < a.total_tiles;
        int nh0 = h0, nw0 = w0, nco = co_blk, nb = b;
        if (have_next) decode(tn, nh0, nw0, nco, nb);
#define C8_CHUNK_SETUP(c)                                                                                     \
            const bool in_tile = (c) + 1 < a.nchunks;                                                             \
            /* the chunk that follows in this block's stream: this tile's next one, else the next tile's first, */ \
            /* else (the block's very last chunk) this one again, harmlessly */                                   \
            const int c_f = in_tile ? (c) + 1 : (have_next ? 0 : (c));                                            \
            const char *w_next = w_tile(have_next ? nco : co_blk);                                                \
            const int b_f = in_tile ? b : nb;                                                                     \
            if (!in_tile && have_next) {                                                                          \
                tile_offsets(nh0, nw0, aoff_nxt);                                                                 \
                wait_deps(nh0, nw0, nb);      /* (before the step barrier that precedes the next tile's first patch request) */ \
            }                                                                                                     \
            const char *act_f = chunk_base(b_f, c_f);                                                             \
            const int cur = g & 1, nxt = cur ^ 1;
            // One (chunk, tap) step.  Steps run tap COLUMN by column (dx outer, dy inner: the weight images are packed in
            // that order), so that a B fragment -- one patch row r at column dx -- serves the (up to) three steps dy = r - n:
            // step (dx, dy) multiplies row n + dy for its output row n.  Rows are re-read into their own registers when
            // they die: row 0 after step dy = 0, row 1 after dy = 1, rows 2 .. NF-1 during dy = 2 (each after the MFMAs of
            // n = r - 2), always for the NEXT column (or the next chunk's column 0); rows NF and NF+1, first needed at
            // dy = 1 / dy = 2, are fetched one step ahead (during dy = 0 / dy = 1 of their own column).  8 reads per step for
            // the 64 x 4-row wave tile instead of 12: the convolution's time follows its LDS read volume (ablation: 278 us
            // at 12 reads per step, 228 at 8, 208 at 0).  LDS reads in issue order (the waits count them):
            //   X (Alo x Bhi): Ahi[0], Ahi[1]; dy = 0: row NF (hi, lo); dy = 1: row NF+1 (hi, lo)
            //   Y (Ahi x Bhi): Alo'[0], Alo'[1], then the hi halves of the dying rows
            //   Z (Ahi x Blo): the lo halves of the dying rows
#define C8_STEP(T)                                                                                                     \
    {                                                                                                                  \
        constexpr int DX = (T) / 3, DY = (T) % 3, NDX = (DX + 1) % 3;                                                  \
        const int sl1 = sl + 1 == C8_RING ? 0 : sl + 1, sl2 = sl == 0 ? C8_RING - 1 : sl - 1;                         \
        const unsigned adw_s = lds_w_addr + sl * WSLOT;                                                                \
        const unsigned adw_n = lds_w_addr + sl1 * WSLOT;                                                               \
        const unsigned adb_c = lds_b_addr + cur * ACT_BYTES;                     /* this column's patch */             \
        const unsigned adb_n = lds_b_addr + (DX < 2 ? cur : nxt) * ACT_BYTES;    /* the next column's   */             \
        /* the weights of step s+1 (issued RING-2 steps ago) and, from step 6 on, the following chunk's patch      */ \
        /* have landed once at most the pieces issued after them are outstanding                                   */ \
        if ((T) >= 1 && (T) <= C8_RING - 3) c8_wait_vm<(C8_RING - 3) * WPI + NIA>();                                   \
        else c8_wait_vm<(C8_RING - 3) * WPI>();                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                  \
        /* Alo and this step's hi rows are in; what the previous step read after them may still fly */                \
        if constexpr (DY == 0) c8_wait_lgkm<(NF > 2 ? NF - 2 : 0)>();                                                  \
        else if constexpr (DY == 1) c8_wait_lgkm<2>();                                                                 \
        else c8_wait_lgkm<(NF > 1 ? 2 : 0)>();                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        /* ---- X */                                                                                                   \
        C8_MM(Alo, 0, Bhi, DY, 0) C8_RD(Ahi[0], 0, adw_s)                                                                 \
        C8_MM(Alo, 1, Bhi, DY, 0) C8_RD(Ahi[1], 512, adw_s)                                                               \
        if constexpr (NF > 1) { C8_MM(Alo, 0, Bhi, 1 + DY, 1)                                                             \
            if constexpr (DY < 2) C8_RD(Bhi[NF + DY], C8_ROW(NF + DY, DX, 0), adb_c)                                   \
            C8_MM(Alo, 1, Bhi, 1 + DY, 1)                                                                                 \
            if constexpr (DY < 2) C8_RDL(Blo[NF + DY], C8_ROW(NF + DY, DX, NPP * 16), adb_c) }                         \
        else if constexpr (DY < 2) { C8_RD(Bhi[NF + DY], C8_ROW(NF + DY, DX, 0), adb_c)                                \
            C8_RDL(Blo[NF + DY], C8_ROW(NF + DY, DX, NPP * 16), adb_c) }                                               \
        if constexpr (NF > 2) { C8_MM(Alo, 0, Bhi, 2 + DY, 2) C8_MM(Alo, 1, Bhi, 2 + DY, 2) }                                \
        if constexpr (NF > 3) { C8_MM(Alo, 0, Bhi, 3 + DY, 3) C8_MM(Alo, 1, Bhi, 3 + DY, 3) }                                \
        /* ---- DMA issue: the image of step s+RING-1 into the slot of step s-1 (all of its reads were consumed       */ \
        /* before this step's barrier); at the chunk's first step also the following chunk's patch                   */ \
        if ((T) == 10 - C8_RING && !in_tile) wptr = w_next; /* the stream continues with the next tile's first image */ \
        if ((T) == 0) { /* (before the weights: the patch must be older than every image the waits let fly) */        \
            if (in_tile || !have_next) issue_act(act_f, aoff_cur, nxt);                                                \
            else issue_act(act_f, aoff_nxt, nxt);                                                                      \
        }                                                                                                              \
        issue_w(wptr, sl2);                                                                                            \
        wptr += wstep;                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        c8_wait_lgkm<(DY < 2 ? 2 : 0)>(); /* Ahi is in (the row read ahead may still fly) */                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        /* ---- Y */                                                                                                   \
        C8_MM(Ahi, 0, Bhi, DY, 0) C8_RD(Alo[0], 2048, adw_n)                                                              \
        C8_MM(Ahi, 1, Bhi, DY, 0) C8_RD(Alo[1], 2048 + 512, adw_n)                                                        \
        if constexpr (DY == 0) C8_RD(Bhi[0], C8_ROW(0, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 1 && NF > 1) C8_RD(Bhi[1], C8_ROW(1, NDX, 0), adb_n)                                       \
        if constexpr (DY == 2 && NF > 2) C8_RD(Bhi[2], C8_ROW(2, NDX, 0), adb_n)                                       \
        if constexpr (NF > 1) { C8_MM(Ahi, 0, Bhi, 1 + DY, 1) C8_MM(Ahi, 1, Bhi, 1 + DY, 1)                                  \
            if constexpr (DY == 2 && NF > 3) C8_RD(Bhi[3], C8_ROW(3, NDX, 0), adb_n) }                                 \
        if constexpr (NF > 2) { C8_MM(Ahi, 0, Bhi, 2 + DY, 2) C8_MM(Ahi, 1, Bhi, 2 + DY, 2) }                                \
        if constexpr (NF > 3) { C8_MM(Ahi, 0, Bhi, 3 + DY, 3) C8_MM(Ahi, 1, Bhi, 3 + DY, 3) }                                \
        /* ---- Z (every lo row of this step was read before this step's X reads: in since the wait before Y) */       \
        C8_MM(Ahi, 0, Blo, DY, 0) C8_MM(Ahi, 1, Blo, DY, 0)                                                                  \
        if constexpr (DY == 0) C8_RDL(Blo[0], C8_ROW(0, NDX, NPP * 16), adb_n)                                         \
        if constexpr (DY == 1 && NF > 1) C8_RDL(Blo[1], C8_ROW(1, NDX, NPP * 16), adb_n)                               \
        if constexpr (DY == 2 && NF > 2) C8_RDL(Blo[2], C8_ROW(2, NDX, NPP * 16), adb_n)                               \
        if constexpr (NF > 1) { C8_MM(Ahi, 0, Blo, 1 + DY, 1) C8_MM(Ahi, 1, Blo, 1 + DY, 1)                                  \
            if constexpr (DY == 2 && NF > 3) C8_RDL(Blo[3], C8_ROW(3, NDX, NPP * 16), adb_n) }                         \
        if constexpr (NF > 2) { C8_MM(Ahi, 0, Blo, 2 + DY, 2) C8_MM(Ahi, 1, Blo, 2 + DY, 2) }                                \
        if constexpr (NF > 3) { C8_MM(Ahi, 0, Blo, 3 + DY, 3) C8_MM(Ahi, 1, Blo, 3 + DY, 3) }                                \
        sl = sl1;                                                                                                      \
    }
            // ---- the step's shared head (waits for this step's DMA, barrier) and DMA issue, as in C8_STEP
#define C8_STEP_HEAD(T)                                                                                                \
        constexpr int DX = (T) / 3, DY = (T) % 3, NDX = (DX + 1) % 3;                                                  \
        const int sl1 = sl + 1 == C8_RING ? 0 : sl + 1, sl2 = sl == 0 ? C8_RING - 1 : sl - 1;                         \
        const unsigned adw_s = lds_w_addr + sl * WSLOT;                                                                \
        const unsigned adw_n = lds_w_addr + sl1 * WSLOT;                                                               \
        const unsigned adb_c = lds_b_addr + cur * ACT_BYTES;                                                           \
        const unsigned adb_n = lds_b_addr + (DX < 2 ? cur : nxt) * ACT_BYTES;                                          \
        (void)adw_s; (void)adw_n; (void)adb_c; (void)adb_n;                                                            \
        if ((T) >= 1 && (T) <= C8_RING - 3) c8_wait_vm<(C8_RING - 3) * WPI + NIA>();                                   \
        else c8_wait_vm<(C8_RING - 3) * WPI>();                                                                        \
        __builtin_amdgcn_s_barrier();
#define C8_STEP_DMA(T)                                                                                                 \
        if ((T) == 10 - C8_RING && !in_tile) wptr = w_next;                                                            \
        if ((T) == 0) {                                                                                                \
            if (in_tile || !have_next) issue_act(act_f, aoff_cur, nxt);                                                \
            else issue_act(act_f, aoff_nxt, nxt);                                                                      \
        }                                                                                                              \
        issue_w(wptr, sl2);                                                                                            \
        wptr += wstep;                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);
            // PASSES == 2: C8_STEP without its Z pass and without any lo row.  LDS reads in issue order:
            //   X (Alo x Bhi): Ahi[0], Ahi[1]; dy = 0: row NF; dy = 1: row NF+1
            //   Y (Ahi x Bhi): Alo'[0], Alo'[1], then the dying rows (for the next column)
            // so a step starts with its Alo and rows in once at most the previous step's dying-row reads fly (1 after dy = 0,
            // 1 after dy = 1 when NF > 1, none after dy = 2: the rows re-read there are this step's)
#define C8_STEP2(T)                                                                                                    \
    {                                                                                                                  \
        C8_STEP_HEAD(T)                                                                                                \
        if constexpr (DY == 1) c8_wait_lgkm<1>();                                                                      \
        else if constexpr (DY == 2) c8_wait_lgkm<(NF > 1 ? 1 : 0)>();                                                  \
        else c8_wait_lgkm<0>();                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        C8_MM(Alo, 0, Bhi, DY, 0) C8_RD(Ahi[0], 0, adw_s)                                                              \
        C8_MM(Alo, 1, Bhi, DY, 0) C8_RD(Ahi[1], 512, adw_s)                                                            \
        if constexpr (NF > 1) { C8_MM(Alo, 0, Bhi, 1 + DY, 1)                                                          \
            if constexpr (DY < 2) C8_RD(Bhi[NF + DY], C8_ROW(NF + DY, DX, 0), adb_c)                                   \
            C8_MM(Alo, 1, Bhi, 1 + DY, 1) }                                                                            \
        else if constexpr (DY < 2) C8_RD(Bhi[NF + DY], C8_ROW(NF + DY, DX, 0), adb_c)                                  \
        if constexpr (NF > 2) { C8_MM(Alo, 0, Bhi, 2 + DY, 2) C8_MM(Alo, 1, Bhi, 2 + DY, 2) }                          \
        if constexpr (NF > 3) { C8_MM(Alo, 0, Bhi, 3 + DY, 3) C8_MM(Alo, 1, Bhi, 3 + DY, 3) }                          \
        C8_STEP_DMA(T)                                                                                                 \
        c8_wait_lgkm<(DY < 2 ? 1 : 0)>(); /* Ahi is in (the row read ahead may still fly) */                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        C8_MM(Ahi, 0, Bhi, DY, 0) C8_RD(Alo[0], 2048, adw_n)                                                           \
        C8_MM(Ahi, 1, Bhi, DY, 0) C8_RD(Alo[1], 2048 + 512, adw_n)                                                     \
        if constexpr (DY == 0) C8_RD(Bhi[0], C8_ROW(0, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 1 && NF > 1) C8_RD(Bhi[1], C8_ROW(1, NDX, 0), adb_n)                                       \
        if constexpr (DY == 2 && NF > 2) C8_RD(Bhi[2], C8_ROW(2, NDX, 0), adb_n)                                       \
        if constexpr (NF > 1) { C8_MM(Ahi, 0, Bhi, 1 + DY, 1) C8_MM(Ahi, 1, Bhi, 1 + DY, 1)                            \
            if constexpr (DY == 2 && NF > 3) C8_RD(Bhi[3], C8_ROW(3, NDX, 0), adb_n) }                                 \
        if constexpr (NF > 2) { C8_MM(Ahi, 0, Bhi, 2 + DY, 2) C8_MM(Ahi, 1, Bhi, 2 + DY, 2) }                          \
        if constexpr (NF > 3) { C8_MM(Ahi, 0, Bhi, 3 + DY, 3) C8_MM(Ahi, 1, Bhi, 3 + DY, 3) }                          \
        sl = sl1;                                                                                                      \
    }
            // PASSES == 1: one product per block, A fragments double-buffered (PA: this step's, PB: the next step's, fetched first
            // thing behind the barrier), MFMAs row by row so that a dying row is re-read as soon as its last product has issued.
            // LDS reads in issue order: PB[0], PB[1]; dy < 2: row NF + dy; then the dying rows.  Waits as in C8_STEP2.
#define C8_STEP1(T, PA, PB)                                                                                            \
    {                                                                                                                  \
        C8_STEP_HEAD(T)                                                                                                \
        if constexpr (DY == 1) c8_wait_lgkm<1>();                                                                      \
        else if constexpr (DY == 2) c8_wait_lgkm<(NF > 1 ? 1 : 0)>();                                                  \
        else c8_wait_lgkm<0>();                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        C8_RD(PB[0], 0, adw_n) C8_RD(PB[1], 512, adw_n)                                                                \
        if constexpr (DY < 2) C8_RD(Bhi[NF + DY], C8_ROW(NF + DY, DX, 0), adb_c)                                       \
        C8_MM(PA, 0, Bhi, DY, 0) C8_MM(PA, 1, Bhi, DY, 0)                                                              \
        if constexpr (DY == 0) C8_RD(Bhi[0], C8_ROW(0, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 1 && NF > 1) C8_RD(Bhi[1], C8_ROW(1, NDX, 0), adb_n)                                       \
        if constexpr (DY == 2 && NF > 2) C8_RD(Bhi[2], C8_ROW(2, NDX, 0), adb_n)                                       \
        C8_STEP_DMA(T)                                                                                                 \
        if constexpr (NF > 1) { C8_MM(PA, 0, Bhi, 1 + DY, 1) C8_MM(PA, 1, Bhi, 1 + DY, 1)                              \
            if constexpr (DY == 2 && NF > 3) C8_RD(Bhi[3], C8_ROW(3, NDX, 0), adb_n) }                                 \
        if constexpr (NF > 2) { C8_MM(PA, 0, Bhi, 2 + DY, 2) C8_MM(PA, 1, Bhi, 2 + DY, 2) }                            \
        if constexpr (NF > 3) { C8_MM(PA, 0, Bhi, 3 + DY, 3) C8_MM(PA, 1, Bhi, 3 + DY, 3) }                            \
        sl = sl1;                                                                                                      \
    }
#define C8_CHUNK1(PA, PB)                                                                                     \
            C8_STEP1(0, PA, PB) C8_STEP1(1, PB, PA) C8_STEP1(2, PA, PB) C8_STEP1(3, PB, PA) C8_STEP1(4, PA, PB)   \
            C8_STEP1(5, PB, PA) C8_STEP1(6, PA, PB) C8_STEP1(7, PB, PA) C8_STEP1(8, PA, PB)
        if constexpr (PASSES == 1) {
            for (int c = 0; c < a.nchunks; c += 2) {
                { C8_CHUNK_SETUP(c) C8_CHUNK1(Ahi, Alo) }
                ++g;
                { C8_CHUNK_SETUP(c + 1) C8_CHUNK1(Alo, Ahi) }
                ++g;
            }
        } else {
            for (int c = 0; c < a.nchunks; ++c, ++g) {
                C8_CHUNK_SETUP(c)
                if constexpr (PASSES == 3) {
                    C8_STEP(0) C8_STEP(1) C8_STEP(2) C8_STEP(3) C8_STEP(4) C8_STEP(5) C8_STEP(6) C8_STEP(7) C8_STEP(8)
                } else {
                    C8_STEP2(0) C8_STEP2(1) C8_STEP2(2) C8_STEP2(3) C8_STEP2(4) C8_STEP2(5) C8_STEP2(6) C8_STEP2(7) C8_STEP2(8)
                }
            }
        }
        c8_wait_lgkm<0>();      // the prefetched fragments of the next tile have landed: their registers are stable
        epilogue();
        publish(tile);
        if (!have_next) break;
        tile = tn; h0 = nh0; w0 = nw0; co_blk = nco; b = nb;
#pragma unroll
        for (int j = 0; j < NIA; ++j) aoff_cur[j] = aoff_nxt[j];
        zero_acc();
        // The fragments the last step fetched for this tile's first step are fetched AGAIN here (their LDS images are
        // untouched: no DMA is issued during the epilogue): that makes the 16 + 8 NF fragment registers dead across the
        // epilogue, which needs them for its operand batches (see epilogue()).
        C8_PRIME(g)
    }
    c8_wait_vm<0>();        // no DMA may land in this block's LDS after it has been released
}

template <int WM, int WN, int NF, int RING, int PASSES = 3>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 4 ? 1 : 2)) void conv_c8_kernel(C8ArgsPair ap, int nb0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    conv_c8_body<WM, WN, NF, RING, PASSES>(ap, nb0, (int)gridDim.x, (int)blockIdx.x, lds, C8Sync{});
}

// Two dependent launches as ONE (round 5 prototype): stage 0 (shape 0, one or two problems) and stage 1 (shape 1, one
// problem that reads what stage 0 writes), both of 4-wave blocks.  Every block walks its stage-0 tiles, then its stage-1 tiles.
template <int WN0, int NF0, int RING0, int WN1, int NF1, int RING1, int PASSES>
__global__ __launch_bounds__(256, 2) void conv_c8_chain_kernel(C8ArgsPair s0, int nb0_0, C8ArgsPair s1, C8Sync sy) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the flag words count launches: every word holds the same value before a launch, and a block's own first stage-0 tile is
    // written by nobody else -- its old value + 1 is what this launch publishes everywhere
    if (sy.pub) {
        typedef __attribute__((address_space(1))) unsigned gu32;
        const long own = (int)blockIdx.x < nb0_0 ? (long)blockIdx.x : (long)sy.dep_tiles_per_prob + ((long)blockIdx.x - nb0_0);
        sy.target = __builtin_amdgcn_readfirstlane(__hip_atomic_load((gu32 *)sy.pub + own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u);
    }
    C8Sync p = sy;
    p.dep = nullptr;                                   // stage 0 publishes, waits for nothing
    conv_c8_body<1, WN0, NF0, RING0, PASSES>(s0, nb0_0, (int)gridDim.x, (int)blockIdx.x, lds, p);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // the LDS changes hands
    C8Sync c = sy;
    c.pub = nullptr;                                   // stage 1 waits, publishes nothing
    conv_c8_body<1, WN1, NF1, RING1, PASSES>(s1, (int)gridDim.x, (int)gridDim.x, (int)blockIdx.x, lds, c);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int c8_round(int x, int m) { return (x + m - 1) / m * m; }

extern "C" int dkt_act_c8_dims(int H, int W, int *Hp, int *Wp) {
    if (H <= 0 || W <= 0 || !Hp || !Wp) return DKT_E_SHAPE;
    *Hp = c8_round(H, 8) + 2;
    *Wp = c8_round(W, 32) + 2;
    return DKT_OK;
}

// fp32 NCHW -> C8S (interior only: the border and the padding channels of the destination must be zero already)
__global__ __launch_bounds__(256) void act_c8_pack_kernel(const float *x, long x_bs, char *dst, long dst_bs, int C, int H, int W,
                                                          int Wp, long plane, int ch0, float scale) {
    const int g = blockIdx.y, b = blockIdx.z;                 // 8-channel group of the source
    const long HW = (long)H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const int oh = (int)(i / W), ow = (int)(i - (long)oh * W);
        unsigned hw[4], lw[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int c0 = g * 8 + 2 * d;
            const float x0 = c0 < C ? x[(long)b * x_bs + (long)c0 * HW + i] * scale : 0.0f;
            const float x1 = c0 + 1 < C ? x[(long)b * x_bs + (long)(c0 + 1) * HW + i] * scale : 0.0f;
            const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1;
            hw[d] = c8_pack_h2(a0, a1);
            lw[d] = c8_pack_h2((_Float16)(x0 - (float)a0), (_Float16)(x1 - (float)a1));
        }
        char *p = dst + (long)b * dst_bs + (long)((ch0 >> 3) + g) * 2 * plane + ((long)(oh + 1) * Wp + (ow + 1)) * 16;
        *(uint4 *)p = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *(uint4 *)(p + plane) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

__global__ __launch_bounds__(256) void act_c8_unpack_kernel(const char *src, long src_bs, float *y, long y_bs, int C, int H, int W,
                                                            int Wp, long plane, int ch0, float inv_scale) {
    const int g = blockIdx.y, b = blockIdx.z;
    const long HW = (long)H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const int oh = (int)(i / W), ow = (int)(i - (long)oh * W);
        const char *p = src + (long)b * src_bs + (long)((ch0 >> 3) + g) * 2 * plane + ((long)(oh + 1) * Wp + (ow + 1)) * 16;
        const f16x8 hi = *(const f16x8 *)p, lo = *(const f16x8 *)(p + plane);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (g * 8 + k < C) y[(long)b * y_bs + (long)(g * 8 + k) * HW + i] = ((float)hi[k] + (float)lo[k]) * inv_scale;
    }
}

extern "C" int dkt_act_c8_pack(const float *x, long x_bstride, void *dst, long dst_bstride_bytes, int B, int C, int H, int W,
                               int ch0, float scale, int device, void *stream) {
    if (!x || !dst) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || (ch0 & 7)) return DKT_E_SHAPE;
    int Hp, Wp;
    dkt_act_c8_dims(H, W, &Hp, &Wp);
    DKT_ENTER(device);
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 255) / 256 > 1024 ? 1024 : (HW + 255) / 256), (unsigned)((C + 7) / 8), (unsigned)B);
    hipLaunchKernelGGL(act_c8_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_bstride, (char *)dst, dst_bstride_bytes,
                       C, H, W, Wp, (long)Hp * Wp * 16, ch0, scale);
    return dkt_launch_status();
}

extern "C" int dkt_act_c8_unpack(const void *src, long src_bstride_bytes, float *y, long y_bstride, int B, int C, int H, int W,
                                 int ch0, float scale, int device, void *stream) {
    if (!src || !y) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || (ch0 & 7) || !(scale > 0.0f)) return DKT_E_SHAPE;
    int Hp, Wp;
    dkt_act_c8_dims(H, W, &Hp, &Wp);
    DKT_ENTER(device);
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 255) / 256 > 1024 ? 1024 : (HW + 255) / 256), (unsigned)((C + 7) / 8), (unsigned)B);
    hipLaunchKernelGGL(act_c8_unpack_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const char *)src, src_bstride_bytes, y,
                       y_bstride, C, H, W, Wp, (long)Hp * Wp * 16, ch0, 1.0f / scale);
    return dkt_launch_status();
}

// ---- weights: (Cout, Cin, 3, 3) fp32 -> [chunk][tap][co64][hi|lo][k8][64 co][8] fp16, sources padded to 16 channels
struct C8PackArgs {
    const float *w;
    _Float16 *dst;
    int Cout, Cin, n_co64, nchunks;
    int src_ch[C8_MAX_SRC];
    int nsrc;
    float scale;
};

__global__ __launch_bounds__(256) void conv_c8_pack_kernel(C8PackArgs a) {
    const long total = (long)a.nchunks * 9 * a.n_co64 * 2048;       // halfs
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i & 7);
    const int co_in = (int)((i >> 3) & 63);
    const int k8 = (int)((i >> 9) & 1);
    const int hl = (int)((i >> 10) & 1);
    long r = i >> 11;
    const int c64 = (int)(r % a.n_co64); r /= a.n_co64;
    const int tp = (int)(r % 9);
    const int tap = (tp % 3) * 3 + tp / 3;      // step order is dx-major: image tp holds tap (dy = tp % 3, dx = tp / 3)
    const int chunk = (int)(r / 9);
    int pc = chunk * 16 + k8 * 8 + k, s = 0, cbase = 0;
    while (s + 1 < a.nsrc && pc >= ((a.src_ch[s] + 15) & ~15)) {
        pc -= (a.src_ch[s] + 15) & ~15;
        cbase += a.src_ch[s];
        ++s;
    }
    const int co = c64 * 64 + co_in;
    float v = 0.0f;
    if (co < a.Cout && pc < a.src_ch[s]) v = a.w[((long)co * a.Cin + cbase + pc) * 9 + tap] * a.scale;
    const _Float16 hi = (_Float16)v;
    a.dst[i] = hl ? (_Float16)(v - (float)hi) : hi;
}

static int c8_chunks(const int *src_ch, int nsrc) {
    int t = 0;
    for (int s = 0; s < nsrc; ++s) t += (src_ch[s] + 15) / 16;
    return t;
}

extern "C" long dkt_conv_c8_packed_bytes(const int *src_channels, int nsrc, int Cout) {
    if (!src_channels || nsrc < 1 || nsrc > C8_MAX_SRC || Cout <= 0) return DKT_E_SHAPE;
    // + 12 KB of slack: a block always copies its tile shape's full 4 x 64 channel step image, also where the layer has fewer
    // channel blocks (the surplus waves idle on whatever follows)
    return (long)c8_chunks(src_channels, nsrc) * 9 * ((Cout + 63) / 64) * 4096 + 3 * 4096;
}

extern "C" int dkt_conv_c8_pack_weights(const float *w, const int *src_channels, int nsrc, int Cout, float scale,
                                        void *packed, int device, void *stream) {
    if (!w || !src_channels || !packed) return DKT_E_NULL;
    if (nsrc < 1 || nsrc > C8_MAX_SRC || Cout <= 0 || !(scale > 0.0f)) return DKT_E_SHAPE;
    C8PackArgs a;
    a.w = w; a.dst = (_Float16 *)packed; a.Cout = Cout; a.Cin = 0; a.nsrc = nsrc; a.scale = scale;
    for (int s = 0; s < C8_MAX_SRC; ++s) {
        a.src_ch[s] = s < nsrc ? src_channels[s] : 0;
        if (s < nsrc && src_channels[s] <= 0) return DKT_E_SHAPE;
        a.Cin += a.src_ch[s];
    }
    a.n_co64 = (Cout + 63) / 64;
    a.nchunks = c8_chunks(src_channels, nsrc);
    DKT_ENTER(device);
    const long total = (long)a.nchunks * 9 * a.n_co64 * 2048;
    hipLaunchKernelGGL(conv_c8_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}

// ---- launch
template <int WM, int WN, int NF, int RING, int PASSES = 3>
static int c8_launch(C8Args a, int B, hipStream_t st, const C8Args *second, int B2) {
    constexpr int NW = WM * WN, TR = WN * NF;
    constexpr int NU = (TR + 2) * C8_PC * 4;
    constexpr int NPR = (NU + 63) / 64, NIA = (NPR + NW - 1) / NW;
    size_t lds = (size_t)2 * (NIA * NW > NPR ? NPR + 1 : NPR) * 1024 + (size_t)RING * WM * 4096;
    const bool head = a.epi == 3 || (second && second->epi == 3);
    if (head) {
        if (NW != 8) return DKT_E_UNSUPPORTED;                // one block per CU: the extra LDS costs no residency
        lds += (size_t)NW * NF * 9 * 32 * 4;
    }
    auto kern = conv_c8_kernel<WM, WN, NF, RING, PASSES>;
    static int slots[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!slots[dev & 63]) {
        size_t want = lds + (NW == 8 && !head ? (size_t)NW * NF * 9 * 32 * 4 : 0);     // (room for a later head launch of this shape)
        if (want > 160 * 1024) want = lds;
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != hipSuccess) return (int)e;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        slots[dev & 63] = per_cu * cus;
    }
    auto shape = [&](C8Args &x, int nb) -> long {
        x.tiles_w = (x.W + 31) / 32;
        x.tiles_xy = x.tiles_w * ((x.H + TR - 1) / TR);
        x.n_co = (x.n_co64 + WM - 1) / WM;
        const long total = (long)x.tiles_xy * x.n_co * nb;
        x.total_tiles = (int)total;
        return total;
    };
    const long cap = slots[dev & 63];
    C8ArgsPair ap;
    const long total0 = shape(a, B);
    if (total0 > 0x7fffffffL) return DKT_E_SHAPE;
    ap.p[0] = a;
    ap.p[1] = a;
    long nb0 = total0 > cap ? cap : total0, nb1 = 0;
    if (second) {
        C8Args b = *second;
        const long total1 = shape(b, B2);
        if (total1 > 0x7fffffffL) return DKT_E_SHAPE;
        ap.p[1] = b;
        nb0 = total0; nb1 = total1;
        if (total0 + total1 > cap) {
            const double w0 = (double)total0 * a.nchunks, w1 = (double)total1 * b.nchunks;
            nb1 = (long)(cap * w1 / (w0 + w1) + 0.5);
            nb1 = nb1 < 1 ? 1 : (nb1 > total1 ? total1 : nb1);
            nb0 = cap - nb1;
            if (nb0 > total0) nb0 = total0;
            if (nb0 < 1) nb0 = 1;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nb0 + nb1)), dim3(64 * NW), lds, st, ap, (int)nb0);
    return dkt_launch_status();
}

static int c8_fill(C8Args &a, const dkt_conv_c8_desc *d) {
    if (!d) return DKT_E_NULL;
    if (d->nsrc < 1 || d->nsrc > C8_MAX_SRC || d->B <= 0 || d->B > 65535 || d->H <= 0 || d->W <= 0 || d->Cout <= 0) return DKT_E_SHAPE;
    if (!d->w || (!d->out && !d->out_c8 && d->epilogue != 1 && d->epilogue != 3)) return DKT_E_NULL;
    if (!(d->out_scale > 0.0f) || !(d->act_scale > 0.0f)) return DKT_E_SHAPE;
    if (d->epilogue < 0 || d->epilogue > 4) return DKT_E_UNSUPPORTED;
    int Hp, Wp;
    dkt_act_c8_dims(d->H, d->W, &Hp, &Wp);
    a.nchunks = 0;
    for (int s = 0; s < C8_MAX_SRC; ++s) {
        a.src[s] = s < d->nsrc ? (const char *)d->src[s] : nullptr;
        a.src_bs[s] = s < d->nsrc ? d->src_bstride[s] : 0;
        a.src_n16[s] = s < d->nsrc ? (d->src_channels[s] + 15) / 16 : 0;
        if (s < d->nsrc && (!d->src[s] || d->src_channels[s] <= 0)) return DKT_E_NULL;
        a.nchunks += a.src_n16[s];
    }
    a.nsrc = d->nsrc;
    a.plane_bytes = (long)Hp * Wp * 16;
    a.Wp = Wp;
    a.w = (const char *)d->w;
    a.n_co64 = (d->Cout + 63) / 64;
    a.bias = d->bias;
    a.out_scale = d->out_scale;
    a.H = d->H; a.W = d->W; a.Cout = d->Cout;
    a.relu = d->relu ? 1 : 0;
    a.epi = d->epilogue;
    a.out = d->out; a.out_bs = d->out_bstride;
    a.out_c8 = (char *)d->out_c8; a.out_c8_bs = d->out_c8_bstride; a.out_c8_plane = a.plane_bytes; a.out_c8_Wp = Wp;
    a.out_c8_ch0 = d->out_c8_ch0;
    a.act_scale = d->act_scale;
    a.e_c0 = d->e0; a.e_c1 = d->e1; a.e_h = d->h;
    a.e_c0_bs = d->e0_bstride; a.e_c1_bs = d->e1_bstride; a.e_h_bs = d->h_bstride;
    a.out2 = d->out2; a.out2_bs = d->out2_bstride;
    a.out2_c8 = (char *)d->out2_c8; a.out2_c8_bs = d->out2_c8_bstride; a.out2_c8_ch0 = d->out2_c8_ch0;
    a.f32_c4 = d->f32_c4 ? 1 : 0;
    a.head_w = d->head_w; a.head_out = d->head_out; a.head_out_bs = d->head_out_bstride; a.head_nout = d->head_outputs;
    if (a.epi == 3 && (!a.head_w || !a.head_out || !a.bias)) return DKT_E_NULL;
    if (a.epi == 3 && (a.head_nout != 1 || a.Cout % 64 != 0)) return DKT_E_UNSUPPORTED;       // one output (stereo heads); the 2-output flow head keeps the hidden tensor
    a.tail = d->tail; a.tail_bs = d->tail_bstride; a.tail_ch = d->tail ? d->tail_channels : 0;
    a.tail_ratio = d->tail_scale > 0.0f ? d->tail_scale / d->act_scale : 1.0f;
    if ((a.out_c8_ch0 & 7) || (a.out2_c8_ch0 & 7)) return DKT_E_SHAPE;
    if (a.epi == 1) {
        if (!d->e0 || !d->e1 || !d->h || (!d->out2 && !d->out2_c8) || !d->out) return DKT_E_NULL;
        if (d->Cout % 128 != 0) return DKT_E_UNSUPPORTED;
    } else if (a.epi == 2) {
        if (!d->e0 || !d->e1 || !d->h) return DKT_E_NULL;
        if (d->Cout % 64 != 0) return DKT_E_UNSUPPORTED;
    } else if (a.epi == 4) {
        if (!d->e0) return DKT_E_NULL;
        if (d->Cout % 4 != 0) return DKT_E_UNSUPPORTED;
    }
    if (a.tail && (a.epi != 0 || !a.out_c8 || a.tail_ch <= 0 || a.Cout + a.tail_ch > a.n_co64 * 64)) return DKT_E_UNSUPPORTED;
    a.tiles_w = a.tiles_xy = a.n_co = a.total_tiles = 0;
    return DKT_OK;
}

// cfg: 0 = by shape; 1: 256 co x 8 rows (8 waves); 2: 128 co x 8 rows (8 waves); 3: 64 co x 8 rows (4 waves, two blocks per CU);
//      4: 64 co x 4 rows (4 waves); 5: 256 co x 4 rows (4 waves, two blocks per CU); 6: 128 co x 8 rows (4 waves, two blocks per CU)
template <int PASSES>
static int c8_dispatch_p(const C8Args &a, int B, int cfg, hipStream_t st, const C8Args *second, int B2) {
    switch (cfg) {
    case 1: return c8_launch<4, 2, 4, 4, PASSES>(a, B, st, second, B2);
    case 2: return c8_launch<2, 4, 2, 4, PASSES>(a, B, st, second, B2);
    case 3: return c8_launch<1, 4, 2, 6, PASSES>(a, B, st, second, B2);
    case 4: return c8_launch<1, 4, 1, 8, PASSES>(a, B, st, second, B2);
    default: break;
    }
    if constexpr (PASSES == 3) {          // (the 4-wave forms kept for the tests exist at full precision only)
        if (cfg == 5) return c8_launch<4, 1, 4, 3>(a, B, st, second, B2);
        if (cfg == 6) return c8_launch<2, 2, 4, 4>(a, B, st, second, B2);
    }
    return DKT_E_UNSUPPORTED;
}

static int c8_dispatch(const C8Args &a, int B, int cfg, int passes, hipStream_t st, const C8Args *second, int B2) {
    if (cfg == 0) {
        const long tiles8 = (long)((a.W + 31) / 32) * ((a.H + 7) / 8) * B;
        if (a.n_co64 >= 4 && tiles8 >= 128) cfg = 1;
        else if (a.n_co64 >= 2 && tiles8 * ((a.n_co64 + 1) / 2) >= 128) cfg = 2;
        else cfg = tiles8 * a.n_co64 >= 200 ? 3 : 4;
    }
    // (one pass: the chunk loop runs two chunks per trip -- odd chunk counts take two passes or more)
    if (passes == 1 && ((a.nchunks & 1) || (second && (second->nchunks & 1)))) return DKT_E_UNSUPPORTED;
    switch (passes) {
    case 0: case 3: return c8_dispatch_p<3>(a, B, cfg, st, second, B2);
    case 2: return c8_dispatch_p<2>(a, B, cfg, st, second, B2);
    case 1: return c8_dispatch_p<1>(a, B, cfg, st, second, B2);
    default: return DKT_E_UNSUPPORTED;
    }
}

extern "C" int dkt_conv2d_c8(const dkt_conv_c8_desc *d, int cfg, int device, void *stream) {
    C8Args a;
    const int rc = c8_fill(a, d);
    if (rc != DKT_OK) return rc;
    DKT_ENTER(device);
    return c8_dispatch(a, d->B, cfg, d->passes, (hipStream_t)stream, nullptr, 0);
}

extern "C" int dkt_conv2d_c8_pair(const dkt_conv_c8_desc *d0, const dkt_conv_c8_desc *d1, int cfg, int device, void *stream) {
    C8Args a, b;
    int rc = c8_fill(a, d0);
    if (rc != DKT_OK) return rc;
    rc = c8_fill(b, d1);
    if (rc != DKT_OK) return rc;
    if (cfg == 0) return DKT_E_UNSUPPORTED;          // both problems run one instantiation: the caller names it
    if ((d0->passes ? d0->passes : 3) != (d1->passes ? d1->passes : 3)) return DKT_E_UNSUPPORTED;
    DKT_ENTER(device);
    return c8_dispatch(a, d0->B, cfg, d0->passes, (hipStream_t)stream, &b, d1->B);
}

// ---- chain launch: stage 0 (d0a [, d0b]) -> stage 1 (d1), one launch, a flag round instead of the kernel boundary
template <int WN0, int NF0, int RING0, int WN1, int NF1, int RING1, int PASSES>
static int c8_chain_launch(C8Args a0, const C8Args *b0, int B, C8Args a1, C8Sync sy, int max_blocks, hipStream_t st) {
    constexpr int TR0 = WN0 * NF0, TR1 = WN1 * NF1;
    auto lds_of = [](int TR, int RING) -> size_t {
        const int NU = (TR + 2) * C8_PC * 4, NPR = (NU + 63) / 64, NIA = (NPR + 3) / 4;
        return (size_t)2 * (NIA * 4 > NPR ? NPR + 1 : NPR) * 1024 + (size_t)RING * 4096;
    };
    const size_t l0 = lds_of(TR0, RING0), l1 = lds_of(TR1, RING1), lds = l0 > l1 ? l0 : l1;
    auto kern = conv_c8_chain_kernel<WN0, NF0, RING0, WN1, NF1, RING1, PASSES>;
    static int slots[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!slots[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) return DKT_E_UNSUPPORTED;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return DKT_E_UNSUPPORTED;
        slots[dev & 63] = per_cu * cus;
    }
    auto shape = [&](C8Args &x, int TR) -> long {
        x.tiles_w = (x.W + 31) / 32;
        x.tiles_xy = x.tiles_w * ((x.H + TR - 1) / TR);
        x.n_co = x.n_co64;                              // WM = 1
        const long total = (long)x.tiles_xy * x.n_co * B;
        x.total_tiles = (int)total;
        return total;
    };
    C8ArgsPair s0, s1;
    const long t0a = shape(a0, TR0);
    s0.p[0] = a0; s0.p[1] = a0;
    long t0b = 0;
    if (b0) {
        C8Args b = *b0;
        t0b = shape(b, TR0);
        if (t0b != t0a) return DKT_E_UNSUPPORTED;       // (the flag words of the two problems are indexed alike)
        s0.p[1] = b;
    }
    const long t1 = shape(a1, TR1);
    s1.p[0] = a1; s1.p[1] = a1;
    if (t0a + t0b > 0x7fffffffL || t1 > 0x7fffffffL) return DKT_E_SHAPE;
    // Every block must be resident while it may wait: the grid never exceeds what the device holds of this kernel, nor
    // max_blocks (two chains on two streams must fit TOGETHER), nor the stage-0 tiles (a block takes the launch
    // count from a stage-0 tile of its own)
    long cap = slots[dev & 63];
    if (max_blocks > 0 && max_blocks < cap) cap = max_blocks;
    long grid = t0a + t0b < cap ? t0a + t0b : cap;
    long nb0 = b0 ? (long)(grid * ((double)t0a * a0.nchunks / ((double)t0a * a0.nchunks + (double)t0b * b0->nchunks)) + 0.5) : grid;
    if (b0) {
        nb0 = nb0 < 1 ? 1 : (nb0 > grid - 1 ? grid - 1 : nb0);
        if (nb0 > t0a) nb0 = t0a;
        if (grid - nb0 > t0b) grid = nb0 + t0b;
    }
    sy.dep_tr = TR0;
    sy.dep_tiles_w = a0.tiles_w; sy.dep_tiles_h = a0.tiles_xy / a0.tiles_w; sy.dep_nco = a0.n_co; sy.dep_nprob = b0 ? 2 : 1;
    sy.dep_tiles_per_prob = (int)t0a;
    sy.target = 0;
    // flags a stage-1 tile polls: (rows) x 3 columns x channel blocks x problems on the 64 lanes of one wave
    const int rows = (TR1 + 1) / TR0 + 2;
    if (rows * 3 * sy.dep_nco * sy.dep_nprob > 64) return DKT_E_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, s0, (int)nb0, s1, sy);
    return dkt_launch_status();
}

extern "C" long dkt_conv2d_c8_chain_flag_words(const dkt_conv_c8_desc *d0, int cfg0, int nprob) {
    if (!d0 || (cfg0 != 3 && cfg0 != 4) || nprob < 1 || nprob > 2) return DKT_E_SHAPE;
    const int TR = cfg0 == 3 ? 8 : 4;
    return (long)nprob * ((d0->W + 31) / 32) * ((d0->H + TR - 1) / TR) * ((d0->Cout + 63) / 64) * d0->B;
}

extern "C" int dkt_conv2d_c8_chain(const dkt_conv_c8_desc *d0a, const dkt_conv_c8_desc *d0b, int cfg0, const dkt_conv_c8_desc *d1,
                                   int cfg1, unsigned *flags, unsigned *err_word, int max_blocks, int timing_only, int device,
                                   void *stream) {
    if (!d0a || !d1 || !flags) return DKT_E_NULL;
    C8Args a0, b0, a1;
    int rc = c8_fill(a0, d0a);
    if (rc != DKT_OK) return rc;
    if (d0b && (rc = c8_fill(b0, d0b)) != DKT_OK) return rc;
    if ((rc = c8_fill(a1, d1)) != DKT_OK) return rc;
    if (d0a->B != d1->B || d0a->H != d1->H || d0a->W != d1->W || (d0b && (d0b->B != d0a->B || d0b->H != d0a->H || d0b->W != d0a->W)))
        return DKT_E_SHAPE;
    const int p = d0a->passes ? d0a->passes : 3;
    if ((d1->passes ? d1->passes : 3) != p || (d0b && (d0b->passes ? d0b->passes : 3) != p)) return DKT_E_UNSUPPORTED;
    if (p == 1 && ((a0.nchunks & 1) || (a1.nchunks & 1) || (d0b && (b0.nchunks & 1)))) return DKT_E_UNSUPPORTED;
    if (a0.epi == 3 || a1.epi == 3 || (d0b && b0.epi == 3)) return DKT_E_UNSUPPORTED;
    C8Sync sy = {};
    sy.pub = flags; sy.dep = flags; sy.err = err_word; sy.timing_only = timing_only ? 1 : 0;
    DKT_ENTER(device);
    hipStream_t st = (hipStream_t)stream;
    const C8Args *pb = d0b ? &b0 : nullptr;
#define C8_CHAIN_CASE(P)                                                                                       \
    if (cfg0 == 4 && cfg1 == 3) return c8_chain_launch<4, 1, 8, 4, 2, 6, P>(a0, pb, d0a->B, a1, sy, max_blocks, st); \
    if (cfg0 == 4 && cfg1 == 4) return c8_chain_launch<4, 1, 8, 4, 1, 8, P>(a0, pb, d0a->B, a1, sy, max_blocks, st);
    if (p == 3) { C8_CHAIN_CASE(3) }
    else if (p == 2) { C8_CHAIN_CASE(2) }
    else { C8_CHAIN_CASE(1) }
#undef C8_CHAIN_CASE
    return DKT_E_UNSUPPORTED;
}

// ---- second layer of the flow / disparity head from the planes of epilogue 3 (core/update.py:10-13, 3x3, padding 1):
//   y[o] = bias[o] + sum over (co block, tap) of plane[(o * n_co + block) * 9 + tap] shifted by the tap;
//   target[o] += y[o]  (raft_stereo.py:165-168: coords1 = coords1 + delta_flow);  diff_out[o] = target[o] - diff_ref[o]
__global__ __launch_bounds__(256) void head_finish_kernel(const float *planes, long planes_bs, int n_co, const float *bias,
                                                          float *target, long target_bs, const float *diff_ref, long diff_ref_bs,
                                                          float *diff_out, long diff_out_bs, int nout, int H, int W) {
    const long HW = (long)H * W;
    const int b = blockIdx.y;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const int oh = (int)(i / W), ow = (int)(i - (long)oh * W);
        for (int o = 0; o < nout; ++o) {
            float s = 0.0f;
            for (int cb = 0; cb < n_co; ++cb) {
                const float *p = planes + (long)b * planes_bs + (long)(o * n_co + cb) * 9 * HW;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
                    const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    const float v = p[(long)t * HW + (in ? (long)ih * W + iw : 0)];
                    s = __fadd_rn(s, in ? v : 0.0f);
                }
            }
            s = __fadd_rn(s, bias ? bias[o] : 0.0f);
            float *tp = target + (long)b * target_bs + (long)o * HW + i;
            const float nv = __fadd_rn(*tp, s);
            *tp = nv;
            if (diff_out) diff_out[(long)b * diff_out_bs + (long)o * HW + i] = __fsub_rn(nv, diff_ref[(long)b * diff_ref_bs + (long)o * HW + i]);
        }
    }
}

extern "C" int dkt_head_finish(const float *planes, long planes_bstride, int n_co, const float *bias, float *target,
                               long target_bstride, const float *diff_ref, long diff_ref_bstride, float *diff_out,
                               long diff_out_bstride, int B, int nout, int H, int W, int device, void *stream) {
    if (!planes || !target || (diff_out && !diff_ref)) return DKT_E_NULL;
    if (B <= 0 || B > 65535 || nout < 1 || nout > 2 || n_co < 1 || H <= 0 || W <= 0) return DKT_E_SHAPE;
    DKT_ENTER(device);
    const long HW = (long)H * W;
    hipLaunchKernelGGL(head_finish_kernel, dim3((unsigned)((HW + 255) / 256 > 2048 ? 2048 : (HW + 255) / 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, planes, planes_bstride, n_co, bias, target, target_bstride, diff_ref, diff_ref_bstride,
                       diff_out, diff_out_bstride, nout, H, W);
    return dkt_launch_status();
}

// number of output-channel blocks (= planes per output and tap) a head launch with tile shape `cfg` produces
extern "C" int dkt_conv2d_c8_head_blocks(int Cout, int cfg) {
    const int n64 = (Cout + 63) / 64;
    if (cfg == 2) return (n64 + 1) / 2;
    if (cfg == 1) return (n64 + 3) / 4;
    return DKT_E_UNSUPPORTED;
}

