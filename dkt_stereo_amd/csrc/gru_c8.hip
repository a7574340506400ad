// gru_c8.hip -- round 4: one ConvGRU step (core/update.py:23-32: z, r, q, h') as ONE persistent kernel on gfx950.
//
// conv_c8.hip ran a ConvGRU as two launches (z|r + gates, q + state update); a third of their time was K-independent:
// one tile per CU, so every block's epilogue hit the memory system at once, the context terms and z went through HBM, and
// the kernel boundary between the two wrote the dirty L2 lines back before the q convolution's first load.
// Here a block owns a tile (8 rows x 32 columns, ALL channels) through both convolutions:
//
//   phase A  z|r convolution, 24 chunks: [h | x].  Output channels are packed so that wave (wm, wn) holds, for the SAME 32
//            hidden channels wm*32 .. +32 and its 4 rows, the z pre-activations (block m = 0) and the r ones (m = 1).
//            The accumulators START at (bias + context term) / scale: cz | cr are read while the DMA pipeline fills,
//            not in the epilogue burst.
//   gate A   z = sigmoid(.) STAYS IN REGISTERS (it is never written); r*h -> the C8S scratch `rh` with write-through (sc1)
//            16-byte stores; then the tile's flag is published (agent scope); the freed r accumulators are re-initialised
//            with (bq + cq) / scale for phase B.
//   phase B  q convolution, x chunks FIRST (they do not depend on r*h), then the 8 chunks of r*h.  Before the first r*h
//            patch is fetched one wave waits for the flags of the 3x3 neighbour tiles (their halo), one agent acquire,
//            then plain DMA loads.  Wave (wm, wn) computes q for exactly the 32 channels x 4 rows whose z it holds.
//   gate B   h' = (1 - z) h + z tanh(.) -> fp32 NCHW in place and the C8S twin in place (a neighbour can only have
//            reached this point after it saw OUR flag, i.e. after we read every patch of the old state).
//
// The weight ring, the activation patches, the counted vmcnt / lgkmcnt waits and the fragment-read schedule are those of
// conv_c8.hip's 256 co x 8 rows shape (WM 4 x WN 2 waves, NF 4, ring 4); phase B runs the same step with ONE 32-channel
// block per wave (MF = 1) and 8 KB weight images in the same 16 KB ring slots.
//
// Residency: the grid never exceeds what the device holds (one block per CU), and a block's next tile is at least
// tiles_w + 2 tiles ahead, so every flag a block waits for is produced by a tile some block has ALREADY started or will
// start without waiting for us (no cycle).  Every spin is bounded; a timeout raises the error word.
#include "dkt_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define G8_MAX_X 3
#define G8_PC 34
#define G8_SPIN_LIMIT (1u << 21)

struct G8Args {
    const char *hc8; long hc8_bs;                 // C8S hidden state: source 0 of z|r, overwritten with h' (bytes per batch item)
    const char *x[G8_MAX_X]; long x_bs[G8_MAX_X]; int x_n16[G8_MAX_X];
    int nx, nxc;                                  // x tensors, their 16-channel chunks in total
    char *rh; long rh_bs;                         // C8S scratch: r*h
    const char *wzr, *wq;                         // step images: z|r [h chunks, x chunks][tap][4 co64], q [x chunks, rh chunks][tap][2 co64]
    const float *bz, *br, *bq;
    const float *cz, *cr, *cq; long cz_bs, cr_bs, cq_bs;
    float *h; long h_bs;                          // fp32 NCHW hidden state, updated in place
    float s_zr, s_q, inv_s_zr, inv_s_q;           // accumulator -> value scales (powers of two) and their inverses
    float act_scale;
    int H, W, Wp; long plane_bytes;
    int tiles_w, tiles_h, tiles_xy, total_tiles;
    unsigned *flags;                              // one word per tile: launches that have published it
};
struct G8ArgsPair {
    G8Args p[2];
    unsigned *err;
};

__device__ __forceinline__ float g8_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float g8_tanh(float x) {
    const float xc = x < -15.0f ? -15.0f : (x > 15.0f ? 15.0f : x);      // NaN passes through
    const float t = __expf(2.0f * xc);
    return (t - 1.0f) * __frcp_rn(t + 1.0f);
}
__device__ __forceinline__ unsigned g8_pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; unsigned u; } v;
    v.h[0] = a;
    v.h[1] = b;
    return v.u;
}
template <int N>
__device__ __forceinline__ void g8_wait_vm() {
    static_assert(N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
template <int OFF>
__device__ __forceinline__ void g8_lds_read(f16x8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void g8_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(1))) unsigned g8_gu32;

// PASSES (round 5): fp16 MFMA products per weight x activation block, as conv_c8.hip's: 3 = fp32-class (w_hi x_hi + w_lo x_hi +
// w_hi x_lo), 2 = without w_hi x_lo, 1 = w_hi x_hi only.  Gates, state update and every stored tensor are the same in all three.
template <int PASSES>
__global__ __launch_bounds__(512, 1) void gru_c8_kernel(G8ArgsPair ap, int nb0) {
    static_assert(PASSES >= 1 && PASSES <= 3, "passes");
    constexpr int NW = 8, WM = 4, NF = 4, TR = 8, PR = TR + 2;
    constexpr int NPP = PR * G8_PC;                  // 340 patch pixels
    constexpr int NU = NPP * 4;                      // 16-byte units per chunk
    constexpr int NPR = (NU + 63) / 64;              // 22 one-KiB DMA pieces
    constexpr int NIA = (NPR + NW - 1) / NW;         // 3 per wave
    constexpr int ACT_BYTES = (NIA * NW > NPR ? NPR + 1 : NPR) * 1024;
    constexpr int WSLOT = WM * 4096;
    constexpr int RING = 4;
    extern __shared__ __attribute__((aligned(16))) char lds[];       // act[2][ACT_BYTES] | wring[RING][WSLOT]
    char *const lds_act = lds;
    char *const lds_w = lds + 2 * ACT_BYTES;

    const bool second = (int)blockIdx.x >= nb0;
    const G8Args &a = ap.p[second ? 1 : 0];
    const int blk_first = second ? nb0 : 0;
    const int blk_count = second ? (int)gridDim.x - nb0 : nb0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int li = lane & 31, kg = lane >> 5;
    const int iHW = a.H * a.W;
    const int nA = 8 + a.nxc, nB = a.nxc + 8;

    int tile = dkt_xcd_tile((int)blockIdx.x - blk_first, blk_count);      // (neighbouring tiles on one XCD: dkt_common.h)
    int b = tile / a.tiles_xy;
    int txy = tile - b * a.tiles_xy;
    int w0 = (txy % a.tiles_w) * 32, h0 = (txy / a.tiles_w) * TR;

    // ---- activation DMA (as conv_c8.hip): piece p = j * NW + wave covers units u = 64 p + lane of the chunk image
    unsigned aoff_cur[NIA], aoff_nxt[NIA];
    auto tile_offsets = [&](int th0, int tw0, unsigned (&off)[NIA]) {
#pragma unroll
        for (int j = 0; j < NIA; ++j) {
            int u = 64 * (j * NW + wave) + lane;
            u = u < NU ? u : 0;
            const int q = u / NPP, pp = u - q * NPP;
            const int pr = pp / G8_PC, pc = pp - pr * G8_PC;
            off[j] = (unsigned)(q * a.plane_bytes + ((long)(th0 + pr) * a.Wp + (tw0 + pc)) * 16);
        }
    };
    auto x_chunk = [&](int tb, int c) -> const char * {              // wave-uniform
        int s = 0;
        while (s + 1 < a.nx && c >= a.x_n16[s]) {
            c -= a.x_n16[s];
            ++s;
        }
        return a.x[s] + (long)tb * a.x_bs[s] + (long)c * 4 * a.plane_bytes;
    };
    auto chunk_A = [&](int tb, int c) -> const char * {
        return c < 8 ? a.hc8 + (long)tb * a.hc8_bs + (long)c * 4 * a.plane_bytes : x_chunk(tb, c - 8);
    };
    auto chunk_B = [&](int tb, int c) -> const char * {
        return c < a.nxc ? x_chunk(tb, c) : a.rh + (long)tb * a.rh_bs + (long)(c - a.nxc) * 4 * a.plane_bytes;
    };
    auto issue_act = [&](const char *base, const unsigned (&off)[NIA], int buf) {
        char *dst = lds_act + buf * ACT_BYTES;
#pragma unroll
        for (int j = 0; j < NIA; ++j)
            __builtin_amdgcn_global_load_lds((const void *)(base + off[j]),
                                             (__attribute__((address_space(3))) void *)(dst + min(j * NW + wave, NPR) * 1024), 16, 0, 0);
    };
    // ---- weight DMA: a phase-A step image is 16 KB (two 1-KiB pieces per wave), a phase-B one 8 KB (one piece per wave)
    auto issue_wA = [&](const char *img, int slot) {
        char *dst = lds_w + slot * WSLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = wave * 2 + j;
            __builtin_amdgcn_global_load_lds((const void *)(img + p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
        }
    };
    auto issue_wB = [&](const char *img, int slot) {
        char *dst = lds_w + slot * WSLOT;
        __builtin_amdgcn_global_load_lds((const void *)(img + wave * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(dst + wave * 1024), 16, 0, 0);
    };
    constexpr long WSTEP_A = 4 * 4096, WSTEP_B = 2 * 4096;

    // acc[0][n]: z pre-activation, then z itself, of channels wm*32 + 4 kg + (r & 3) + 8 (r >> 2), pixel (row wn*4 + n, column li)
    // acc[1][n]: r pre-activation in phase A, the q accumulator in phase B (same channels, same pixels)
    f32x16 acc[2][NF];
    const int ch_lane = wm * 32 + 4 * kg;            // + (r & 3) + 8 * (r >> 2)

    // Channel-strided fp32 accesses of the gates: channel ch_lane + cu(r), cu(r) = (r & 3) + 8 (r >> 2).  The per-r part goes
    // into a wave-uniform base (scalar registers), the lane part is ONE 32-bit offset per pixel row: no 64-bit per-lane
    // addresses (sixteen of those per row were spilled and re-loaded one by one, each behind a drained vmcnt).
#define G8_CU(r) (((r) & 3) + 8 * ((r) >> 2))
    const unsigned lane_ch_off = (unsigned)(ch_lane * iHW);
    auto row_off = [&](int th0, int tw0, int n, bool &inside, int &oh, int &ow) -> unsigned {
        oh = th0 + wn * NF + n;
        ow = tw0 + li;
        inside = oh < a.H && ow < a.W;
        return lane_ch_off + (inside ? (unsigned)(oh * a.W + ow) : 0u);
    };
    // accumulators of phase A from the context terms: (bias + c) / scale (scale is a power of two: exact)
    auto init_A = [&](int tb, int th0, int tw0) {
        const float *pz = a.cz + (long)tb * a.cz_bs, *pr = a.cr + (long)tb * a.cr_bs;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            bool inside; int oh, ow;
            const unsigned vo = row_off(th0, tw0, n, inside, oh, ow);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][n][r] = (pz + (long)G8_CU(r) * iHW)[vo];
                acc[1][n][r] = (pr + (long)G8_CU(r) * iHW)[vo];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float vz = (a.bz + G8_CU(r))[ch_lane], vr = (a.br + G8_CU(r))[ch_lane];
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                acc[0][n][r] = __fmul_rn(__fadd_rn(vz, acc[0][n][r]), a.inv_s_zr);
                acc[1][n][r] = __fmul_rn(__fadd_rn(vr, acc[1][n][r]), a.inv_s_zr);
            }
        }
    };

    // fragment addresses: A (phase A) = slot + wm*4096 + hl*2048 + kg*1024 + (m*32 + li)*16
    //                     A (phase B) = slot + (wm>>1)*4096 + hl*2048 + kg*1024 + ((wm&1)*32 + li)*16
    //                     B = act + ((2 kg + hl) * NPP + (wn*NF + n + dy) * 34 + li + dx) * 16
    const int a_laneA = wm * 4096 + kg * 1024 + li * 16;
    const int a_laneB = (wm >> 1) * 4096 + kg * 1024 + ((wm & 1) * 32 + li) * 16;
    const int b_lane = (2 * kg * NPP + wn * NF * G8_PC + li) * 16;
    f16x8 Ahi[2], Alo[2], Bhi[NF + 2], Blo[NF + 2];
    const unsigned lds_w_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds_w;
    const unsigned lds_wA = lds_w_base + a_laneA, lds_wB = lds_w_base + a_laneB;
    const unsigned lds_b_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds_act + b_lane;

#define G8_MM(A, m, B, r, n)                                                                  \
    {                                                                                         \
        acc[MFPV == 2 ? m : 1][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[m], B[r], acc[MFPV == 2 ? m : 1][n], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
#define G8_RD(dst, off, addr)                  \
    {                                          \
        g8_lds_read<(off)>(dst, addr);         \
        __builtin_amdgcn_sched_barrier(0);     \
    }
#define G8_ROW(r, dx, plane) ((plane) + ((r) * G8_PC + (dx)) * 16)

    // One (chunk, tap) step, conv_c8.hip's C8_STEP at NF = 4 with MFP = 2 (phase A) or 1 (phase B) channel blocks per wave.
    // `last` (the phase's last chunk): from step 6 on the ring is fed with the NEXT phase's images (w_next, other piece
    // count), so the waits of steps 7 and 8 count pieces of the other kind: vmcnt(1) is exact for B images and merely
    // early for A ones.
#define G8_STEP(T, MFP)                                                                                                \
    {                                                                                                                  \
        constexpr int DX = (T) / 3, DY = (T) % 3, NDX = (DX + 1) % 3;                                                  \
        constexpr int WPI = (MFP), MFPV = (MFP);                                                                       \
        const int sl1 = sl + 1 == RING ? 0 : sl + 1, sl2 = sl == 0 ? RING - 1 : sl - 1;                                \
        const unsigned adw_s = (MFP == 2 ? lds_wA : lds_wB) + sl * WSLOT;                                              \
        const unsigned adw_n = (MFP == 2 ? lds_wA : lds_wB) + sl1 * WSLOT;                                             \
        const unsigned adb_c = lds_b_addr + cur * ACT_BYTES;                                                           \
        const unsigned adb_n = lds_b_addr + (DX < 2 ? cur : nxt) * ACT_BYTES;                                          \
        if ((T) == 1) g8_wait_vm<WPI + NIA>();                                                                         \
        else if ((T) >= 7 && MFP == 2) { if (last) g8_wait_vm<1>(); else g8_wait_vm<2>(); }                            \
        else g8_wait_vm<WPI>();                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                  \
        g8_wait_lgkm<2>();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        /* ---- X (Alo x Bhi) */                                                                                       \
        G8_MM(Alo, 0, Bhi, DY, 0) G8_RD(Ahi[0], 0, adw_s)                                                              \
        if constexpr (MFP == 2) { G8_MM(Alo, 1, Bhi, DY, 0) G8_RD(Ahi[1], 512, adw_s) }                                \
        G8_MM(Alo, 0, Bhi, 1 + DY, 1)                                                                                  \
        if constexpr (DY < 2) G8_RD(Bhi[NF + DY], G8_ROW(NF + DY, DX, 0), adb_c)                                       \
        if constexpr (MFP == 2) G8_MM(Alo, 1, Bhi, 1 + DY, 1)                                                          \
        if constexpr (MFP == 1) G8_MM(Alo, 0, Bhi, 2 + DY, 2)                                                          \
        if constexpr (DY < 2) G8_RD(Blo[NF + DY], G8_ROW(NF + DY, DX, NPP * 16), adb_c)                                \
        if constexpr (MFP == 2) { G8_MM(Alo, 0, Bhi, 2 + DY, 2) G8_MM(Alo, 1, Bhi, 2 + DY, 2) }                        \
        G8_MM(Alo, 0, Bhi, 3 + DY, 3)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Alo, 1, Bhi, 3 + DY, 3)                                                          \
        /* ---- DMA issue: the image of step s+3 into the slot of step s-1; at the chunk's first step the next patch */ \
        if ((T) == 6 && last) wptr = w_next;                                                                           \
        if ((T) == 0) { if (use_nxt_off) issue_act(act_f, aoff_nxt, nxt); else issue_act(act_f, aoff_cur, nxt); }      \
        if ((T) >= 6 && last) {                                                                                        \
            if constexpr (MFP == 2) { issue_wB(wptr, sl2); wptr += WSTEP_B; }                                          \
            else { issue_wA(wptr, sl2); wptr += WSTEP_A; }                                                             \
        } else {                                                                                                       \
            if constexpr (MFP == 2) { issue_wA(wptr, sl2); wptr += WSTEP_A; }                                          \
            else { issue_wB(wptr, sl2); wptr += WSTEP_B; }                                                             \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        g8_wait_lgkm<(DY < 2 ? 2 : 0)>();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        /* ---- Y (Ahi x Bhi) */                                                                                       \
        G8_MM(Ahi, 0, Bhi, DY, 0) G8_RD(Alo[0], 2048, adw_n)                                                           \
        if constexpr (MFP == 2) { G8_MM(Ahi, 1, Bhi, DY, 0) G8_RD(Alo[1], 2048 + 512, adw_n) }                         \
        if constexpr (MFP == 1) G8_MM(Ahi, 0, Bhi, 1 + DY, 1)                                                          \
        if constexpr (DY == 0) G8_RD(Bhi[0], G8_ROW(0, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 1) G8_RD(Bhi[1], G8_ROW(1, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 2) G8_RD(Bhi[2], G8_ROW(2, NDX, 0), adb_n)                                                 \
        if constexpr (MFP == 2) { G8_MM(Ahi, 0, Bhi, 1 + DY, 1) G8_MM(Ahi, 1, Bhi, 1 + DY, 1) }                        \
        if constexpr (MFP == 1) G8_MM(Ahi, 0, Bhi, 2 + DY, 2)                                                          \
        if constexpr (DY == 2) G8_RD(Bhi[3], G8_ROW(3, NDX, 0), adb_n)                                                 \
        if constexpr (MFP == 2) { G8_MM(Ahi, 0, Bhi, 2 + DY, 2) G8_MM(Ahi, 1, Bhi, 2 + DY, 2) }                        \
        G8_MM(Ahi, 0, Bhi, 3 + DY, 3)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Bhi, 3 + DY, 3)                                                          \
        /* ---- Z (Ahi x Blo) */                                                                                       \
        G8_MM(Ahi, 0, Blo, DY, 0)                                                                                      \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Blo, DY, 0)                                                              \
        if constexpr (DY == 0) G8_RD(Blo[0], G8_ROW(0, NDX, NPP * 16), adb_n)                                          \
        if constexpr (DY == 1) G8_RD(Blo[1], G8_ROW(1, NDX, NPP * 16), adb_n)                                          \
        if constexpr (DY == 2) G8_RD(Blo[2], G8_ROW(2, NDX, NPP * 16), adb_n)                                          \
        G8_MM(Ahi, 0, Blo, 1 + DY, 1)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Blo, 1 + DY, 1)                                                          \
        if constexpr (DY == 2) G8_RD(Blo[3], G8_ROW(3, NDX, NPP * 16), adb_n)                                          \
        G8_MM(Ahi, 0, Blo, 2 + DY, 2)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Blo, 2 + DY, 2)                                                          \
        G8_MM(Ahi, 0, Blo, 3 + DY, 3)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Blo, 3 + DY, 3)                                                          \
        sl = sl1;                                                                                                      \
    }
    // the fragments a phase's first step does not fetch itself (after the DMA of its first patch and image has landed).
    // PASSES == 1 keeps the A fragments double-buffered in Ahi / Alo: a chunk has nine steps, so even chunks of a phase compute
    // their first step from Ahi and odd ones from Alo -- the chunk loops run two chunks per trip (the host refuses odd chunk
    // counts at one pass), because a run-time choice between the two step sequences sent the accumulators to scratch.
#define G8_FIRST_FRAGS(MFP)                                                                              \
    {                                                                                                    \
        const unsigned aw = (MFP == 2 ? lds_wA : lds_wB) + sl * WSLOT, ab = lds_b_addr + (g & 1) * ACT_BYTES; \
        if constexpr (PASSES == 1) {                                                                     \
            G8_RD(Ahi[0], 0, aw) if constexpr (MFP == 2) G8_RD(Ahi[1], 512, aw)                          \
        } else {                                                                                         \
            G8_RD(Alo[0], 2048, aw)                                                                      \
            if constexpr (MFP == 2) G8_RD(Alo[1], 2048 + 512, aw)                                        \
        }                                                                                                \
        G8_RD(Bhi[0], G8_ROW(0, 0, 0), ab) G8_RD(Bhi[1], G8_ROW(1, 0, 0), ab)                            \
        G8_RD(Bhi[2], G8_ROW(2, 0, 0), ab) G8_RD(Bhi[3], G8_ROW(3, 0, 0), ab)                            \
        if constexpr (PASSES == 3) {                                                                     \
            G8_RD(Blo[0], G8_ROW(0, 0, NPP * 16), ab) G8_RD(Blo[1], G8_ROW(1, 0, NPP * 16), ab)          \
            G8_RD(Blo[2], G8_ROW(2, 0, NPP * 16), ab) G8_RD(Blo[3], G8_ROW(3, 0, NPP * 16), ab)          \
        }                                                                                                \
        g8_wait_lgkm<0>();                                                                               \
    }
    // ---- reduced-pass steps (conv_c8.hip's C8_STEP2 / C8_STEP1 at NF = 4): same DMA stream and vmcnt waits as G8_STEP
#define G8_STEP_HEAD(T, MFP)                                                                                           \
        constexpr int DX = (T) / 3, DY = (T) % 3, NDX = (DX + 1) % 3;                                                  \
        constexpr int WPI = (MFP), MFPV = (MFP);                                                                       \
        const int sl1 = sl + 1 == RING ? 0 : sl + 1, sl2 = sl == 0 ? RING - 1 : sl - 1;                                \
        const unsigned adw_s = (MFP == 2 ? lds_wA : lds_wB) + sl * WSLOT;                                              \
        const unsigned adw_n = (MFP == 2 ? lds_wA : lds_wB) + sl1 * WSLOT;                                             \
        const unsigned adb_c = lds_b_addr + cur * ACT_BYTES;                                                           \
        const unsigned adb_n = lds_b_addr + (DX < 2 ? cur : nxt) * ACT_BYTES;                                          \
        (void)adw_s; (void)adw_n; (void)adb_c; (void)adb_n;                                                            \
        if ((T) == 1) g8_wait_vm<WPI + NIA>();                                                                         \
        else if ((T) >= 7 && MFP == 2) { if (last) g8_wait_vm<1>(); else g8_wait_vm<2>(); }                            \
        else g8_wait_vm<WPI>();                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                  \
        if constexpr (DY == 0) g8_wait_lgkm<0>(); else g8_wait_lgkm<1>();                                              \
        __builtin_amdgcn_sched_barrier(0);
#define G8_STEP_DMA(T, MFP)                                                                                            \
        if ((T) == 6 && last) wptr = w_next;                                                                           \
        if ((T) == 0) { if (use_nxt_off) issue_act(act_f, aoff_nxt, nxt); else issue_act(act_f, aoff_cur, nxt); }      \
        if ((T) >= 6 && last) {                                                                                        \
            if constexpr (MFP == 2) { issue_wB(wptr, sl2); wptr += WSTEP_B; }                                          \
            else { issue_wA(wptr, sl2); wptr += WSTEP_A; }                                                             \
        } else {                                                                                                       \
            if constexpr (MFP == 2) { issue_wA(wptr, sl2); wptr += WSTEP_A; }                                          \
            else { issue_wB(wptr, sl2); wptr += WSTEP_B; }                                                             \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);
    // two passes.  LDS reads in issue order: X: Ahi[0] (, Ahi[1]); dy < 2: row NF + dy.  Y: Alo'[0] (, Alo'[1]), the dying rows
#define G8_STEP2(T, MFP)                                                                                               \
    {                                                                                                                  \
        G8_STEP_HEAD(T, MFP)                                                                                           \
        G8_MM(Alo, 0, Bhi, DY, 0) G8_RD(Ahi[0], 0, adw_s)                                                              \
        if constexpr (MFP == 2) { G8_MM(Alo, 1, Bhi, DY, 0) G8_RD(Ahi[1], 512, adw_s) }                                \
        G8_MM(Alo, 0, Bhi, 1 + DY, 1)                                                                                  \
        if constexpr (DY < 2) G8_RD(Bhi[NF + DY], G8_ROW(NF + DY, DX, 0), adb_c)                                       \
        if constexpr (MFP == 2) G8_MM(Alo, 1, Bhi, 1 + DY, 1)                                                          \
        G8_MM(Alo, 0, Bhi, 2 + DY, 2)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Alo, 1, Bhi, 2 + DY, 2)                                                          \
        G8_MM(Alo, 0, Bhi, 3 + DY, 3)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Alo, 1, Bhi, 3 + DY, 3)                                                          \
        G8_STEP_DMA(T, MFP)                                                                                            \
        g8_wait_lgkm<(DY < 2 ? 1 : 0)>();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        G8_MM(Ahi, 0, Bhi, DY, 0) G8_RD(Alo[0], 2048, adw_n)                                                           \
        if constexpr (MFP == 2) { G8_MM(Ahi, 1, Bhi, DY, 0) G8_RD(Alo[1], 2048 + 512, adw_n) }                         \
        if constexpr (DY == 0) G8_RD(Bhi[0], G8_ROW(0, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 1) G8_RD(Bhi[1], G8_ROW(1, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 2) G8_RD(Bhi[2], G8_ROW(2, NDX, 0), adb_n)                                                 \
        G8_MM(Ahi, 0, Bhi, 1 + DY, 1)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Bhi, 1 + DY, 1)                                                          \
        if constexpr (DY == 2) G8_RD(Bhi[3], G8_ROW(3, NDX, 0), adb_n)                                                 \
        G8_MM(Ahi, 0, Bhi, 2 + DY, 2)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Bhi, 2 + DY, 2)                                                          \
        G8_MM(Ahi, 0, Bhi, 3 + DY, 3)                                                                                  \
        if constexpr (MFP == 2) G8_MM(Ahi, 1, Bhi, 3 + DY, 3)                                                          \
        sl = sl1;                                                                                                      \
    }
    // one pass: A fragments double-buffered (PA: this step's, PB: the next step's, fetched behind the barrier), row by row
#define G8_STEP1(T, MFP, PA, PB)                                                                                       \
    {                                                                                                                  \
        G8_STEP_HEAD(T, MFP)                                                                                           \
        G8_RD(PB[0], 0, adw_n)                                                                                         \
        if constexpr (MFP == 2) G8_RD(PB[1], 512, adw_n)                                                               \
        if constexpr (DY < 2) G8_RD(Bhi[NF + DY], G8_ROW(NF + DY, DX, 0), adb_c)                                       \
        G8_MM(PA, 0, Bhi, DY, 0)                                                                                       \
        if constexpr (MFP == 2) G8_MM(PA, 1, Bhi, DY, 0)                                                               \
        if constexpr (DY == 0) G8_RD(Bhi[0], G8_ROW(0, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 1) G8_RD(Bhi[1], G8_ROW(1, NDX, 0), adb_n)                                                 \
        if constexpr (DY == 2) G8_RD(Bhi[2], G8_ROW(2, NDX, 0), adb_n)                                                 \
        G8_STEP_DMA(T, MFP)                                                                                            \
        G8_MM(PA, 0, Bhi, 1 + DY, 1)                                                                                   \
        if constexpr (MFP == 2) G8_MM(PA, 1, Bhi, 1 + DY, 1)                                                           \
        if constexpr (DY == 2) G8_RD(Bhi[3], G8_ROW(3, NDX, 0), adb_n)                                                 \
        G8_MM(PA, 0, Bhi, 2 + DY, 2)                                                                                   \
        if constexpr (MFP == 2) G8_MM(PA, 1, Bhi, 2 + DY, 2)                                                           \
        G8_MM(PA, 0, Bhi, 3 + DY, 3)                                                                                   \
        if constexpr (MFP == 2) G8_MM(PA, 1, Bhi, 3 + DY, 3)                                                           \
        sl = sl1;                                                                                                      \
    }
#define G8_CHUNK(MFP)                                                                                                  \
    if constexpr (PASSES == 3) {                                                                                       \
        G8_STEP(0, MFP) G8_STEP(1, MFP) G8_STEP(2, MFP) G8_STEP(3, MFP) G8_STEP(4, MFP) G8_STEP(5, MFP) G8_STEP(6, MFP) \
        G8_STEP(7, MFP) G8_STEP(8, MFP)                                                                                \
    } else {                                                                                                           \
        G8_STEP2(0, MFP) G8_STEP2(1, MFP) G8_STEP2(2, MFP) G8_STEP2(3, MFP) G8_STEP2(4, MFP) G8_STEP2(5, MFP)          \
        G8_STEP2(6, MFP) G8_STEP2(7, MFP) G8_STEP2(8, MFP)                                                             \
    }
#define G8_CHUNK1(MFP, PA, PB)                                                                                         \
    G8_STEP1(0, MFP, PA, PB) G8_STEP1(1, MFP, PB, PA) G8_STEP1(2, MFP, PA, PB) G8_STEP1(3, MFP, PB, PA)                \
    G8_STEP1(4, MFP, PA, PB) G8_STEP1(5, MFP, PB, PA) G8_STEP1(6, MFP, PA, PB) G8_STEP1(7, MFP, PB, PA)                \
    G8_STEP1(8, MFP, PA, PB)

    // ---- C8S stores: the lane's channel quads (r >> 2 = j) of a pair (2 jp, 2 jp + 1) are completed to 8-channel groups with
    // lane ^ 32 (v_permlane32_swap), after which the lane holds group 2 jp + kg of the wave's 32-channel block
    auto split_pair = [&](const float (&va)[4], const float (&vb)[4], u32x4 &hi, u32x4 &lo) {
        unsigned ha[2], la[2], hb[2], lb[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float x0 = va[2 * d] * a.act_scale, x1 = va[2 * d + 1] * a.act_scale;
            const float y0 = vb[2 * d] * a.act_scale, y1 = vb[2 * d + 1] * a.act_scale;
            const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1, b0 = (_Float16)y0, b1 = (_Float16)y1;
            ha[d] = g8_pack_h2(a0, a1);
            la[d] = g8_pack_h2((_Float16)(x0 - (float)a0), (_Float16)(x1 - (float)a1));
            hb[d] = g8_pack_h2(b0, b1);
            lb[d] = g8_pack_h2((_Float16)(y0 - (float)b0), (_Float16)(y1 - (float)b1));
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            auto r = __builtin_amdgcn_permlane32_swap(ha[d], hb[d], false, false);
            ha[d] = r[0]; hb[d] = r[1];
            auto q = __builtin_amdgcn_permlane32_swap(la[d], lb[d], false, false);
            la[d] = q[0]; lb[d] = q[1];
        }
        hi = (u32x4){ha[0], ha[1], hb[0], hb[1]};
        lo = (u32x4){la[0], la[1], lb[0], lb[1]};
    };

    unsigned target = 0, target_raw = 0;       // the flag value this launch publishes for its tiles
    auto read_target = [&]() {                 // (the load is waited for at the gate, not here)
        target_raw = __hip_atomic_load((g8_gu32 *)a.flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- gate A: z stays in acc[0]; r*h -> rh (write-through); acc[1] <- (bq + cq) / scale
    // Memory order matters more than arithmetic here (the wave's vmcnt is in-order: a load issued behind a write-through store
    // is not back before that store has reached memory): the state values of block n + 1 are requested before block n is
    // computed, and a block's context loads go out before its stores.
    auto gate_A = [&]() {
        const float *ph = a.h + (long)b * a.h_bs;
        const float *pq = a.cq + (long)b * a.cq_bs;
        // buffer descriptor of this batch item's r*h tensor (wave-uniform): 16-byte sc1 stores = write-through to memory
        char *rhb = a.rh + (long)b * a.rh_bs;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)rhb);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)rhb >> 32));
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((size_t)hi32 << 32) | lo32), 0, (int)(16 * 2 * a.plane_bytes), 0x00020000);
        float hv[2][16];
        {
            bool inside; int oh, ow;
            const unsigned vo = row_off(h0, w0, 0, inside, oh, ow);
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[0][r] = (ph + (long)G8_CU(r) * iHW)[vo];
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            bool inside; int oh, ow;
            const unsigned vo = row_off(h0, w0, n, inside, oh, ow);
            if (n + 1 < NF) {
                bool i2; int oh2, ow2;
                const unsigned vo2 = row_off(h0, w0, n + 1, i2, oh2, ow2);
#pragma unroll
                for (int r = 0; r < 16; ++r) hv[(n + 1) & 1][r] = (ph + (long)G8_CU(r) * iHW)[vo2];
            }
            __builtin_amdgcn_sched_barrier(0);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][n][r] = g8_sigmoid(__fmul_rn(acc[0][n][r], a.s_zr));
                v[r] = __fmul_rn(g8_sigmoid(__fmul_rn(acc[1][n][r], a.s_zr)), hv[n & 1][r]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][n][r] = (pq + (long)G8_CU(r) * iHW)[vo];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const float va[4] = {v[8 * jp], v[8 * jp + 1], v[8 * jp + 2], v[8 * jp + 3]};
                const float vb[4] = {v[8 * jp + 4], v[8 * jp + 5], v[8 * jp + 6], v[8 * jp + 7]};
                u32x4 hi, lo;
                split_pair(va, vb, hi, lo);
                const int gq = wm * 4 + 2 * jp + kg;
                const unsigned off = (unsigned)((long)gq * 2 * a.plane_bytes + ((long)(oh + 1) * a.Wp + (ow + 1)) * 16);
                if (inside) {
                    __builtin_amdgcn_raw_buffer_store_b128(hi, rsrc, off, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(lo, rsrc, off + (unsigned)a.plane_bytes, 0, 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float vq = (a.bq + G8_CU(r))[ch_lane];
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[1][n][r] = __fmul_rn(__fadd_rn(vq, acc[1][n][r]), a.inv_s_q);
        }
    };
    // ---- gate B: h' = (1 - z) h + z tanh(q) -> fp32 in place + C8S in place
    auto gate_B = [&]() {
        float *ph = a.h + (long)b * a.h_bs;
        char *pc8 = const_cast<char *>(a.hc8) + (long)b * a.hc8_bs;
        float hv[2][16];
        {
            bool inside; int oh, ow;
            const unsigned vo = row_off(h0, w0, 0, inside, oh, ow);
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[0][r] = (ph + (long)G8_CU(r) * iHW)[vo];
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            bool inside; int oh, ow;
            const unsigned vo = row_off(h0, w0, n, inside, oh, ow);
            if (n + 1 < NF) {
                bool i2; int oh2, ow2;
                const unsigned vo2 = row_off(h0, w0, n + 1, i2, oh2, ow2);
#pragma unroll
                for (int r = 0; r < 16; ++r) hv[(n + 1) & 1][r] = (ph + (long)G8_CU(r) * iHW)[vo2];
            }
            __builtin_amdgcn_sched_barrier(0);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = acc[0][n][r];
                const float q = g8_tanh(__fmul_rn(acc[1][n][r], a.s_q));
                v[r] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, z), hv[n & 1][r]), __fmul_rn(z, q));
            }
            if (inside) {
#pragma unroll
                for (int r = 0; r < 16; ++r) (ph + (long)G8_CU(r) * iHW)[vo] = v[r];
            }
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const float va[4] = {v[8 * jp], v[8 * jp + 1], v[8 * jp + 2], v[8 * jp + 3]};
                const float vb[4] = {v[8 * jp + 4], v[8 * jp + 5], v[8 * jp + 6], v[8 * jp + 7]};
                u32x4 hi, lo;
                split_pair(va, vb, hi, lo);
                const int gq = wm * 4 + 2 * jp + kg;
                char *p = pc8 + (long)gq * 2 * a.plane_bytes + ((long)(oh + 1) * a.Wp + (ow + 1)) * 16;
                if (inside) {
                    *(u32x4 *)p = hi;
                    *(u32x4 *)(p + a.plane_bytes) = lo;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- the flags of the 3x3 neighbour tiles (their r*h halo), polled by wave 0: lanes 0..7 one neighbour each
    auto wait_neighbours = [&]() {
        if (wave == 0) {
            const int k = lane < 4 ? lane : lane + 1;                // 0..8 without the centre
            const int ty = txy / a.tiles_w + k / 3 - 1, tx = txy % a.tiles_w + k % 3 - 1;
            const bool valid = lane < 8 && ty >= 0 && ty < a.tiles_h && tx >= 0 && tx < a.tiles_w;
            g8_gu32 *f = (g8_gu32 *)a.flags + (valid ? b * a.tiles_xy + ty * a.tiles_w + tx : tile);
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = !valid || (int)(v - target) >= 0;
                if (__all(ok)) break;
                if (spins > G8_SPIN_LIMIT) {
                    if (lane == 0 && ap.err) __hip_atomic_fetch_or((g8_gu32 *)ap.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bit 0: this launch
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // one buffer_inv sc1: this CU's L1 drops its stale lines
        }
    };

    // ------------------------------------------------------------------------------------------------
    // main stream
    // ------------------------------------------------------------------------------------------------
    // @trace(0)
    read_target();
    tile_offsets(h0, w0, aoff_cur);
    issue_act(chunk_A(b, 0), aoff_cur, 0);
    const char *wptr = a.wzr;
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) {
        issue_wA(wptr, s);
        wptr += WSTEP_A;
    }
    init_A(b, h0, w0);
    g8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    int g = 0;              // chunks consumed by this block: activation buffer parity
    int sl = 0;             // ring slot of the step being computed
    G8_FIRST_FRAGS(2)
    // @trace(1)
    for (;;) {
        const int tn = tile + blk_count;
        const bool have_next = tn < a.total_tiles;
        const int nb = have_next ? tn / a.tiles_xy : b;
        const int ntxy = have_next ? tn - nb * a.tiles_xy : txy;
        const int nw0 = (ntxy % a.tiles_w) * 32, nh0 = (ntxy / a.tiles_w) * TR;
        // ---------------- phase A: z | r
#define G8_A_SETUP(c)                                                              \
            const bool last = (c) + 1 == nA;                                           \
            const char *act_f = last ? chunk_B(b, 0) : chunk_A(b, (c) + 1);            \
            const char *w_next = a.wq;                                                 \
            constexpr bool use_nxt_off = false;                                        \
            const int cur = g & 1, nxt = cur ^ 1;
        if constexpr (PASSES == 1) {
            for (int c = 0; c < nA; c += 2) {
                { G8_A_SETUP(c) G8_CHUNK1(2, Ahi, Alo) }
                ++g;
                { G8_A_SETUP(c + 1) G8_CHUNK1(2, Alo, Ahi) }
                ++g;
            }
        } else {
            for (int c = 0; c < nA; ++c, ++g) {
                G8_A_SETUP(c)
                G8_CHUNK(2)
            }
        }
        g8_wait_lgkm<0>();
        // @trace(2)
        target = __builtin_amdgcn_readfirstlane(target_raw + 1u);
        gate_A();
        // publish: every storing wave drains (R1), then ONE lane stores the flag with agent scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) __hip_atomic_store((g8_gu32 *)a.flags + tile, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        G8_FIRST_FRAGS(1)
        // @trace(3)
        // ---------------- phase B: q
#define G8_B_SETUP(c)                                                                                                  \
            const bool last = (c) + 1 == nB;                                                                           \
            if ((c) == a.nxc - 1) wait_neighbours();       /* before the barrier that precedes the first r*h patch's DMA */ \
            const char *act_f = last ? (have_next ? chunk_A(nb, 0) : chunk_B(b, (c))) : chunk_B(b, (c) + 1);           \
            const char *w_next = a.wzr;                                                                                \
            const bool use_nxt_off = last && have_next;                                                                \
            if (use_nxt_off) tile_offsets(nh0, nw0, aoff_nxt);                                                         \
            const int cur = g & 1, nxt = cur ^ 1;
        if constexpr (PASSES == 1) {
            for (int c = 0; c < nB; c += 2) {
                { G8_B_SETUP(c) G8_CHUNK1(1, Ahi, Alo) }
                ++g;
                { G8_B_SETUP(c + 1) G8_CHUNK1(1, Alo, Ahi) }
                ++g;
            }
        } else {
            for (int c = 0; c < nB; ++c, ++g) {
                G8_B_SETUP(c)
                G8_CHUNK(1)
            }
        }
        g8_wait_lgkm<0>();
        // @trace(4)
        gate_B();
        // @trace(5)
        if (!have_next) break;
        tile = tn; b = nb; txy = ntxy; h0 = nh0; w0 = nw0;
#pragma unroll
        for (int j = 0; j < NIA; ++j) aoff_cur[j] = aoff_nxt[j];
        read_target();
        init_A(b, h0, w0);
        G8_FIRST_FRAGS(2)
    }
    g8_wait_vm<0>();        // no DMA may land in this block's LDS after it has been released
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int g8_fill(G8Args &a, const dkt_gru_c8_desc *d) {
    if (!d) return DKT_E_NULL;
    if (d->B <= 0 || d->B > 65535 || d->H <= 0 || d->W <= 0) return DKT_E_SHAPE;
    if (d->hidden != 128) return DKT_E_UNSUPPORTED;            // four waves x 32 channels
    if (d->nx < 1 || d->nx > G8_MAX_X) return DKT_E_SHAPE;
    if (!d->h_c8 || !d->rh_c8 || !d->w_zr || !d->w_q || !d->bz || !d->br || !d->bq || !d->cz || !d->cr || !d->cq || !d->h || !d->flags)
        return DKT_E_NULL;
    if (!(d->scale_zr > 0.0f) || !(d->scale_q > 0.0f) || !(d->act_scale > 0.0f)) return DKT_E_SHAPE;
    int Hp, Wp;
    dkt_act_c8_dims(d->H, d->W, &Hp, &Wp);
    a.hc8 = (const char *)d->h_c8; a.hc8_bs = d->h_c8_bstride;
    a.nxc = 0;
    for (int s = 0; s < G8_MAX_X; ++s) {
        a.x[s] = s < d->nx ? (const char *)d->x[s] : nullptr;
        a.x_bs[s] = s < d->nx ? d->x_bstride[s] : 0;
        a.x_n16[s] = s < d->nx ? (d->x_channels[s] + 15) / 16 : 0;
        if (s < d->nx && (!d->x[s] || d->x_channels[s] <= 0)) return DKT_E_NULL;
        a.nxc += a.x_n16[s];
    }
    a.nx = d->nx;
    a.rh = (char *)d->rh_c8; a.rh_bs = d->rh_c8_bstride;
    a.wzr = (const char *)d->w_zr; a.wq = (const char *)d->w_q;
    a.bz = d->bz; a.br = d->br; a.bq = d->bq;
    a.cz = d->cz; a.cr = d->cr; a.cq = d->cq;
    a.cz_bs = d->cz_bstride; a.cr_bs = d->cr_bstride; a.cq_bs = d->cq_bstride;
    a.h = d->h; a.h_bs = d->h_bstride;
    a.s_zr = d->scale_zr; a.s_q = d->scale_q;
    a.inv_s_zr = 1.0f / d->scale_zr; a.inv_s_q = 1.0f / d->scale_q;
    a.act_scale = d->act_scale;
    a.H = d->H; a.W = d->W; a.Wp = Wp; a.plane_bytes = (long)Hp * Wp * 16;
    a.tiles_w = (d->W + 31) / 32;
    a.tiles_h = (d->H + 7) / 8;
    a.tiles_xy = a.tiles_w * a.tiles_h;
    const long total = (long)a.tiles_xy * d->B;
    if (total > 0x7fffffffL || 16 * 2 * a.plane_bytes > 0x7fffffffL) return DKT_E_SHAPE;
    a.total_tiles = (int)total;
    a.flags = d->flags;
    return DKT_OK;
}

extern "C" long dkt_gru_c8_flag_words(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return DKT_E_SHAPE;
    return (long)((W + 31) / 32) * ((H + 7) / 8) * B;
}

template <int PASSES>
static int g8_launch_p(const G8ArgsPair &ap, bool pair, hipStream_t st);

static int g8_launch(const dkt_gru_c8_desc *d0, const dkt_gru_c8_desc *d1, unsigned *err, hipStream_t st) {
    G8ArgsPair ap;
    int rc = g8_fill(ap.p[0], d0);
    if (rc != DKT_OK) return rc;
    ap.p[1] = ap.p[0];
    if (d1) {
        rc = g8_fill(ap.p[1], d1);
        if (rc != DKT_OK) return rc;
    }
    ap.err = err;
    const int p0 = d0->passes ? d0->passes : 3, p1 = d1 ? (d1->passes ? d1->passes : 3) : p0;
    if (p0 != p1) return DKT_E_UNSUPPORTED;
    if (p0 == 1 && ((ap.p[0].nxc & 1) || (d1 && (ap.p[1].nxc & 1)))) return DKT_E_UNSUPPORTED;     // (two chunks per loop trip)
    switch (p0) {
    case 3: return g8_launch_p<3>(ap, d1 != nullptr, st);
    case 2: return g8_launch_p<2>(ap, d1 != nullptr, st);
    case 1: return g8_launch_p<1>(ap, d1 != nullptr, st);
    default: return DKT_E_UNSUPPORTED;
    }
}

// How the launch's blocks are split between its two problems when the device cannot hold one block per tile (round 5).  A
// block walks its problem's tiles with a fixed stride, and a tile waits for its 3x3 neighbours: when the stride is a multiple
// of the tiles of ONE image, every round of the launch is closed under the neighbour relation (no tile waits for a tile of the
// next round) -- round 4 split the blocks by work alone (243 + 13 at cfg4's 8 pairs per GPU: the rider's 13 blocks took 12
// rounds while the others took 8, and the last tiles of every round waited for the first of the next: 3731 us per launch
// against 8 x 395).  Cost model: rounds x (chunks + a fixed part) per problem; ties go to image-aligned strides.
static void g8_split(const G8ArgsPair &ap, long cap, long &nb0, long &nb1) {
    const long total0 = ap.p[0].total_tiles, total1 = ap.p[1].total_tiles;
    const double c0 = 16 + 2 * ap.p[0].nxc + 10, c1 = 16 + 2 * ap.p[1].nxc + 10;
    double best = 1e300;
    nb0 = 1; nb1 = 1;
    for (long n1 = 1; n1 <= total1 && n1 < cap; ++n1) {
        const long room = cap - n1;
        long cand[2] = {room < total0 ? room : total0, 0};
        const long txy = ap.p[0].tiles_xy;
        cand[1] = (cand[0] >= txy && cand[0] < total0) ? cand[0] / txy * txy : cand[0];
        for (int k = 0; k < 2; ++k) {
            const long n0 = cand[k];
            if (n0 < 1) continue;
            const double r0 = (double)((total0 + n0 - 1) / n0) * c0, r1 = (double)((total1 + n1 - 1) / n1) * c1;
            double cost = r0 > r1 ? r0 : r1;
            const bool aligned0 = n0 >= total0 || n0 % txy == 0;
            const bool aligned1 = n1 >= total1 || n1 % ap.p[1].tiles_xy == 0;
            cost *= 1.0 + (aligned0 ? 0.0 : 0.06) + (aligned1 ? 0.0 : 0.02);        // what the cross-round waits cost, measured
            cost += 1e-6 * (n0 + n1);                                               // (fewer blocks when nothing else decides)
            if (cost < best) { best = cost; nb0 = n0; nb1 = n1; }
        }
    }
}

template <int PASSES>
static int g8_launch_p(const G8ArgsPair &ap, bool pair, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * 23 * 1024 + (size_t)4 * 16384;
    auto kern = gru_c8_kernel<PASSES>;
    static int slots[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!slots[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds) != hipSuccess || per_cu < 1) return DKT_E_UNSUPPORTED;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return DKT_E_UNSUPPORTED;
        slots[dev & 63] = cus;             // one 110 KB block per CU
    }
    const long cap = slots[dev & 63];
    const long total0 = ap.p[0].total_tiles, total1 = pair ? ap.p[1].total_tiles : 0;
    long nb0 = total0 > cap ? cap : total0, nb1 = 0;
    if (nb0 < total0 && nb0 >= ap.p[0].tiles_xy) nb0 = nb0 / ap.p[0].tiles_xy * ap.p[0].tiles_xy;      // whole images per round
    if (pair) {
        nb0 = total0; nb1 = total1;
        if (total0 + total1 > cap) g8_split(ap, cap, nb0, nb1);
    }
    // a block's next tile must not be a neighbour of (or precede a neighbour of) its current one: stride >= tiles_w + 2
    if (nb0 < total0 && nb0 < ap.p[0].tiles_w + 2) return DKT_E_UNSUPPORTED;
    if (pair && nb1 < total1 && nb1 < ap.p[1].tiles_w + 2) return DKT_E_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nb0 + nb1)), dim3(512), lds, st, ap, (int)nb0);
    return dkt_launch_status();
}

extern "C" int dkt_gru_c8(const dkt_gru_c8_desc *d, unsigned *err_word, int device, void *stream) {
    DKT_ENTER(device);
    return g8_launch(d, nullptr, err_word, (hipStream_t)stream);
}

extern "C" int dkt_gru_c8_pair(const dkt_gru_c8_desc *d0, const dkt_gru_c8_desc *d1, unsigned *err_word, int device, void *stream) {
    if (!d1) return DKT_E_NULL;
    DKT_ENTER(device);
    return g8_launch(d0, d1, err_word, (hipStream_t)stream);
}
