// lds_dma_probe2.hip -- second stage of the LDS-DMA hazard experiment: lds_dma_probe.hip showed that ONE
// 1-KiB global_load_lds per single-wave block lands exactly where it should for every (LDS size, destination offset,
// allocation base) combination.  This probe reproduces the structure of conv3x3_few_kernel instead: 8 waves per
// block, wave-private double buffers, predicated 16-byte DMA pieces, a long chunk loop with a read-back check of
// every chunk -- beside co-resident blocks of another kernel that (0) sleep, (1) hammer their LDS, (2) stream global
// memory through their LDS, and that are (a) long-lived or (b) short-lived ("churn": blocks start and retire on the
// CU while the victim's copies are in flight).
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe2.hip -o tools/_build/lds_dma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned lds[];

constexpr int NW = 8;
constexpr int PIECES = 6;                 // 1-KiB DMA pieces per chunk and wave
constexpr int BUFW = PIECES * 256 + 192;  // words per buffer (the few kernel's 6912-byte buffers: 1728 words)

struct VRep { unsigned alloc; int bad_words; int bad_chunks; int first_bad_chunk; };

// src: NSRC words, src[i] = tag | i.  Wave w of block b reads chunk k from src + ((b*NW + w)*NIT + k) * PIECES*256 (mod NSRC).
__global__ __launch_bounds__(64 * NW) void victim8(const unsigned *src, long nsrc_words, VRep *rep, int lds_bytes,
                                                   int nit, int pred_mod) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned *mybuf = lds + wave * 2 * BUFW;
    for (int i = lane; i < 2 * BUFW; i += 64) mybuf[i] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // pred_mod > 0: lane l of piece j copies only if (l + j) % pred_mod != 0 (its LDS cell stays zero)
    auto chunk_base = [&](int k) { return ((((long)blockIdx.x * NW + wave) * nit + k) * (PIECES * 256)) % (nsrc_words - PIECES * 256); };
    auto stage = [&](int k, int which) {
        const unsigned *g = src + chunk_base(k);
        unsigned *dst = mybuf + which * BUFW;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const bool on = pred_mod <= 0 || ((lane + j) % pred_mod) != 0;
            if (on) __builtin_amdgcn_global_load_lds(g + j * 256 + lane * 4, (__attribute__((address_space(3))) void *)(dst + j * 256), 16, 0, 0);
        }
    };
    int bad_words = 0, bad_chunks = 0, first_bad = -1;
    stage(0, 0);
    for (int k = 0; k < nit; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (k + 1 < nit) stage(k + 1, (k + 1) & 1);
        const unsigned *b = mybuf + (k & 1) * BUFW;
        const long cb = chunk_base(k);
        int bw = 0;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const bool on = pred_mod <= 0 || ((lane + j) % pred_mod) != 0;
            const uint4 v = *(const uint4 *)(b + j * 256 + lane * 4);
            const unsigned e0 = 0x5A000000u | (unsigned)((cb + j * 256 + lane * 4) & 0xffffff);
            if (on) bw += (v.x != e0) + (v.y != e0 + 1) + (v.z != e0 + 2) + (v.w != e0 + 3);
            else bw += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
        }
        // masked cells must read zero next time too: re-zero what this chunk wrote (as the few kernel never has to,
        // its mask being constant per tile) -- only when the mask moves with k; here it does not, so nothing to do
        for (int o = 32; o; o >>= 1) bw += __shfl_xor(bw, o);
        if (bw) { bad_words += bw; ++bad_chunks; if (first_bad < 0) first_bad = k; }
    }
    __syncthreads();
    // per block: sum over waves through LDS
    if (lane == 0) { lds[wave * 4 + 0] = bad_words; lds[wave * 4 + 1] = bad_chunks; lds[wave * 4 + 2] = first_bad; }
    __syncthreads();
    if (tid == 0) {
        VRep r; r.alloc = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
        r.bad_words = r.bad_chunks = 0; r.first_bad_chunk = -1;
        for (int w = 0; w < NW; ++w) {
            r.bad_words += lds[w * 4]; r.bad_chunks += lds[w * 4 + 1];
            if ((int)lds[w * 4 + 2] >= 0 && r.first_bad_chunk < 0) r.first_bad_chunk = lds[w * 4 + 2];
        }
        rep[blockIdx.x] = r;
    }
}

// co-resident kernel: 256 threads, `lds_bytes` of LDS, runs for `spin` clocks
//   mode 0: sleeps; 1: ds_write_b128 / ds_read_b128 over its LDS; 2: global loads -> LDS writes -> LDS reads (a staging loop)
__global__ __launch_bounds__(256) void holder(const unsigned *src, long nsrc_words, unsigned *sink, int lds_bytes, long spin, int mode) {
    const int tid = threadIdx.x;
    const int slots = lds_bytes / 16;
    uint4 acc = make_uint4(tid, 1, 2, 3);
    const long t0 = clock64();
    long it = 0;
    while (clock64() - t0 < spin) {
        if (mode == 0) __builtin_amdgcn_s_sleep(16);
        else if (mode == 1) {
            for (int s = tid; s < slots; s += 256) ((uint4 *)lds)[s] = acc;
            __syncthreads();
            for (int s = tid; s < slots; s += 256) { const uint4 v = ((uint4 *)lds)[(s + 17) % slots]; acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; }
            __syncthreads();
        } else {
            for (int s = tid; s < slots; s += 256) {
                const uint4 g = *(const uint4 *)(src + (((long)blockIdx.x * 4099 + it * 257 + s) * 4) % (nsrc_words - 4));
                ((uint4 *)lds)[s] = g;
            }
            __syncthreads();
            for (int s = tid; s < slots; s += 256) { const uint4 v = ((uint4 *)lds)[(s + 17) % slots]; acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; }
            __syncthreads();
        }
        ++it;
    }
    if (acc.x == 0x12345678u && acc.y == 42u) sink[0] = acc.z + acc.w;
}

int main() {
    const long NSRC = 64L << 20;            // 256 MB of source words
    unsigned *src, *sink;
    VRep *rep;
    CK(hipMalloc(&src, NSRC * 4));
    CK(hipMalloc(&sink, 64));
    {
        std::vector<unsigned> h(NSRC);
        for (long i = 0; i < NSRC; ++i) h[i] = 0x5A000000u | (unsigned)(i & 0xffffff);
        CK(hipMemcpy(src, h.data(), NSRC * 4, hipMemcpyHostToDevice));
    }
    const int NV = 256;
    CK(hipMalloc(&rep, NV * 4 * sizeof(VRep)));
    CK(hipFuncSetAttribute((const void *)victim8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)holder, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    const int min_lds = NW * 2 * BUFW * 4;     // 110.6 KB, the few kernel's buffer area
    printf("# victim8: %d waves, %d-byte double buffers per wave (min LDS %d B)\n", NW, BUFW * 4, min_lds);
    printf("# victim_KB holder_KB holder_mode(0 sleep 1 lds 2 stage) churn pred | victim blocks, base!=0 | bad blocks (base0 / base!=0) bad chunks bad words first bad chunk | sample LDS_ALLOC of bad\n");
    const int vk_list[] = {111, 123, 127, 160};
    const int hk_list[] = {0, 8, 16, 33};
    for (int pred : {0, 9})
        for (int vk : vk_list)
            for (int hk : hk_list)
                for (int mode : {0, 1, 2})
                    for (int churn : {0, 1}) {
                        if (hk == 0 && (mode || churn)) continue;
                        if (vk + hk > 160) continue;
                        const int S = vk * 1024 < min_lds ? min_lds : vk * 1024;
                        long nb = 0, nb1 = 0, bad0 = 0, bad1 = 0, bc = 0, bw = 0;
                        int fb = -1;
                        unsigned sample = 0;
                        for (int r = 0; r < 4; ++r) {
                            const int nv = NV * (r & 1 ? 2 : 1);
                            CK(hipMemset(rep, 0, nv * sizeof(VRep)));
                            if (hk) {
                                if (churn) hipLaunchKernelGGL(holder, dim3(256 * 40), dim3(256), hk * 1024, s1, src, NSRC, sink, hk * 1024, 8000L, mode);
                                else hipLaunchKernelGGL(holder, dim3(256), dim3(256), hk * 1024, s1, src, NSRC, sink, hk * 1024, 600000L, mode);
                            }
                            hipLaunchKernelGGL(victim8, dim3(nv), dim3(64 * NW), S, s2, src, NSRC, rep, S, 64, pred);
                            CK(hipDeviceSynchronize());
                            std::vector<VRep> h(nv);
                            CK(hipMemcpy(h.data(), rep, nv * sizeof(VRep), hipMemcpyDeviceToHost));
                            for (const VRep &x : h) {
                                const bool b1 = (x.alloc & 0xfff) != 0;
                                ++nb; nb1 += b1;
                                if (x.bad_words) {
                                    (b1 ? bad1 : bad0)++;
                                    bc += x.bad_chunks; bw += x.bad_words;
                                    if (fb < 0) fb = x.first_bad_chunk;
                                    if (!sample) sample = x.alloc;
                                }
                            }
                        }
                        printf("%3d %3d %d %d %d | %5ld %5ld | %5ld %5ld %7ld %9ld %3d | %08x\n", vk, hk, mode, churn, pred, nb, nb1, bad0, bad1, bc, bw, fb, sample);
                        fflush(stdout);
                    }
    return 0;
}
