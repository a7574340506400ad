// lds_dma_probe.hip -- bounded experiment on the LDS-DMA hazard of DESIGN 3.4 (VERDICT r02 "next round" 1a).
//
// Question: when does a `global_load_lds` (LDS-DMA) copy NOT arrive at `lds_base_of_block + destination offset`?
// Sweep: the victim block's LDS size, the DMA destination offset inside it, and the LDS size of a co-resident block of
// ANOTHER kernel on the same CU (none / 16 / 33 / 64 KB), which moves the victim's allocation base away from 0.
// Every victim block: fills its whole LDS with a sentinel, issues ONE 1-KiB DMA (64 lanes x 16 B) to `dst_off`,
// waits (vmcnt(0) + barrier), then scans its whole LDS: where did the 256 words land, if anywhere?  It also records
// HW_REG_LDS_ALLOC (base / size of its allocation).  The co-resident kernel fills its LDS with its own sentinel,
// spins, and scans for foreign words (a DMA that landed in the neighbour's allocation).
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o tools/_build/lds_dma_probe
// Run on the GPU box: tools/_build/lds_dma_probe > gpurun_out/lds_dma_probe.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned lds[];

constexpr unsigned SENT_V = 0xDEAD0000u, SENT_C = 0xC0C00000u, SRC_TAG = 0x5A000000u;

struct Report {
    unsigned lds_alloc;     // raw HW_REG_LDS_ALLOC
    int landed;             // byte offset (inside this block's LDS) of the first non-sentinel word, -1: none
    int changed;            // number of non-sentinel words
    int exact;              // 1: exactly the 256 expected words at dst_off in lane order
};

// mode 0: destination through M0 (the builtin's LDS pointer argument); mode 1: M0 = 0 .. and the instruction's
// immediate offset (only < 4 KB: offsets above that are not encodable) -- the sweep uses mode 0
__global__ __launch_bounds__(64) void victim(const unsigned *src, Report *rep, int lds_bytes, int dst_off, int use_imm) {
    const int lane = threadIdx.x;
    const int words = lds_bytes / 4;
    for (int i = lane; i < words; i += 64) lds[i] = SENT_V | (unsigned)(i & 0xffff);
    __syncthreads();
    const unsigned *g = src + (size_t)(blockIdx.x & 1023) * 256 + lane * 4;
    if (use_imm)
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)((char *)lds + dst_off - 2048), 16, 2048, 0);
    else
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)((char *)lds + dst_off), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int first = 0x7fffffff, changed = 0, good = 0;
    for (int i = lane; i < words; i += 64) {
        const unsigned v = lds[i];
        if (v != (SENT_V | (unsigned)(i & 0xffff))) {
            ++changed;
            if (i * 4 < first) first = i * 4;
            const int k = i - dst_off / 4;
            if (k >= 0 && k < 256 && v == (SRC_TAG | (unsigned)(((blockIdx.x & 1023) * 256 + k) & 0xffffff))) ++good;
        }
    }
    for (int o = 32; o; o >>= 1) {
        first = min(first, __shfl_xor(first, o));
        changed += __shfl_xor(changed, o);
        good += __shfl_xor(good, o);
    }
    if (lane == 0) {
        Report r;
        r.lds_alloc = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));   // HW_REG_LDS_ALLOC, all 32 bits
        r.landed = first == 0x7fffffff ? -1 : first;
        r.changed = changed;
        r.exact = (good == 256 && changed == 256) ? 1 : 0;
        rep[blockIdx.x] = r;
    }
}

// co-resident: holds `lds_bytes` of LDS for `spin` clocks, then reports foreign words in its LDS
__global__ __launch_bounds__(64) void holder(int *foreign, unsigned *alloc, int lds_bytes, long spin) {
    const int lane = threadIdx.x;
    const int words = lds_bytes / 4;
    for (int i = lane; i < words; i += 64) lds[i] = SENT_C | (unsigned)(i & 0xffff);
    __syncthreads();
    const long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    int bad = 0;
    for (int i = lane; i < words; i += 64) bad += lds[i] != (SENT_C | (unsigned)(i & 0xffff));
    for (int o = 32; o; o >>= 1) bad += __shfl_xor(bad, o);
    if (lane == 0) {
        foreign[blockIdx.x] = bad;
        alloc[blockIdx.x] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
    }
}

int main(int argc, char **argv) {
    const int NV = 2048;            // victim blocks per launch
    const int NH = 256;             // holder blocks (one per CU when the dispatcher spreads them)
    unsigned *src;
    Report *rep;
    int *foreign;
    unsigned *halloc;
    CK(hipMalloc(&src, 1024 * 256 * 4));
    CK(hipMalloc(&rep, NV * sizeof(Report)));
    CK(hipMalloc(&foreign, NH * 4));
    CK(hipMalloc(&halloc, NH * 4));
    std::vector<unsigned> h(1024 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = SRC_TAG | (unsigned)(i & 0xffffff);
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)victim, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)holder, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    const int sizes_kb[] = {16, 32, 48, 64, 72, 80, 96, 112, 123, 127, 128, 136, 144, 160};
    const int holder_kb[] = {0, 16, 33, 64, 96};
    const int use_imm = argc > 1 ? atoi(argv[1]) : 0;
    printf("# victim_KB holder_KB dst_off_KB | blocks base0 base!=0 | bad@base0 bad@base!=0 | landed-dst histogram of bad blocks (bytes:count, 'none' = no word of the block's LDS changed) | holder foreign words | LDS_ALLOC samples (raw hex: bad first)\n");
    for (int hk : holder_kb)
        for (int sk : sizes_kb) {
            if (hk + sk > 160) continue;
            const int S = sk * 1024;
            std::vector<int> offs;
            for (int o = 0; o + 1024 <= S; o += (S <= 32 * 1024 ? 4096 : 8192)) offs.push_back(o);
            offs.push_back(S - 1024);
            if (S > 65536) { offs.push_back(65536 - 1024); offs.push_back(65536); offs.push_back(65536 + 1024); }
            if (S > 131072) { offs.push_back(131072 - 1024); offs.push_back(131072); }
            for (int off : offs) {
                if (use_imm && off < 2048) continue;
                long n0 = 0, n1 = 0, b0 = 0, b1 = 0, fw = 0;
                std::map<std::string, int> hist;
                std::vector<unsigned> sample_bad, sample_ok;
                for (int rep_i = 0; rep_i < 3; ++rep_i) {
                    CK(hipMemsetAsync(rep, 0xff, NV * sizeof(Report), s2));
                    CK(hipStreamSynchronize(s2));
                    if (hk) hipLaunchKernelGGL(holder, dim3(NH), dim3(64), hk * 1024, s1, foreign, halloc, hk * 1024, 400000L);
                    hipLaunchKernelGGL(victim, dim3(NV), dim3(64), S, s2, src, rep, S, off, use_imm);
                    CK(hipDeviceSynchronize());
                    std::vector<Report> r(NV);
                    CK(hipMemcpy(r.data(), rep, NV * sizeof(Report), hipMemcpyDeviceToHost));
                    if (hk) {
                        std::vector<int> f(NH);
                        CK(hipMemcpy(f.data(), foreign, NH * 4, hipMemcpyDeviceToHost));
                        for (int x : f) fw += x;
                    }
                    for (const Report &x : r) {
                        const unsigned base = x.lds_alloc & 0xff;       // LDS_BASE field (gfx9: bits 7:0); raw value printed too
                        const bool bad = !x.exact;
                        if (base == 0) { ++n0; b0 += bad; } else { ++n1; b1 += bad; }
                        if (bad) {
                            char buf[64];
                            if (x.landed < 0) snprintf(buf, sizeof buf, "none");
                            else snprintf(buf, sizeof buf, "%+d(%dw)", x.landed - off, x.changed);
                            ++hist[buf];
                            if (sample_bad.size() < 3) sample_bad.push_back(x.lds_alloc);
                        } else if (sample_ok.size() < 3 && base != 0) sample_ok.push_back(x.lds_alloc);
                    }
                }
                printf("%3d %3d %6.1f | %5ld %5ld %5ld | %5ld %5ld |", sk, hk, off / 1024.0, n0 + n1, n0, n1, b0, b1);
                int shown = 0;
                for (auto &kv : hist) if (shown++ < 6) printf(" %s:%d", kv.first.c_str(), kv.second);
                printf(" | %ld |", fw);
                for (unsigned a : sample_bad) printf(" bad:%08x", a);
                for (unsigned a : sample_ok) printf(" ok:%08x", a);
                printf("\n");
                fflush(stdout);
            }
        }
    return 0;
}
