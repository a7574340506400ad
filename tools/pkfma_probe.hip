// pkfma_probe.hip -- third stage of the DESIGN 3.4 hazard experiment.  Stages 1/2 (lds_dma_probe*.hip) and the instrumented
// product kernel showed the LDS-DMA copies are exact under the failing co-residency; the wrong outputs of
// conv3x3_few_kernel are the LOW halves of its v_pk_fma_f32 accumulator pairs in lanes 48..63, and they vanish when the
// kernel is built without packed fp32 math.  This probe isolates that: waves running chains of v_pk_fma_f32 (register
// operands only, or fed from LDS by ds_read2_b64 / ds_read_b128) against a scalar v_fma_f32 reference, alone and beside
// waves of a second kernel on the same SIMDs that issue MFMAs / hammer LDS.
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/pkfma_probe.hip -o tools/_build/pkfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ __attribute__((aligned(16))) float lds[];

__device__ __forceinline__ f2 pkfma_bcast(f2 x, f2 w, f2 acc) {          // acc.lo += x.lo * w.lo ; acc.hi += x.hi * w.lo
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
}
__device__ __forceinline__ float sfma(float a, float b, float c) {
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
}

struct Rep { int bad_lo, bad_hi, bad_lane_hist[4]; };

// mode 0: operands generated in registers; 1: operands read from LDS (float4 + two scalars per step, the few kernel's
// pattern: the compiler may merge them to ds_read2_b64 / ds_read_b128); the weight is an LDS broadcast read
__global__ __launch_bounds__(512) void victim(Rep *rep, int nit, int mode, int lds_floats, const float *gsrc, int dma) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *my = lds + wave * 1024;
    for (int i = lane; i < 1024; i += 64) my[i] = (float)((i * 37 + wave * 11) % 251) * 0.0078125f - 0.9f;
    __syncthreads();
    f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    float *dmabuf = lds + 8 * 1024 + wave * 2 * 1728;          // wave-private double buffer, as the few kernel's
    float gacc = 0.f;
    for (int k = 0; k < nit; ++k) {
        float v[6], w;
        if (dma && (k & 7) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float *g = gsrc + ((long)(blockIdx.x * 8 + wave) * 4096 + (k >> 3) * 1024) % (1 << 24);
            float *dst = dmabuf + ((k >> 3) & 1) * 1728;
            if (dma == 1) {
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    __builtin_amdgcn_global_load_lds(g + j * 256 + lane * 4, (__attribute__((address_space(3))) void *)(dst + j * 256), 16, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) gacc += g[j * 256 + lane * 4];     // plain loads to VGPRs instead
            }
        }
        if (mode == 0) {
#pragma unroll
            for (int e = 0; e < 6; ++e) v[e] = (float)(((lane + e) * 13 + k * 7) % 97) * 0.015625f - 0.7f;
            w = (float)((k * 29) % 61) * 0.03125f - 0.9f;
        } else {
            const float *pr = my + ((k * 72) % 900) + 4 * (lane & 15) + 4;
            const float4 v4 = *(const float4 *)pr;
            v[0] = pr[-1]; v[1] = v4.x; v[2] = v4.y; v[3] = v4.z; v[4] = v4.w; v[5] = pr[4];
            w = my[(k * 12) % 1000];
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            f2 x01 = {v[dx], v[dx + 1]}, x23 = {v[dx + 2], v[dx + 3]}, ww = {w, w * 0.5f};
            a01 = pkfma_bcast(x01, ww, a01);
            a23 = pkfma_bcast(x23, ww, a23);
            r0 = sfma(v[dx], w, r0); r1 = sfma(v[dx + 1], w, r1); r2 = sfma(v[dx + 2], w, r2); r3 = sfma(v[dx + 3], w, r3);
        }
    }
    if (gacc == 1.2345f) rep->bad_hi = 0x7fffffff;
    const int blo = (__float_as_int(a01.x) != __float_as_int(r0)) + (__float_as_int(a23.x) != __float_as_int(r2));
    const int bhi = (__float_as_int(a01.y) != __float_as_int(r1)) + (__float_as_int(a23.y) != __float_as_int(r3));
    if (blo | bhi) {
        atomicAdd(&rep->bad_lo, blo);
        atomicAdd(&rep->bad_hi, bhi);
        atomicAdd(&rep->bad_lane_hist[lane >> 4], 1);
    }
}

// co-resident kernel: 256 threads.  mode 1: MFMA chain; 2: MFMA chain + LDS write/read traffic; 3: LDS traffic only; 0: sleep
__global__ __launch_bounds__(256, 2) void holder(float *sink, long spin, int mode, int lds_bytes) {
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (tid + i)); b[i] = (_Float16)(0.02f * (i + 1)); }
    const int slots = lds_bytes / 16;
    float4 t = make_float4(tid, 1, 2, 3);
    const long t0 = clock64();
    while (clock64() - t0 < spin) {
        if (mode == 0) __builtin_amdgcn_s_sleep(16);
        if (mode == 1 || mode == 2)
            for (int r = 0; r < 8; ++r)
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        if (mode >= 2 && slots) {
            for (int s = tid; s < slots; s += 256) ((float4 *)lds)[s] = t;
            __syncthreads();
            for (int s = tid; s < slots; s += 256) { const float4 v = ((float4 *)lds)[(s + 17) % slots]; t.x += v.x; t.y += v.y; }
            __syncthreads();
        }
    }
    float s = t.x + t.y;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1.2345f) sink[0] = s;
}

int main() {
    Rep *rep;
    float *sink, *gsrc;
    CK(hipMalloc(&gsrc, ((1L << 24) + 8192) * 4));
    CK(hipMemset(gsrc, 0x3c, ((1L << 24) + 8192) * 4));
    CK(hipMalloc(&rep, sizeof(Rep)));
    CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute((const void *)victim, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)holder, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    printf("# victim_mode(0 reg 1 lds) holder_mode(-1 none 0 sleep 1 mfma 2 mfma+lds 3 lds) | lanes checked | bad lo halves, bad hi halves | bad lanes by quarter 0-15 16-31 32-47 48-63\n");
    for (int dma : {0, 1, 2})
    for (int vm : {0, 1})
        for (int hm : {-1, 0, 1, 2, 3}) {
            Rep tot = {0, 0, {0, 0, 0, 0}};
            long lanes = 0;
            for (int r = 0; r < 6; ++r) {
                CK(hipMemset(rep, 0, sizeof(Rep)));
                if (hm >= 0) hipLaunchKernelGGL(holder, dim3(512), dim3(256), 33 * 1024, s1, sink, 800000L, hm, 33 * 1024);
                hipLaunchKernelGGL(victim, dim3(1024), dim3(512), 120 * 1024, s2, rep, 2000, vm, 30 * 1024, gsrc, dma);
                CK(hipDeviceSynchronize());
                Rep h;
                CK(hipMemcpy(&h, rep, sizeof(Rep), hipMemcpyDeviceToHost));
                tot.bad_lo += h.bad_lo; tot.bad_hi += h.bad_hi;
                for (int q = 0; q < 4; ++q) tot.bad_lane_hist[q] += h.bad_lane_hist[q];
                lanes += 1024L * 512;
            }
            printf("dma%d %d %2d | %9ld | %7d %7d | %6d %6d %6d %6d\n", dma, vm, hm, lanes, tot.bad_lo, tot.bad_hi,
                   tot.bad_lane_hist[0], tot.bad_lane_hist[1], tot.bad_lane_hist[2], tot.bad_lane_hist[3]);
            fflush(stdout);
        }
    return 0;
}
