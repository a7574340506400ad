// status.hip -- what a checked forward needs to know about the refinement loop, in ONE launch and ONE small copy (round 6).
//
// RAFTStereo.forward / igev_iterate verify three things behind every pair (raft_stereo.py:85-187 has no such check: the
// reference computes in fp32 and has no flag protocol -- these are the conditions under which THIS implementation equals it):
//   * the result is finite (split-fp16 convolutions turn an out-of-range activation into Inf / NaN instead of saturating),
//   * no fused ConvGRU / chain launch gave up waiting for a neighbour tile (gru_c8.hip: the error word),
//   * the C8S tensors whose magnitude follows the input still sit inside the window their scales were picked for.
// Round 5 did that with three host synchronisations and ~10 torch reductions per forward; this kernel folds all of it into
// one pass: maxima as fp16 bit patterns of |hi| (monotone for non-negative values; a NaN pattern is larger than Inf's, so
// neither can be missed), combined with one atomicMax per wave.
#include "dkt_common.h"

struct StatusArgs {
    dkt_c8_range_job job[DKT_STATUS_MAX_JOBS];
    long first[DKT_STATUS_MAX_JOBS + 1];      // prefix sums of the jobs' 16-byte vectors
    int njobs;
    const unsigned *finite_src;
    long finite_n;
    int *err_word;
    unsigned *status;
};

__device__ __forceinline__ unsigned wave_max(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned u = (unsigned)__shfl_xor((int)v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void loop_status_kernel(StatusArgs a) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    if (tid == 0) a.status[0] = a.err_word ? (unsigned)atomicExch(a.err_word, 0) : 0u;
    for (int j = 0; j < a.njobs; j++) {
        const dkt_c8_range_job jb = a.job[j];
        const int Hp = (jb.H + 7) / 8 * 8 + 2, Wp = (jb.W + 31) / 32 * 32 + 2;
        const long plane = (long)Hp * Wp;
        const int G = 2 * ((jb.C + 15) / 16);
        const int c0 = jb.C - jb.tail;                   // first tail channel
        const long n = a.first[j + 1] - a.first[j];
        unsigned body = 0, tail = 0;
        for (long i = tid; i < n; i += nthreads) {
            const long bg = i / plane, rem = i - bg * plane;
            const int b = (int)(bg / G), g = (int)(bg - (long)b * G);
            const uint4 v = *(const uint4 *)((const char *)jb.t + (long)b * jb.bstride_bytes + (((long)g * 2) * plane + rem) * 16);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned lo = w[k] & 0x7fffu, hi = (w[k] >> 16) & 0x7fffu;
                const int c = g * 8 + 2 * k;
                if (c < c0) body = lo > body ? lo : body; else tail = lo > tail ? lo : tail;
                if (c + 1 < c0) body = hi > body ? hi : body; else tail = hi > tail ? hi : tail;
            }
        }
        body = wave_max(body);
        tail = wave_max(tail);
        if ((threadIdx.x & 63) == 0) {
            if (body) atomicMax(&a.status[2 + 2 * j], body);
            if (tail) atomicMax(&a.status[3 + 2 * j], tail);
        }
    }
    unsigned bad = 0;
    for (long i = tid; i < a.finite_n; i += nthreads) bad |= ((a.finite_src[i] & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
    bad = wave_max(bad);
    if (bad && (threadIdx.x & 63) == 0) atomicMax(&a.status[1], 1u);
}

extern "C" int dkt_loop_status(const dkt_c8_range_job *jobs, int njobs, const float *finite_src, long finite_n, int *err_word,
                               unsigned *status, int device, void *stream) {
    if (!status || (njobs > 0 && !jobs) || (finite_n > 0 && !finite_src)) return DKT_E_NULL;
    if (njobs < 0 || njobs > DKT_STATUS_MAX_JOBS || finite_n < 0) return DKT_E_SHAPE;
    DKT_ENTER(device);
    StatusArgs a;
    a.njobs = njobs;
    a.first[0] = 0;
    for (int j = 0; j < njobs; j++) {
        const dkt_c8_range_job &jb = jobs[j];
        if (!jb.t) return DKT_E_NULL;
        if (jb.B < 1 || jb.C < 1 || jb.H < 1 || jb.W < 1 || jb.tail < 0 || jb.tail > jb.C) return DKT_E_SHAPE;
        if (((uintptr_t)jb.t & 15) || (jb.bstride_bytes & 15)) return DKT_E_ALIGN;
        int Hp, Wp;
        dkt_act_c8_dims(jb.H, jb.W, &Hp, &Wp);
        a.job[j] = jb;
        a.first[j + 1] = a.first[j] + (long)jb.B * (2 * ((jb.C + 15) / 16)) * Hp * Wp;
    }
    a.finite_src = (const unsigned *)finite_src;
    a.finite_n = finite_n;
    a.err_word = err_word;
    a.status = status;
    hipError_t e = hipMemsetAsync(status, 0, sizeof(unsigned) * (2 + 2 * (size_t)njobs), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(loop_status_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, a);
    return dkt_launch_status();
}
