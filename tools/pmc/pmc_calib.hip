// Known-traffic kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM
// section: FETCH_SIZE reads 1/2 of the bytes of a 16-B/lane stream; other widths are uncalibrated).  Built on
// the GPU box:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/pmc/pmc_calib.hip -o /tmp/libpmc_calib.so
#include <hip/hip_runtime.h>
extern "C" {
__global__ void calib_copy4(const float *__restrict__ s, float *__restrict__ d, long n) {     // 4 B/lane load + store
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = s[i];
}
__global__ void calib_copy16(const float4 *__restrict__ s, float4 *__restrict__ d, long n) {  // 16 B/lane
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = s[i];
}
__global__ void calib_read4(const float *__restrict__ s, float *__restrict__ d, long n) {     // read-only, 4 B/lane
    float a = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) a += s[i];
    if (a == 12345.678f) d[0] = a;
}
__global__ void calib_read16(const float4 *__restrict__ s, float *__restrict__ d, long n) {   // read-only, 16 B/lane
    float a = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) { float4 v = s[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.678f) d[0] = a;
}
__global__ void calib_write4(float *__restrict__ d, long n) {                                  // write-only, 4 B/lane
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = 1.0f;
}
int calib_run(int which, const void *s, void *d, long nfloats, void *stream) {
    dim3 g(256 * 16), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
        case 0: hipLaunchKernelGGL(calib_copy4, g, b, 0, st, (const float *)s, (float *)d, nfloats); break;
        case 1: hipLaunchKernelGGL(calib_copy16, g, b, 0, st, (const float4 *)s, (float4 *)d, nfloats / 4); break;
        case 2: hipLaunchKernelGGL(calib_read4, g, b, 0, st, (const float *)s, (float *)d, nfloats); break;
        case 3: hipLaunchKernelGGL(calib_read16, g, b, 0, st, (const float4 *)s, (float *)d, nfloats / 4); break;
        default: hipLaunchKernelGGL(calib_write4, g, b, 0, st, (float *)d, nfloats); break;
    }
    return (int)hipGetLastError();
}
}
