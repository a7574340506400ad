// dkt_common.h -- shared host/device helpers for libdktstereo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dktstereo.h"

#define DKT_WAVE 64

// RAII-free device scope: switch to `device` (if >=0 and different) for the
// duration of one ABI call, restore on exit.  No global state.
struct DktDeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DktDeviceScope(int device) {
        if (device < 0) return;
        err = hipGetDevice(&prev);
        if (err != hipSuccess) return;
        if (prev != device) {
            err = hipSetDevice(device);
            switched = (err == hipSuccess);
        }
    }
    ~DktDeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
};

#define DKT_ENTER(device)                  \
    DktDeviceScope _scope(device);         \
    if (_scope.err != hipSuccess) return (int)_scope.err

// consume (and return) any launch error without synchronising
static inline int dkt_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DKT_OK : (int)e;
}

struct DktPtrs {
    const float *p[DKT_MAX_LEVELS];
};
struct DktMutPtrs {
    float *p[DKT_MAX_LEVELS];
};

// ---------------------------------------------------------------------------
// The reference's sampler arithmetic, bit for bit
// (core/utils/utils.py:63 + ATen grid_sample, align_corners=True, zero pad):
//   xg = 2*x/(W-1) - 1 ;  ix = (xg + 1) * ((W-1)/2)
//   fl = floor(ix) ; w = ix - fl ; e = 1 - w
//   out = fma(v[fl+1], w, v[fl]*e)
// Explicit _rn intrinsics so that -ffp-contract cannot fuse anything else.
// ---------------------------------------------------------------------------
struct DktTap {
    float fl;  // floor(ix) as float
    float w;   // weight of tap fl+1
    float e;   // weight of tap fl
};

__device__ __forceinline__ DktTap dkt_tap(float x, float wm1, float half_wm1) {
    float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, x), wm1), 1.0f);
    float ix = __fmul_rn(__fadd_rn(xg, 1.0f), half_wm1);
    DktTap t;
    t.fl = floorf(ix);
    t.w = __fsub_rn(ix, t.fl);
    t.e = __fsub_rn(1.0f, t.w);
    return t;
}

// dkt_tap with the IEEE division 2x / (W-1) evaluated as two Newton corrections on the host-rounded
// reciprocal inv = RN(1/(W-1)):  q0 = a*inv;  q1 = q0 + (a - b*q0)*inv (faithful);
// q2 = q1 + (a - b*q1)*inv (correctly rounded, Markstein) -- 5 instructions instead of the ~10 of
// v_div_scale / v_div_fmas / v_div_fixup.  Bit-identical to dkt_tap (tests/test_gpu_parity.py compares
// ~10^8 coordinates per width); callers keep dkt_tap for W == 1 (division by zero).
__device__ __forceinline__ DktTap dkt_tap_rcp(float x, float wm1, float inv, float half_wm1) {
    const float a2 = __fmul_rn(2.0f, x);
    float q = __fmul_rn(a2, inv);
    q = __fmaf_rn(__fmaf_rn(-wm1, q, a2), inv, q);
    q = __fmaf_rn(__fmaf_rn(-wm1, q, a2), inv, q);
    const float xg = __fsub_rn(q, 1.0f);
    const float ix = __fmul_rn(__fadd_rn(xg, 1.0f), half_wm1);
    DktTap t;
    t.fl = floorf(ix);
    t.w = __fsub_rn(ix, t.fl);
    t.e = __fsub_rn(1.0f, t.w);
    return t;
}

__device__ __forceinline__ float dkt_blend(float v0, float v1, const DktTap &t) {
    return __fmaf_rn(v1, t.w, __fmul_rn(v0, t.e));
}

// XCD-aware tile order of the fused ConvGRU launch (round 5; in conv_c8.hip / conv2d.hip the same order measured 0.6 % slower).  The dispatcher deals the blocks of a launch round-robin
// to the part's 8 XCDs, each with an L2 of its own: with tile = block index, a tile's neighbours -- which read the same halo
// rows of every operand -- sit on other XCDs, and every L2 fetches them again.  Of the n blocks that walk one problem, block lb
// (problem-local index; blocks with equal lb & 7 share an XCD whatever the problem's first block is) takes the lb >> 3-th tile of
// the contiguous run of tiles its XCD owns: neighbouring tiles share an L2.  A bijection of [0, n).  Fused ConvGRU launch:
// 391.5 -> 380.1 us (B = 1), 3 125 -> 3 052 us (B = 8), profiles/r05_gru_c8_phases.txt.
__device__ __forceinline__ int dkt_xcd_tile(int lb, int n) {
    const int x = lb & 7, per = n >> 3, rem = n & 7;
    return x * per + (x < rem ? x : rem) + (lb >> 3);
}

// ReLU that propagates NaN like torch.relu (fmaxf(NaN, 0) returns 0 and would turn a diverged
// activation into a plausible-looking zero).
__device__ __forceinline__ float dkt_relu(float v) {
    return v < 0.0f ? 0.0f : v;
}
