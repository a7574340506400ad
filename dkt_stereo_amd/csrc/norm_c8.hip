// norm_c8.hip -- the instance-norm glue of the feature encoder (core/extractor.py:21-60, norm_fn='instance') as ONE
// streaming pass that leaves the next convolution's operand in the "C8S" layout of conv_c8.hip:
//     t = (c - mean_c) * invstd_c            the raw convolution output, statistics from dkt_instance_norm_stats/_finalize
//     t = relu(t)                            (c_relu)
//     t = relu(a' + t)                       (a given: the residual join; a' = a, or [relu]((a - mean_a) * invstd_a) when the
//                                             residual operand is itself a not-yet-normalised tensor)
//     -> y (fp32 NCHW, optional: the next block's residual operand) and / or dst (C8S, optional: the next layer's operand)
// Arithmetic and operation order are those of instnorm_apply_kernel / instnorm_add_relu_kernel (norm.hip); the split is
// dkt_act_c8_pack's.  One thread = V consecutive pixels x 8 channels: V float4-wide plane loads, 16-byte C8S stores of
// 64 contiguous bytes per thread and half.  HBM-bound: 4 B/px/ch in (+4 for a), 4 out (+4 for y).
#include "dkt_common.h"

__device__ __forceinline__ unsigned nc8_pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; unsigned u; } v;
    v.h[0] = a;
    v.h[1] = b;
    return v.u;
}

struct NormC8Args {
    const float *c, *cn, *a, *an;
    float *y;
    char *dst;
    long dst_bs, plane;
    int C, H, W, Wp, ch0, c_relu, a_relu;
    float scale;
    long HW;
};

template <int V>
__global__ __launch_bounds__(256) void instnorm_join_c8_kernel(NormC8Args p) {
    const long q = (blockIdx.x * 256L + threadIdx.x) * V;          // first pixel of this thread
    if (q >= p.HW) return;
    const int ng = (p.C + 7) / 8;
    const int g = blockIdx.y % ng, b = blockIdx.y / ng;
    float v[8][V];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ch = g * 8 + k;
        const bool ok = ch < p.C;
        const long pl = (long)b * p.C + (ok ? ch : p.C - 1);
        const float mean = p.cn[2 * pl], inv = p.cn[2 * pl + 1];
        float cv[V], av[V];
        if constexpr (V == 4) {
            const float4 t = *(const float4 *)(p.c + pl * p.HW + q);
            cv[0] = t.x; cv[1] = t.y; cv[2] = t.z; cv[3] = t.w;
            if (p.a) {
                const float4 u = *(const float4 *)(p.a + pl * p.HW + q);
                av[0] = u.x; av[1] = u.y; av[2] = u.z; av[3] = u.w;
            }
        } else {
            cv[0] = p.c[pl * p.HW + q];
            if (p.a) av[0] = p.a[pl * p.HW + q];
        }
        float am = 0.0f, ai = 1.0f;
        if (p.a && p.an) {
            am = p.an[2 * pl];
            ai = p.an[2 * pl + 1];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t = (cv[j] - mean) * inv;
            if (p.c_relu) t = dkt_relu(t);
            if (p.a) {
                float u = av[j];
                if (p.an) {
                    u = (u - am) * ai;
                    if (p.a_relu) u = dkt_relu(u);
                }
                t = dkt_relu(u + t);
            }
            v[k][j] = ok ? t : 0.0f;
        }
        if (p.y && ok) {
            if constexpr (V == 4) *(float4 *)(p.y + pl * p.HW + q) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
            else p.y[pl * p.HW + q] = v[k][0];
        }
    }
    if (!p.dst) return;
    const int oy = (int)(q / p.W), ox = (int)(q - (long)oy * p.W);      // V pixels of one row (W % V == 0)
    char *d = p.dst + (long)b * p.dst_bs + (long)((p.ch0 >> 3) + g) * 2 * p.plane + ((long)(oy + 1) * p.Wp + (ox + 1)) * 16;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = v[2 * e][j] * p.scale, x1 = v[2 * e + 1][j] * p.scale;
            const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1;
            hw[e] = nc8_pack_h2(a0, a1);
            lw[e] = nc8_pack_h2((_Float16)(x0 - (float)a0), (_Float16)(x1 - (float)a1));
        }
        *(uint4 *)(d + j * 16) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *(uint4 *)(d + p.plane + j * 16) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
}

extern "C" int dkt_instance_norm_join_c8(const float *c, const float *c_mean_invstd, int c_relu,
                                         const float *a, const float *a_mean_invstd, int a_relu,
                                         float *y, void *dst, long dst_bstride_bytes, int ch0, float act_scale,
                                         int B, int C, int H, int W, int device, void *stream) {
    if (!c || !c_mean_invstd || (!y && !dst)) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (ch0 & 7) || !(act_scale > 0.0f)) return DKT_E_SHAPE;
    NormC8Args p;
    p.c = c; p.cn = c_mean_invstd; p.a = a; p.an = a ? a_mean_invstd : nullptr;
    p.y = y; p.dst = (char *)dst; p.dst_bs = dst_bstride_bytes;
    int Hp, Wp;
    dkt_act_c8_dims(H, W, &Hp, &Wp);
    p.plane = (long)Hp * Wp * 16;
    p.C = C; p.H = H; p.W = W; p.Wp = Wp; p.ch0 = ch0; p.c_relu = c_relu ? 1 : 0; p.a_relu = a_relu ? 1 : 0;
    p.scale = act_scale;
    p.HW = (long)H * W;
    const long gy = (long)B * ((C + 7) / 8);
    if (gy > 65535) return DKT_E_SHAPE;
    const bool vec = (W & 3) == 0 && ((((uintptr_t)c) | ((uintptr_t)a) | ((uintptr_t)y)) & 15) == 0;
    DKT_ENTER(device);
    if (vec) {
        const long nb = (p.HW / 4 + 255) / 256;
        hipLaunchKernelGGL(instnorm_join_c8_kernel<4>, dim3((unsigned)nb, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        const long nb = (p.HW + 255) / 256;
        hipLaunchKernelGGL(instnorm_join_c8_kernel<1>, dim3((unsigned)nb, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, p);
    }
    return dkt_launch_status();
}
