// resample_c8.hip -- pool2x / interp of the update block (core/update.py:87-95) writing the "C8S" operand of the next
// convolution directly (conv_c8.hip): fp32 NCHW in, pre-split fp16 (hi, lo) [B][C/8][2][Hp][Wp][8] out.  Same arithmetic
// and operation order as dkt_pool2x / dkt_interp_bilinear (norm.hip), i.e. ATen's; the split is the one of
// dkt_act_c8_pack.  One thread = one output pixel x 8 channels (one 16-byte store per half).
#include "dkt_common.h"

__device__ __forceinline__ unsigned rs_pack_h2(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; unsigned u; } v;
    v.h[0] = a;
    v.h[1] = b;
    return v.u;
}

__device__ __forceinline__ void rs_store_c8(char *p, long plane, const float (&v)[8], float scale) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float x0 = v[2 * d] * scale, x1 = v[2 * d + 1] * scale;
        const _Float16 a0 = (_Float16)x0, a1 = (_Float16)x1;
        hw[d] = rs_pack_h2(a0, a1);
        lw[d] = rs_pack_h2((_Float16)(x0 - (float)a0), (_Float16)(x1 - (float)a1));
    }
    *(uint4 *)p = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *(uint4 *)(p + plane) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

// F.avg_pool2d(x, 3, stride=2, padding=1) (count_include_pad): sum of the 9 taps in (dy, dx) order, then / 9
__device__ __forceinline__ void pool2x_c8_body(const float *__restrict__ x, long x_bs, char *__restrict__ dst, long dst_bs,
                                               int C, int H, int W, int Ho, int Wo, int Wp, long plane, int ch0, float scale,
                                               int bx, int oy, int bz) {
    const int ox = bx * 128 + threadIdx.x;
    const int g = bz % ((C + 7) / 8), b = bz / ((C + 7) / 8);
    if (ox >= Wo) return;
    const float *xb = x + (long)b * x_bs;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = g * 8 + k;
        const float *p = xb + (long)(c < C ? c : C - 1) * H * W;
        float t[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy - 1 + dy;
            const bool yok = iy >= 0 && iy < H;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox - 1 + dx;
                const bool ok = yok && ix >= 0 && ix < W;
                const float u = p[(long)(yok ? iy : 0) * W + (ix >= 0 && ix < W ? ix : 0)];
                t[dy * 3 + dx] = ok ? u : 0.0f;
            }
        }
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 9; ++i) s = __fadd_rn(s, t[i]);
        v[k] = c < C ? __fdiv_rn(s, 9.0f) : 0.0f;
    }
    rs_store_c8(dst + (long)b * dst_bs + (long)((ch0 >> 3) + g) * 2 * plane + ((long)(oy + 1) * Wp + (ox + 1)) * 16, plane, v, scale);
}

// F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=True) in ATen's order (see interp_kernel, norm.hip)
__device__ __forceinline__ void interp_c8_body(const float *__restrict__ x, long x_bs, char *__restrict__ dst, long dst_bs,
                                               int C, int H, int W, int Ho, int Wo, float sy, float sx, int Wp, long plane,
                                               int ch0, float scale, int bx, int oy, int bz) {
    const int ox = bx * 128 + threadIdx.x;
    const int g = bz % ((C + 7) / 8), b = bz / ((C + 7) / 8);
    if (ox >= Wo) return;
    const float fy = __fmul_rn(sy, (float)oy);
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float ly1 = __fsub_rn(fy, (float)y0), ly0 = __fsub_rn(1.0f, ly1);
    const float fx = __fmul_rn(sx, (float)ox);
    const int x0 = (int)fx;
    const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float lx1 = __fsub_rn(fx, (float)x0), lx0 = __fsub_rn(1.0f, lx1);
    const float *xb = x + (long)b * x_bs;
    float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = g * 8 + k;
        const float *p = xb + (long)(c < C ? c : C - 1) * H * W;
        a0[k] = p[(long)y0 * W + x0]; a1[k] = p[(long)y0 * W + x1];
        b0[k] = p[(long)y1 * W + x0]; b1[k] = p[(long)y1 * W + x1];
    }
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float top = __fadd_rn(__fmul_rn(lx0, a0[k]), __fmul_rn(lx1, a1[k]));
        const float bot = __fadd_rn(__fmul_rn(lx0, b0[k]), __fmul_rn(lx1, b1[k]));
        v[k] = g * 8 + k < C ? __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot)) : 0.0f;
    }
    rs_store_c8(dst + (long)b * dst_bs + (long)((ch0 >> 3) + g) * 2 * plane + ((long)(oy + 1) * Wp + (ox + 1)) * 16, plane, v, scale);
}

__global__ __launch_bounds__(128) void pool2x_c8_kernel(const float *__restrict__ x, long x_bs, char *__restrict__ dst, long dst_bs,
                                                        int C, int H, int W, int Ho, int Wo, int Wp, long plane, int ch0, float scale) {
    pool2x_c8_body(x, x_bs, dst, dst_bs, C, H, W, Ho, Wo, Wp, plane, ch0, scale, blockIdx.x, blockIdx.y, blockIdx.z);
}

__global__ __launch_bounds__(128) void interp_c8_kernel(const float *__restrict__ x, long x_bs, char *__restrict__ dst, long dst_bs,
                                                        int C, int H, int W, int Ho, int Wo, float sy, float sx, int Wp, long plane,
                                                        int ch0, float scale) {
    interp_c8_body(x, x_bs, dst, dst_bs, C, H, W, Ho, Wo, sy, sx, Wp, plane, ch0, scale, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Two independent resampling jobs in ONE launch (the loop's pool2x(net[0]) | interp(net[2]) in front of the middle GRU and
// interp(net[1]) | pool2x(net[1]) behind it, core/update.py:120-132): each is ~10 us of work behind ~5 us of launch gap on
// the forked chain.  Blocks [0, nb0) run job 0, the rest job 1; same arithmetic as the single launches.
struct RsJob {
    const float *x; long x_bs;
    char *dst; long dst_bs;
    int C, H, W, Ho, Wo, Wp; long plane; int ch0; float scale;
    float sy, sx;
    int kind;            // 0: pool2x, 1: interp
    int nbx, nby;        // blocks along x, rows
};
struct RsPair { RsJob j[2]; };

__global__ __launch_bounds__(128) void resample_pair_c8_kernel(RsPair a, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const RsJob &j = a.j[second ? 1 : 0];
    int id = (int)blockIdx.x - (second ? nb0 : 0);
    const int bx = id % j.nbx; id /= j.nbx;
    const int oy = id % j.nby, bz = id / j.nby;
    if (j.kind == 0) pool2x_c8_body(j.x, j.x_bs, j.dst, j.dst_bs, j.C, j.H, j.W, j.Ho, j.Wo, j.Wp, j.plane, j.ch0, j.scale, bx, oy, bz);
    else interp_c8_body(j.x, j.x_bs, j.dst, j.dst_bs, j.C, j.H, j.W, j.Ho, j.Wo, j.sy, j.sx, j.Wp, j.plane, j.ch0, j.scale, bx, oy, bz);
}

static int rs_fill(RsJob &j, const dkt_resample_c8_job *d, long &blocks) {
    if (!d || !d->x || !d->dst) return DKT_E_NULL;
    if (d->B <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || (d->ch0 & 7) || !(d->scale > 0.0f)) return DKT_E_SHAPE;
    if (d->kind != 0 && d->kind != 1) return DKT_E_UNSUPPORTED;
    j.x = d->x; j.x_bs = d->x_bstride; j.dst = (char *)d->dst; j.dst_bs = d->dst_bstride_bytes;
    j.C = d->C; j.H = d->H; j.W = d->W; j.ch0 = d->ch0; j.scale = d->scale; j.kind = d->kind;
    if (d->kind == 0) {
        j.Ho = (d->H - 1) / 2 + 1; j.Wo = (d->W - 1) / 2 + 1;
        j.sy = j.sx = 0.0f;
    } else {
        if (d->Ho <= 0 || d->Wo <= 0) return DKT_E_SHAPE;
        j.Ho = d->Ho; j.Wo = d->Wo;
        j.sy = j.Ho > 1 ? (float)(d->H - 1) / (float)(j.Ho - 1) : 0.0f;
        j.sx = j.Wo > 1 ? (float)(d->W - 1) / (float)(j.Wo - 1) : 0.0f;
    }
    int Hp, Wp;
    dkt_act_c8_dims(j.Ho, j.Wo, &Hp, &Wp);
    j.Wp = Wp; j.plane = (long)Hp * Wp * 16;
    j.nbx = (j.Wo + 127) / 128; j.nby = j.Ho;
    blocks = (long)j.nbx * j.nby * d->B * ((d->C + 7) / 8);
    return DKT_OK;
}

extern "C" int dkt_resample_pair_c8(const dkt_resample_c8_job *job0, const dkt_resample_c8_job *job1, int device, void *stream) {
    RsPair a;
    long n0 = 0, n1 = 0;
    int rc = rs_fill(a.j[0], job0, n0);
    if (rc != DKT_OK) return rc;
    rc = rs_fill(a.j[1], job1, n1);
    if (rc != DKT_OK) return rc;
    if (n0 + n1 > 0x7fffffffL) return DKT_E_SHAPE;
    DKT_ENTER(device);
    hipLaunchKernelGGL(resample_pair_c8_kernel, dim3((unsigned)(n0 + n1)), dim3(128), 0, (hipStream_t)stream, a, (int)n0);
    return dkt_launch_status();
}

extern "C" int dkt_pool2x_c8(const float *x, long x_bstride, void *dst, long dst_bstride_bytes, int B, int C, int H, int W,
                             int ch0, float scale, int device, void *stream) {
    if (!x || !dst) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (ch0 & 7) || !(scale > 0.0f)) return DKT_E_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int Hp, Wp;
    dkt_act_c8_dims(Ho, Wo, &Hp, &Wp);
    const long gz = (long)B * ((C + 7) / 8);
    if (gz > 65535 || Ho > 65535) return DKT_E_SHAPE;
    DKT_ENTER(device);
    hipLaunchKernelGGL(pool2x_c8_kernel, dim3((unsigned)((Wo + 127) / 128), (unsigned)Ho, (unsigned)gz), dim3(128), 0, (hipStream_t)stream,
                       x, x_bstride, (char *)dst, dst_bstride_bytes, C, H, W, Ho, Wo, Wp, (long)Hp * Wp * 16, ch0, scale);
    return dkt_launch_status();
}

extern "C" int dkt_interp_c8(const float *x, long x_bstride, void *dst, long dst_bstride_bytes, int B, int C, int H, int W,
                             int Ho, int Wo, int ch0, float scale, int device, void *stream) {
    if (!x || !dst) return DKT_E_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || (ch0 & 7) || !(scale > 0.0f)) return DKT_E_SHAPE;
    int Hp, Wp;
    dkt_act_c8_dims(Ho, Wo, &Hp, &Wp);
    const long gz = (long)B * ((C + 7) / 8);
    if (gz > 65535 || Ho > 65535) return DKT_E_SHAPE;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.0f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.0f;
    DKT_ENTER(device);
    hipLaunchKernelGGL(interp_c8_kernel, dim3((unsigned)((Wo + 127) / 128), (unsigned)Ho, (unsigned)gz), dim3(128), 0, (hipStream_t)stream,
                       x, x_bstride, (char *)dst, dst_bstride_bytes, C, H, W, Ho, Wo, sy, sx, Wp, (long)Hp * Wp * 16, ch0, scale);
    return dkt_launch_status();
}
