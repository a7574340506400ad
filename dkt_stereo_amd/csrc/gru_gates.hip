// gru_gates.hip -- ConvGRU gate fusions for gfx950.
// Reference: core/update.py:23-32 (== meta_arch/igev_stereo/update.py:33-41).
//
// The reference evaluates each GRU as ~12 elementwise torch kernels plus three
// torch.cat copies.  Here the z|r convolutions are one merged convolution and
// the gate arithmetic is two streaming kernels:
//   gate_zr : z = sigmoid(az+cz), r = sigmoid(ar+cr), rh = r*h   (7 planes of traffic)
//   gate_out: q = tanh(aq+cq), h' = (1-z)*h + z*q                (5 planes)
// rh is stored straight into the first Ch channels of convq's [r*h | x] input
// buffer, so the second torch.cat of the reference never happens.
// HBM-bound: float4 accesses, grid-stride, ~2048 blocks.
#include "dkt_common.h"

__device__ __forceinline__ float dkt_sigmoid(float x) {
    return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x)));
}

struct GateZrArgs {
    const float *azr, *cz, *cr, *h;
    float *z, *rh;
    long cz_bs, cr_bs, h_bs, rh_bs;
    long CHW;  // Ch*HW
    long total4;  // B*CHW/4 (vector path) or B*CHW (scalar path)
};

template <int V>
__global__ __launch_bounds__(256) void gru_gate_zr_kernel(GateZrArgs a) {
    const long per_b = a.CHW / V;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < a.total4; i += (long)gridDim.x * 256L) {
        const long b = i / per_b;
        const long e = (i - b * per_b) * V;
        const float *paz = a.azr + b * 2 * a.CHW + e;
        const float *par = paz + a.CHW;
        const float *pcz = a.cz + b * a.cz_bs + e;
        const float *pcr = a.cr + b * a.cr_bs + e;
        const float *ph = a.h + b * a.h_bs + e;
        float *pz = a.z + b * a.CHW + e;
        float *prh = a.rh + b * a.rh_bs + e;
        if (V == 4) {
            float4 az = *(const float4 *)paz, ar = *(const float4 *)par;
            float4 cz = *(const float4 *)pcz, cr = *(const float4 *)pcr;
            float4 h = *(const float4 *)ph;
            float4 z, rh;
            z.x = dkt_sigmoid(__fadd_rn(az.x, cz.x)); z.y = dkt_sigmoid(__fadd_rn(az.y, cz.y));
            z.z = dkt_sigmoid(__fadd_rn(az.z, cz.z)); z.w = dkt_sigmoid(__fadd_rn(az.w, cz.w));
            rh.x = __fmul_rn(dkt_sigmoid(__fadd_rn(ar.x, cr.x)), h.x);
            rh.y = __fmul_rn(dkt_sigmoid(__fadd_rn(ar.y, cr.y)), h.y);
            rh.z = __fmul_rn(dkt_sigmoid(__fadd_rn(ar.z, cr.z)), h.z);
            rh.w = __fmul_rn(dkt_sigmoid(__fadd_rn(ar.w, cr.w)), h.w);
            *(float4 *)pz = z;
            *(float4 *)prh = rh;
        } else {
            pz[0] = dkt_sigmoid(__fadd_rn(paz[0], pcz[0]));
            prh[0] = __fmul_rn(dkt_sigmoid(__fadd_rn(par[0], pcr[0])), ph[0]);
        }
    }
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

static unsigned gate_blocks(long total) {
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

extern "C" int dkt_gru_gate_zr(const float *azr, const float *cz, long cz_bstride,
                               const float *cr, long cr_bstride, const float *h, long h_bstride,
                               float *z, float *rh, long rh_bstride,
                               int B, int Ch, long HW, int device, void *stream) {
    if (!azr || !cz || !cr || !h || !z || !rh) return DKT_E_NULL;
    if (B <= 0 || Ch <= 0 || HW <= 0) return DKT_E_SHAPE;
    DKT_ENTER(device);
    GateZrArgs a;
    a.azr = azr; a.cz = cz; a.cr = cr; a.h = h; a.z = z; a.rh = rh;
    a.cz_bs = cz_bstride; a.cr_bs = cr_bstride; a.h_bs = h_bstride; a.rh_bs = rh_bstride;
    a.CHW = (long)Ch * HW;
    const bool vec = (a.CHW % 4 == 0) && (cz_bstride % 4 == 0) && (cr_bstride % 4 == 0) &&
                     (h_bstride % 4 == 0) && (rh_bstride % 4 == 0) && aligned16(azr) && aligned16(cz) &&
                     aligned16(cr) && aligned16(h) && aligned16(z) && aligned16(rh);
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        a.total4 = (long)B * a.CHW / 4;
        hipLaunchKernelGGL(gru_gate_zr_kernel<4>, dim3(gate_blocks(a.total4)), dim3(256), 0, st, a);
    } else {
        a.total4 = (long)B * a.CHW;
        hipLaunchKernelGGL(gru_gate_zr_kernel<1>, dim3(gate_blocks(a.total4)), dim3(256), 0, st, a);
    }
    return dkt_launch_status();
}

struct GateOutArgs {
    const float *aq, *cq, *z, *h;
    float *hout;
    long cq_bs, h_bs, hout_bs;
    long CHW;
    long total4;
};

__device__ __forceinline__ float dkt_gru_out(float aq, float cq, float z, float h) {
    const float q = tanhf(__fadd_rn(aq, cq));
    // (1-z)*h + z*q, two rounded products and a rounded sum (core/update.py:31)
    return __fadd_rn(__fmul_rn(__fsub_rn(1.0f, z), h), __fmul_rn(z, q));
}

template <int V>
__global__ __launch_bounds__(256) void gru_gate_out_kernel(GateOutArgs a) {
    const long per_b = a.CHW / V;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < a.total4; i += (long)gridDim.x * 256L) {
        const long b = i / per_b;
        const long e = (i - b * per_b) * V;
        const float *paq = a.aq + b * a.CHW + e;
        const float *pcq = a.cq + b * a.cq_bs + e;
        const float *pz = a.z + b * a.CHW + e;
        const float *ph = a.h + b * a.h_bs + e;
        float *po = a.hout + b * a.hout_bs + e;
        if (V == 4) {
            float4 aq = *(const float4 *)paq, cq = *(const float4 *)pcq;
            float4 z = *(const float4 *)pz, h = *(const float4 *)ph;
            float4 o;
            o.x = dkt_gru_out(aq.x, cq.x, z.x, h.x);
            o.y = dkt_gru_out(aq.y, cq.y, z.y, h.y);
            o.z = dkt_gru_out(aq.z, cq.z, z.z, h.z);
            o.w = dkt_gru_out(aq.w, cq.w, z.w, h.w);
            *(float4 *)po = o;
        } else {
            po[0] = dkt_gru_out(paq[0], pcq[0], pz[0], ph[0]);
        }
    }
}

extern "C" int dkt_gru_gate_out(const float *aq, const float *cq, long cq_bstride,
                                const float *z, const float *h, long h_bstride,
                                float *hout, long hout_bstride,
                                int B, int Ch, long HW, int device, void *stream) {
    if (!aq || !cq || !z || !h || !hout) return DKT_E_NULL;
    if (B <= 0 || Ch <= 0 || HW <= 0) return DKT_E_SHAPE;
    DKT_ENTER(device);
    GateOutArgs a;
    a.aq = aq; a.cq = cq; a.z = z; a.h = h; a.hout = hout;
    a.cq_bs = cq_bstride; a.h_bs = h_bstride; a.hout_bs = hout_bstride;
    a.CHW = (long)Ch * HW;
    const bool vec = (a.CHW % 4 == 0) && (cq_bstride % 4 == 0) && (h_bstride % 4 == 0) &&
                     (hout_bstride % 4 == 0) && aligned16(aq) && aligned16(cq) && aligned16(z) &&
                     aligned16(h) && aligned16(hout);
    hipStream_t st = (hipStream_t)stream;
    if (vec) {
        a.total4 = (long)B * a.CHW / 4;
        hipLaunchKernelGGL(gru_gate_out_kernel<4>, dim3(gate_blocks(a.total4)), dim3(256), 0, st, a);
    } else {
        a.total4 = (long)B * a.CHW;
        hipLaunchKernelGGL(gru_gate_out_kernel<1>, dim3(gate_blocks(a.total4)), dim3(256), 0, st, a);
    }
    return dkt_launch_status();
}
