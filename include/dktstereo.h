/*
 * dktstereo.h -- C ABI of libdktstereo.so, the MI355X (gfx950) implementation of
 * DKT-Stereo's stereo-inference hot path.
 *
 * The reference (jiaw-z/DKT-Stereo) is pure Python/PyTorch and has no FFI layer
 * of its own; its seam for this path is Python duck typing.  The
 * entry points below are therefore what a binding for that seam needs: one call
 * per reference operator, plain device pointers and sizes, no torch types.  Each
 * declaration cites the reference code it replaces (file:line relative to the
 * reference tree).  dkt_stereo_amd/_ffi.py is the ctypes binding the reference's
 * classes are re-exposed through; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (torch
 *     tensors); the library never allocates, frees or retains caller memory;
 *   - all tensors are float32, NCHW-contiguous unless a stride argument says
 *     otherwise; strides are in ELEMENTS;
 *   - `stream` is a hipStream_t (passed as void* so that C callers need no HIP
 *     headers); work is enqueued stream-ordered, nothing synchronises;
 *   - `device` is the HIP device ordinal the pointers live on, or -1 for "the
 *     calling thread's current device" (nn.DataParallel drives replicas from
 *     one Python thread per GPU, tools/ft_dkt.py:119);
 *   - return 0 on success; negative = argument error detected on the host
 *     before any launch (DKT_E_*); positive = a hipError_t from the launch.
 *     Nothing throws, nothing exits.  dkt_strerror() names either kind;
 *   - stateless and re-entrant: no globals besides read-only tables.
 */
#ifndef DKTSTEREO_H
#define DKTSTEREO_H

#ifdef __cplusplus
extern "C" {
#endif

#define DKT_ABI_VERSION 1
#define DKT_MAX_LEVELS 8

enum {
    DKT_OK = 0,
    DKT_E_NULL = -1,      /* null pointer argument */
    DKT_E_SHAPE = -2,     /* non-positive or inconsistent dimension */
    DKT_E_LEVELS = -3,    /* num_levels outside 1..DKT_MAX_LEVELS or level width reaches 0 */
    DKT_E_RADIUS = -4,    /* radius outside 0..DKT_MAX_RADIUS */
    DKT_E_GROUPS = -5,    /* channels not divisible by groups */
    DKT_E_ALIGN = -6,     /* pointer/stride alignment requirement not met */
    DKT_E_UNSUPPORTED = -7
};
#define DKT_MAX_RADIUS 8

int dkt_version(void);
const char *dkt_strerror(int rc);

/* ---- RAFT-Stereo 1-D correlation --------------------------------------- */

/* All-pairs 1-D correlation fused with the average-pool pyramid.
 * Replaces CorrBlock1D.corr + the pyramid loop of CorrBlock1D.__init__
 * (core/corr.py:148-156 and :111-125; same body meta_arch/raft_stereo/corr.py)
 * and, with divisor = 1, Combined_Geo_Encoding_Volume.corr + its init_corr
 * pyramid (meta_arch/igev_stereo/geometry.py:62-69, :27-29).
 *   pyr[0][n, w2] = (sum_c f1[b,c,h,w1] * f2[b,c,h,w2]) / divisor,  n = (b*H+h)*W1+w1
 *   (divisor = sqrt(C) as the reference divides, corr.py:156)
 *   pyr[i][n, k]  = (pyr[i-1][n,2k] + pyr[i-1][n,2k+1]) * 0.5,  width W2>>i
 * f1: (B,C,H,W1), f2: (B,C,H,W2); pyr: HOST array of L device pointers.
 * Exact fp32 products and fp32 accumulation (v_mfma_f32_32x32x2_f32). */
int dkt_corr1d_build(const float *f1, const float *f2, float *const *pyr,
                     int B, int C, int H, int W1, int W2, int L, float divisor,
                     int device, void *stream);

/* Per-iteration pyramid lookup.  Replaces CorrBlock1D.__call__
 * (core/corr.py:127-146) including its bilinear_sampler/grid_sample round trip
 * (core/utils/utils.py:59-74) and the final permute(0,3,1,2).contiguous().
 *   coords_x: x coordinates, element (b,h,w) at coords_x[b*coords_bstride + h*W1 + w]
 *             (channel 0 of the (B,2,H,W) coords tensor: coords_bstride = 2*H*W1)
 *   out: (B, L*(2r+1), H, W1), channel = level*(2r+1) + tap */
int dkt_corr1d_lookup(const float *const *pyr, const float *coords_x, long coords_bstride,
                      float *out, int B, int H, int W1, int W2, int L, int r,
                      int device, void *stream);

/* Diagonal-major ("skewed") copy of the pyramid and the lookup that reads it:
 *   skew[i][row][s][w1] = pyr[i][row*W1 + w1][(s + (w1 >> i)) mod (W2>>i)],  row = b*H + h
 * with the w1 axis padded to dkt_corr1d_skew_pitch(W1) floats (a multiple of 32: every (row, s)
 * line is 128-byte aligned), i.e. skew[i] holds B*H*(W2>>i)*pitch floats.  Neighbouring pixels with similar disparity then read neighbouring
 * floats: a wave's tap is one contiguous 256-byte load instead of 64 separate lines
 * (corr1d_skew.hip).  dkt_corr1d_lookup_skew returns exactly what dkt_corr1d_lookup does. */
int dkt_corr1d_skew_pitch(int W1);
int dkt_corr1d_skew(const float *const *pyr, float *const *skew, int B, int H, int W1, int W2, int L,
                    int device, void *stream);
int dkt_corr1d_lookup_skew(const float *const *skew, const float *coords_x, long coords_bstride,
                           float *out, int B, int H, int W1, int W2, int L, int r,
                           int device, void *stream);

/* On-the-fly lookup without a volume.  Replaces PytorchAlternateCorrBlock1D
 * (core/corr.py:64-107): samples the (i-times W-pooled) right feature map at
 * the 2r+1 taps and dots it with the left feature vector, / sqrt(C).
 * f2pyr: HOST array of L device pointers to (B,C,H,W2>>i) pooled right maps
 * (built with dkt_pool_w).  coords: (B,2,H,W1) x and y planes. */
int dkt_corr1d_lookup_otf(const float *f1, const float *const *f2pyr, const float *coords,
                          float *out, int B, int C, int H, int W1, int W2, int L, int r,
                          int device, void *stream);

/* avg_pool2d(x,[1,2],stride=[1,2]) on rows: src (rows,W) -> dst (rows,W/2).
 * core/corr.py:104, :124. */
int dkt_pool_w(const float *src, float *dst, long rows, int W, int device, void *stream);

/* L2 normalisation over channels, the prologue of CorrBlock1D_Cosine.corr
 * (core/corr.py:201-202): dst[b,c,p] = src[b,c,p] / ||src[b,:,p]||_2 */
int dkt_l2norm_channels(const float *src, float *dst, int B, int C, long HW,
                        int device, void *stream);

/* ---- up-sampling after the loop ------------------------------------- */

/* RAFTStereo.upsample_flow, meta_arch/raft_stereo/raft_stereo.py:70-82, in one pass:
 * flow (N,D,H,W), mask (N,9*f*f,H,W) -> out (N,D,f*H,f*W);
 * out[n,d,f*h+i,f*w+j] = sum_k softmax_k(mask[n,(k*f+i)*f+j,h,w]) * f*flow[n,d,h+ky-1,w+kx-1]. */
int dkt_convex_upsample(const float *flow, const float *mask, float *out, int N, int D, int H, int W,
                        int factor, int device, void *stream);

/* context_upsample, meta_arch/igev_stereo/submodule.py:242-254: disp_low (B,1,h,w),
 * up_weights (B,9,4h,4w) -> out (B,4h,4w). */
int dkt_context_upsample(const float *disp_low, const float *up_weights, float *out, int B, int h, int w,
                         int device, void *stream);

/* ---- backward of lookup / pyramid (autograd of core/corr.py:119-146) -------- */

/* Gradient of every pyramid level from the gradient of one lookup's output:
 *   grad_pyr[i][n, x0] += g*(1-w),  grad_pyr[i][n, x0+1] += g*w   per tap (zero-padding taps drop out).
 * grad_pyr: HOST array of L device pointers, shaped like the pyramid, ZEROED (or holding the
 * sum of earlier lookups' gradients) by the caller.  coords are detached upstream
 * (raft_stereo.py:152): no coordinate gradient. */
int dkt_corr1d_lookup_bwd(const float *grad_out, const float *coords_x, long coords_bstride,
                          float *const *grad_pyr, int B, int H, int W1, int W2, int L, int r,
                          int device, void *stream);

/* IGEV flavour (autograd of meta_arch/igev_stereo/geometry.py:23-58 w.r.t. the volume and the init
 * correlation; disp is detached upstream, igev_stereo.py:200).  grad_geo[i]: (B,C,D>>i,H,W), grad_init[i]:
 * (B*H*W, W2>>i), both zeroed (or holding earlier lookups' sums) by the caller. */
int dkt_geo_lookup_bwd(const float *grad_out, const float *disp, const float *coords,
                       float *const *grad_geo, float *const *grad_init,
                       int B, int C, int D, int H, int W, int W2, int L, int r, int device, void *stream);
/* grad_vol (BC, D, HW) = T_0 with T_{L-1} = g_{L-1}, T_i[d] = g_i[d] + T_{i+1}[d/2]/2 (pairwise-mean pyramid). */
int dkt_geo_pool_bwd(const float *const *grad_geo, float *grad_vol, long BC, int D, long HW, int L,
                     int device, void *stream);

/* Folds the avg_pool2d backward chain and the 1/sqrt(C) of CorrBlock1D.corr into the gradient
 * of the un-pooled all-pairs product: grad_vol (B*H*W1, W2) = T_0 / divisor with
 * T_{L-1} = g_{L-1}, T_i[c] = g_i[c] + T_{i+1}[c/2]/2.  The two contractions that follow
 * (grad_fmap1 = grad_vol x fmap2, grad_fmap2 = grad_vol^T x fmap1) are plain library GEMMs. */
int dkt_corr1d_pool_bwd(const float *const *grad_pyr, float *grad_vol, int B, int H, int W1, int W2,
                        int L, float divisor, int device, void *stream);

/* ---- PCVNet correlation block / CGI normalised correlation -------- */

/* F.avg_pool2d(x,[1,factor],stride=[1,factor]) on rows: src (rows,W) -> dst (rows,W/factor);
 * the pyramid of meta_arch/pcvnet/corr.py:27-31 (factor 4 when n_downsample == 2, else 2). */
int dkt_pool_rows(const float *src, float *dst, long rows, int W, int factor, int device, void *stream);

/* PCVNet CorrBlock1D.__call__(coords, sigma), meta_arch/pcvnet/corr.py:33-51.
 * pyr: HOST array of L device pointers, level i (B*H*W1, W_i), W_0 = W2, W_{i+1} = W_i/factor.
 * coords, sigma: (B,G,H,W1).  out: (B, L*G*S, H, W1), channel = level*G*S + g*S + s,
 * sampled at (dx_s * sigma + coords) / factor^level, dx = -(S/2)..(S/2) (S odd). */
int dkt_pcv_lookup(const float *const *pyr, const float *coords, const float *sigma, float *out,
                   int B, int G, int H, int W1, int W2, int L, int S, int factor,
                   int device, void *stream);

/* y = x / (||x||_2 over each of G contiguous channel groups + eps): the normalisation inside
 * groupwise_correlation_norm / norm_correlation, meta_arch/cgi/submodule.py:149,168 (eps 1e-5).
 * build_gwc_volume_norm(ref,tgt,D,G) = dkt_gwc_volume on the two normalised maps;
 * build_norm_correlation_volume is G = 1. */
int dkt_group_l2norm(const float *x, float *y, int B, int C, long HW, int G, float eps,
                     int device, void *stream);

/* ---- IGEV combined geometry encoding volume ------------------------------ */

/* Pairwise mean along D of a (B*C, D, HW) volume -> (B*C, D/2, HW): the
 * geometry-volume pyramid of geometry.py:23-25 without the permute copy of :18. */
int dkt_pool_d(const float *src, float *dst, long BC, int D, long HW, int device, void *stream);

/* Replaces Combined_Geo_Encoding_Volume.__call__ (geometry.py:34-58).
 *   geo_pyr[i]: (B,C,D>>i,H,W)  -- the network's native layout, read in place
 *   init_pyr[i]: (B*H*W, W2>>i) -- from dkt_corr1d_build(divisor=1)
 *   disp: (B,1,H,W); coords: (B,H,W,1) x positions (igev_stereo.py:195)
 *   out: (B, L*K*(C+1), H, W), per level [c*K+k for c<C] then [init k] */
int dkt_geo_lookup(const float *const *geo_pyr, const float *const *init_pyr,
                   const float *disp, const float *coords, float *out,
                   int B, int C, int D, int H, int W, int W2, int L, int r,
                   int device, void *stream);

/* ---- cost-volume builders -------------------------------------------------- */

/* Replaces build_gwc_volume + groupwise_correlation
 * (meta_arch/igev_stereo/submodule.py:152-170 == meta_arch/gwcnet/submodules.py:39-58).
 *   vol[b,g,d,h,w] = mean_{c in group g} ref[b,c,h,w]*tgt[b,c,h,w-d] (w>=d) else 0
 * vol_bstride: batch stride of vol in elements (G*D*H*W when dense; larger to
 * write straight into a wider (B,G+2C',D,H,W) buffer, gwc_main.py:315). */
int dkt_gwc_volume(const float *ref, const float *tgt, float *vol,
                   int B, int C, int H, int W, int D, int G, long vol_bstride,
                   int device, void *stream);
/* The same volume as a banded matrix product on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32, gwc_mfma.hip):
 * an fma chain over the group's channels in ascending order -- within one rounding per product of dkt_gwc_volume's
 * (and the reference's) sum of rounded products; D = 48, C/G in {4, 8, 12, 16}, W % 4 == 0, 16-byte aligned volume
 * rows, else DKT_E_UNSUPPORTED (callers fall back to dkt_gwc_volume). */
int dkt_gwc_volume_mfma(const float *ref, const float *tgt, float *vol,
                        int B, int C, int H, int W, int D, int G, long vol_batch_stride,
                        int device, void *stream);

/* Replaces build_concat_volume.  ref_masked = 1: GwcNet semantics
 * (meta_arch/gwcnet/submodules.py:25-36, reference half only where w >= d);
 * ref_masked = 0: IGEV copy (meta_arch/igev_stereo/submodule.py:207-218,
 * reference half for every w).  vol: (B,2C,D,H,W) with batch stride vol_bstride. */
int dkt_concat_volume(const float *ref, const float *tgt, float *vol,
                      int B, int C, int H, int W, int D, int ref_masked, long vol_bstride,
                      int device, void *stream);

/* ---- ConvGRU gate fusions ---------------------------------------------------- */

/* First gate stage of ConvGRU.forward (core/update.py:27-29 ==
 * meta_arch/igev_stereo/update.py:37-39) given the raw outputs of the merged
 * convz|convr convolution:
 *   z  = sigmoid(azr[:, :Ch] + cz);  r = sigmoid(azr[:, Ch:] + cr);  rh = r * h
 * azr: (B,2Ch,HW) dense.  cz, cr, h: (B,Ch,HW) with batch strides (context
 * tensors are split views, raft_stereo.py:114).  z: dense (B,Ch,HW).
 * rh: written with batch stride rh_bstride (straight into the first Ch channels
 * of the [r*h | x] input buffer of convq, replacing the torch.cat of :29). */
int dkt_gru_gate_zr(const float *azr, const float *cz, long cz_bstride,
                    const float *cr, long cr_bstride, const float *h, long h_bstride,
                    float *z, float *rh, long rh_bstride,
                    int B, int Ch, long HW, int device, void *stream);

/* Second gate stage (core/update.py:29-31): q = tanh(aq + cq);
 * h' = (1 - z) * h + z * q.  hout may alias h. */
int dkt_gru_gate_out(const float *aq, const float *cq, long cq_bstride,
                     const float *z, const float *h, long h_bstride,
                     float *hout, long hout_bstride,
                     int B, int Ch, long HW, int device, void *stream);

/* ---- update-operator convolutions ------------------------------------------------ */

/* Correlation lookup fused with the 1x1 convolution that consumes it: core/corr.py:127-146
 * (CorrBlock1D.__call__) followed by relu(convc1(corr)) of BasicMotionEncoder (core/update.py:72,79).
 * One launch; the (B, L*K, H, W1) lookup tensor is never written.  `skew` is the pyramid written by
 * dkt_corr1d_skew; weight_t is convc1.weight viewed (Cout, L*K) and TRANSPOSED to (L*K, Cout), fp32,
 * used as is: the product runs on the exact-fp32 matrix instruction (an fp32 fma chain over the taps).
 *   out[b,co,h,w] = [relu]( bias[co] + sum_k weight[co,k] * lookup[b,k,h,w] )
 * tap (optional, may be NULL): receives lookup[b,k,h,w] itself -- bit-identical to dkt_corr1d_lookup_skew.
 * Supported: Cout <= 64, L in {2,3,4}, r in {3,4}; otherwise DKT_E_UNSUPPORTED (callers then run
 * dkt_corr1d_lookup_skew + dkt_conv2d_f16s). */
int dkt_corr1d_lookup_conv1x1(const float *const *skew, const float *coords_x, long coords_bstride,
                              const float *weight_t, const float *bias, float *out, long out_bstride,
                              float *tap, long tap_bstride,
                              int B, int H, int W1, int W2, int L, int r, int Cout, int relu,
                              int device, void *stream);

/* nn.Conv2d (stride 1, "same" zero padding, 1x1 or 3x3) as used by ConvGRU,
 * BasicMotionEncoder, FlowHead/DispHead and the mask heads (core/update.py:9-10,
 * 19-21, 72-76, 111-113; meta_arch/igev_stereo/update.py same lines), evaluated
 * as an implicit GEMM on the fp16 matrix cores with split operands and fp32
 * accumulation (see conv2d.hip).  `src` are up to DKT_CONV_MAX_SRC NCHW fp32
 * tensors that the reference concatenates along channels before the
 * convolution (torch.cat at core/update.py:24-25,29,83) -- they are read in
 * place.  passes: 3 = w_hi*x_hi + w_lo*x_hi + w_hi*x_lo (fp32-class accuracy),
 * 2 = activations rounded to fp16, 1 = plain fp16.
 *   out[b,co,h,w] = out_scale * acc + bias[co], optional ReLU; (B,Cout,H,W) with
 *   batch stride out_bstride.
 * The weights must have been packed by dkt_conv2d_pack_weights with the SAME
 * src_channels list.  Activations are multiplied by in_scale (a power of two, 1 by default in the
 * Python layer) before they are split into fp16 parts; out_scale = 1 / (weight scale * in_scale).
 * Range: |x * in_scale| < 65520; 22 significant bits for |x * in_scale| >= 2^-3, an absolute
 * floor of 2^-25 below.  Out-of-range, Inf and NaN activations yield non-finite outputs (they are
 * never saturated silently). */
#define DKT_CONV_MAX_SRC 4
long dkt_conv2d_packed_elems(const int *src_channels, int nsrc, int Cout, int KH, int KW);
int dkt_conv2d_pack_weights(const float *w /* (Cout,sum(src_channels),KH,KW) */,
                            const int *src_channels, int nsrc, int Cout, int KH, int KW, float scale,
                            void *w_hi /* fp16[packed_elems] */, void *w_lo, int device, void *stream);
int dkt_conv2d_f16s(const float *const *src, const int *src_channels, const long *src_bstride, int nsrc,
                    const void *w_hi, const void *w_lo, const float *bias, float out_scale, float in_scale,
                    float *out, long out_bstride, int B, int H, int W, int Cout, int KH, int KW,
                    int relu, int passes, int device, void *stream);

/* dkt_conv2d_f16s with a stride (1 or 2) and padding K/2: the down-sampling convolutions of the
 * encoders (core/extractor.py:16,34: 3x3 stride 2 and the 1x1 stride-2 projection).  H, W are the
 * INPUT size; out is (B, Cout, (H-1)/stride+1, (W-1)/stride+1). */
int dkt_conv2d_f16s_strided(const float *const *src, const int *src_channels, const long *src_bstride,
                            int nsrc, const void *w_hi, const void *w_lo, const float *bias,
                            float out_scale, float in_scale, float *out, long out_bstride,
                            int B, int H, int W, int Cout, int KH, int KW, int stride, int relu,
                            int passes, int device, void *stream);

/* ConvGRU with the gate arithmetic fused into the convolution epilogues (core/update.py:23-32):
 * no z|r / q pre-activation tensors ever reach HBM.
 *   gate_zr : merged convz|convr over [h | x...] (2*Ch outputs, packed as one layer):
 *             z = sigmoid(conv_z + cz) -> z;  rh = sigmoid(conv_r + cr) * h -> rh
 *   gate_out: convq over [rh | x...]:  hout = (1 - z)*h + z*tanh(conv_q + cq)   (hout may alias h)
 * Ch must be a multiple of 64 for gate_zr.  Same arithmetic as dkt_gru_gate_zr/_out. */
int dkt_conv2d_f16s_gate_zr(const float *const *src, const int *src_channels, const long *src_bstride, int nsrc,
                            const void *w_hi, const void *w_lo, const float *bias, float out_scale, float in_scale,
                            const float *cz, long cz_bstride, const float *cr, long cr_bstride,
                            const float *h, long h_bstride, float *z, long z_bstride, float *rh, long rh_bstride,
                            int B, int H, int W, int Ch, int KH, int KW, int passes, int device, void *stream);
int dkt_conv2d_f16s_gate_out(const float *const *src, const int *src_channels, const long *src_bstride, int nsrc,
                             const void *w_hi, const void *w_lo, const float *bias, float out_scale, float in_scale,
                             const float *cq, long cq_bstride, const float *z, long z_bstride,
                             const float *h, long h_bstride, float *hout, long hout_bstride,
                             int B, int H, int W, int Ch, int KH, int KW, int passes, int device, void *stream);

/* Two independent convolutions of dkt_conv2d_f16s' kind (stride 1, optional ConvGRU gate epilogue) in ONE
 * launch: the resident blocks of the persistent kernel are split between the two problems in proportion to
 * their work.  Used to fold the coarsest GRU of iteration i+1 (36 tiles at 1/16 KITTI) into the launches of
 * the finest GRU of iteration i (core/update.py:118-127; the two are independent), where it fits in the
 * tile-quantisation slack.  Both problems must have the same KH and fall in the same output-width class
 * (<= 32, <= 64, <= 128, wider), else DKT_E_UNSUPPORTED.
 *   epilogue 0: out = out_scale*acc + bias [ReLU];  Cout = layer outputs
 *   epilogue 1: merged z|r layer, Cout = 2*Ch: z -> out, r*h -> out2 (e0 = cz, e1 = cr)
 *   epilogue 2: q layer, Cout = Ch: (1-z)*h + z*tanh(v + cq) -> out (e0 = cq, e1 = z; out may alias h) */
typedef struct dkt_conv_desc {
    const float *src[DKT_CONV_MAX_SRC];
    long src_bstride[DKT_CONV_MAX_SRC];
    int src_channels[DKT_CONV_MAX_SRC];
    int nsrc;
    const void *w_hi, *w_lo;
    const float *bias;
    float out_scale, in_scale;
    float *out;
    long out_bstride;
    int B, H, W, Cout, KH, KW, relu;
    int epilogue;
    const float *e0; long e0_bstride;
    const float *e1; long e1_bstride;
    const float *h;  long h_bstride;
    float *out2;     long out2_bstride;
    const float *in_norm;   /* optional, dkt_conv2d_f16s_desc only: (B*src_channels[0], 2) = (mean, 1/std) per input
                             * plane; the layer convolves relu((x - mean) * invstd) instead of x (nsrc = 1, 3x3,
                             * 32 < Cout <= 128).  NULL: plain input. */
    int stride;             /* dkt_conv2d_f16s_desc only: 0 / 1 = stride 1, 2 = stride 2 (as dkt_conv2d_f16s_strided) */
    float *stats_ws;        /* optional, dkt_conv2d_f16s_desc with epilogue 0: scratch of dkt_conv2d_stats_ws_floats floats; */
    void *stats_part;       /* ... the instance-norm statistics of the OUTPUT (InstanceNorm2d of core/extractor.py:21-33 over
                             * this layer's result) are accumulated in the epilogue and left in stats_part, a
                             * dkt_instance_norm_workspace(B*Cout, Ho*Wo) buffer in the format dkt_instance_norm_stats writes
                             * (dkt_instance_norm_finalize / _add_relu read it): the statistics pass disappears */
} dkt_conv_desc;
long dkt_conv2d_stats_ws_floats(int B, int Cout, int Ho, int Wo);
int dkt_conv2d_f16s_pair(const dkt_conv_desc *p0, const dkt_conv_desc *p1, int passes, int device, void *stream);
/* One stride-1 convolution given as a descriptor.  Besides the epilogues above:
 *   epilogue 3: residual join of a residual block whose norm is folded into the weights
 *               (core/extractor.py:52-60): out = relu(e0 + [ReLU](out_scale*acc + bias)), e0 shaped like out;
 *   in_norm   : see the field. */
int dkt_conv2d_f16s_desc(const dkt_conv_desc *p, int passes, int device, void *stream);

/* The 7x7 stems (Cin <= 4, stride 1, padding 3: core/update.py:75, igev_stereo/update.py:81,
 * core/extractor.py:136) on the fp16 matrix cores with split operands (stem7.hip): K laid out
 * over (dy, dx, ci) instead of 32-channel chunks.  Weights are pre-packed once per layer
 * (dkt_conv2d_stem7_pack, `scale` a power of two as for dkt_conv2d_pack_weights);
 * out = conv * out_scale + bias [ReLU].  Same fp32-class accuracy as dkt_conv2d_f16s(passes=3). */
long dkt_conv2d_stem7_packed_elems(int Cout);
int dkt_conv2d_stem7_pack(const float *w, int Cout, int Cin, float scale, void *w_hi, void *w_lo,
                          int device, void *stream);
int dkt_conv2d_stem7(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                     const float *bias, float out_scale, float in_scale, float *y, long y_bstride,
                     int B, int Cin, int Cout, int H, int W, int relu, int device, void *stream);

/* Direct exact-fp32 convolution (stride 1, "same" padding) for the two extreme shapes of the
 * update block: 3x3 with Cout <= 4 (flow_head.conv2 / disp_head.conv2, core/update.py:10) and
 * 7x7 with Cin <= 4 (convf1 / convd1, core/update.py:75).  w: (Cout,Cin,K,K); optional ReLU. */
int dkt_conv2d_direct(const float *x, long x_bstride, const float *w, const float *bias,
                      float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                      int KH, int KW, int relu, int device, void *stream);
/* The 3x3 few-output form with y += conv(x) + bias: the disparity update of the refinement loop
 * (raft_stereo.py:165-168: coords1 = coords1 + delta_flow; igev_stereo.py:209) in the head layer's epilogue. */
int dkt_conv2d_direct_accumulate(const float *x, long x_bstride, const float *w, const float *bias,
                                 float *y, long y_bstride, int B, int Cin, int Cout, int H, int W,
                                 int KH, int KW, int device, void *stream);
/* ... and diff_out = y_new - diff_ref on top (planes shaped like y): the flow operand of the next iteration's motion
 * encoder, flow = coords1 - coords0 (raft_stereo.py:147), written by the same epilogue. */
int dkt_conv2d_direct_accumulate_diff(const float *x, long x_bstride, const float *w, const float *bias,
                                      float *y, long y_bstride, const float *diff_ref, long diff_ref_bstride,
                                      float *diff_out, long diff_out_bstride, int B, int Cin, int Cout,
                                      int H, int W, int KH, int KW, int device, void *stream);

/* ---- streaming helpers around the convolutions ---------------------------------------- */

/* pool2x / interp of the update block (core/update.py:87-95): avg_pool2d(x, 3, stride=2,
 * padding=1) and F.interpolate(x, (Ho,Wo), mode="bilinear", align_corners=True) on
 * `planes` = B*C contiguous H x W planes. */
int dkt_pool2x(const float *x, float *y, long planes, int H, int W, int device, void *stream);
int dkt_interp_bilinear(const float *x, float *y, long planes, int H, int W, int Ho, int Wo,
                        int device, void *stream);

/* Encoder glue (core/extractor.py:47-60,176-178; not on the scoped hot path, but inside the
 * timed forward): InstanceNorm2d(affine=False) with optional fused ReLU over `planes` = N*C
 * planes of HW elements (workspace: dkt_instance_norm_workspace() bytes, device memory), and
 * the residual join relu(a + b). */
long dkt_instance_norm_workspace(int planes, long HW);
int dkt_instance_norm(const float *x, float *y, void *workspace, int planes, long HW,
                      float eps, int relu, int device, void *stream);
/* dkt_instance_norm split in two, and the residual-block tail fused (core/extractor.py:52-60):
 * y = relu(a + relu(instance_norm(c))) with the statistics of c from dkt_instance_norm_stats
 * (same workspace, same planes / HW). */
int dkt_instance_norm_stats(const float *x, void *workspace, int planes, long HW, int device, void *stream);
int dkt_instance_norm_add_relu(const float *a, const float *c, float *y, const void *workspace,
                               int planes, long HW, float eps, int device, void *stream);
/* dkt_instance_norm_add_relu whose residual operand `a` is itself still un-normalised (the stem output in front of
 * the first residual block, core/extractor.py:176-178; the projection of a down-sampling block, :36-38):
 * y = relu(a' + relu(instance_norm(c))),  a' = [relu]((a - mean_a) * invstd_a)  with (mean_a, invstd_a) per plane from
 * dkt_instance_norm_finalize -- the normalise pass over `a` is never run. */
int dkt_instance_norm_add_relu_lazy(const float *a, const float *a_mean_invstd /* (planes,2) */, int a_relu,
                                    const float *c, float *y, const void *workspace, int planes, long HW, float eps,
                                    int device, void *stream);
/* The statistics of dkt_instance_norm_stats as (mean, 1/sqrt(var + eps)) float pairs, one per plane: the
 * `in_norm` operand of dkt_conv2d_f16s_desc, which folds relu(instance_norm(x)) between the two 3x3 layers
 * of a residual block (core/extractor.py:46-50) into the second layer's staging. */
int dkt_instance_norm_finalize(const void *workspace, int planes, long HW, float eps, float *mean_invstd /* (planes,2) */,
                               int device, void *stream);

int dkt_add_relu(const float *a, const float *b, float *y, long n, int device, void *stream);

/* ---- round 3: the 3x3 convolution on pre-split activations ("C8S" layout, conv_c8.hip) -------------------------
 * Same operator and arithmetic as dkt_conv2d_f16s(passes = 3) -- core/update.py:19-21,27-31 (ConvGRU), :72-76,84
 * (motion encoder), :9 (FlowHead.conv1) -- but the activation operands arrive as fp16 (hi, lo) pairs written by the
 * producing kernel, so that both operands reach LDS by DMA.
 *
 * C8S tensor of C channels at H x W:  [B][G = 2*ceil(C/16)][2: hi, lo][Hp][Wp][8] fp16 with
 * Hp = roundup(H, 8) + 2, Wp = roundup(W, 32) + 2 (dkt_act_c8_dims); pixel (y, x) lives at (y+1, x+1); the border and the
 * padding channels MUST be zero (allocate zeroed; producers write the interior only).  value = (hi + lo) / scale. */
int dkt_act_c8_dims(int H, int W, int *Hp, int *Wp);
/* fp32 NCHW (B,C,H,W) -> channels [ch0, ch0+C) of a C8S tensor (ch0 a multiple of 8), and back (tests, glue). */
int dkt_act_c8_pack(const float *x, long x_bstride, void *dst, long dst_bstride_bytes, int B, int C, int H, int W,
                    int ch0, float scale, int device, void *stream);
int dkt_act_c8_unpack(const void *src, long src_bstride_bytes, float *y, long y_bstride, int B, int C, int H, int W,
                      int ch0, float scale, int device, void *stream);
/* weights (Cout, sum(src_channels), 3, 3) fp32 -> the kernel's step images [chunk][tap][co/64][hi|lo][k/8][64][8] fp16
 * (every source padded to a multiple of 16 channels, Cout to 64) followed by 12 KB of slack the kernel may read (zero it);
 * `scale` a power of two as for dkt_conv2d_pack_weights. */
long dkt_conv_c8_packed_bytes(const int *src_channels, int nsrc, int Cout);
int dkt_conv_c8_pack_weights(const float *w, const int *src_channels, int nsrc, int Cout, float scale,
                             void *packed, int device, void *stream);
typedef struct dkt_conv_c8_desc {
    const void *src[DKT_CONV_MAX_SRC];      /* C8S operands = the reference's torch.cat list, all H x W */
    long src_bstride[DKT_CONV_MAX_SRC];     /* bytes per batch item */
    int src_channels[DKT_CONV_MAX_SRC];
    int nsrc;
    const void *w;                          /* dkt_conv_c8_pack_weights image */
    const float *bias;
    float out_scale;                        /* 1 / (weight scale * activation scale of the sources) */
    float act_scale;                        /* power of two applied to C8S OUTPUTS before the split */
    int B, H, W, Cout, relu;
    int epilogue;                           /* 0 plain, 1 ConvGRU z|r gates, 2 ConvGRU state update (as dkt_conv_desc),
                                             * 3 flow / disparity head (core/update.py:6-14): relu(conv1(x)) is reduced against conv2's
                                             *   weights per tap instead of being written (cfg 1 or 2); finish with dkt_head_finish
                                             * 4 residual join of a residual block whose norm is folded into the weights
                                             *   (core/extractor.py:52-60): relu(e0 + [relu](conv + bias)), Cout % 4 == 0 */
    float *out; long out_bstride;           /* fp32 NCHW destination (optional when out_c8 is given; epilogue 1: z) */
    void *out_c8; long out_c8_bstride;      /* C8S destination (optional), bytes per batch item; epilogue 2: h' */
    int out_c8_ch0;                         /* first channel written (multiple of 8) */
    const float *e0; long e0_bstride;       /* cz | cq */
    const float *e1; long e1_bstride;       /* cr | z  */
    const float *h;  long h_bstride;
    float *out2; long out2_bstride;         /* epilogue 1: r*h as fp32 NCHW (optional) */
    void *out2_c8; long out2_c8_bstride;    /* epilogue 1: r*h as C8S (optional; same H, W) */
    int out2_c8_ch0;
    const float *tail; long tail_bstride;   /* epilogue 0 + out_c8: channels Cout .. Cout+tail_channels-1 of the C8S output are */
    int tail_channels;                      /* copied from this fp32 NCHW tensor: torch.cat([out, flow]) of core/update.py:85 */
    const float *head_w;                    /* epilogue 3: weights of the head's SECOND 3x3 layer, [outputs][Cout][12] (9 taps + pad) */
    float *head_out; long head_out_bstride; /*   planes [B][(output * blocks + block) * 9 + tap][H][W], blocks = dkt_conv2d_c8_head_blocks */
    int head_outputs;                       /*   1 (stereo: x only / disparity) or 2 */
    int f32_c4;                             /* 1: out, out2, e0, e1, h are "C4" tensors [B][ceil(C/4)][H][W][4] (one 16-byte access per
                                             * lane and channel quad in the epilogue) instead of NCHW; bstrides stay in floats */
    float tail_scale;                       /* power of two applied to the `tail` channels of the C8S output instead of act_scale
                                             * (0 = act_scale): flow / disparity values next to features of another magnitude */
    int passes;                             /* round 5: fp16 MFMA products per weight x activation block. 0 / 3 = w_hi x_hi + w_lo x_hi +
                                             * w_hi x_lo (fp32-class, the parity path); 2 = without w_hi x_lo (activations rounded to
                                             * fp16); 1 = w_hi x_hi only.  Reduced passes serve the precision schedules of the
                                             * refinement loop (the reference's own switch: raft_stereo.py:95,156 `mixed_precision`);
                                             * tile shapes 1..4 */
} dkt_conv_c8_desc;
/* cfg: 0 = tile shape by layer / image size, 1..5 force one (conv_c8.hip c8_dispatch). */
int dkt_conv2d_c8(const dkt_conv_c8_desc *d, int cfg, int device, void *stream);
/* FlowHead.conv2 from epilogue 3's planes, added to `target` (raft_stereo.py:165-168), optionally diff_out = target - diff_ref */
int dkt_conv2d_c8_head_blocks(int Cout, int cfg);
int dkt_head_finish(const float *planes, long planes_bstride, int n_co, const float *bias, float *target,
                    long target_bstride, const float *diff_ref, long diff_ref_bstride, float *diff_out,
                    long diff_out_bstride, int B, int nout, int H, int W, int device, void *stream);
/* two independent convolutions in one launch (the coarsest GRU rides with the finest); cfg != 0 */
int dkt_conv2d_c8_pair(const dkt_conv_c8_desc *d0, const dkt_conv_c8_desc *d1, int cfg, int device, void *stream);
/* Round 5: two DEPENDENT layers in one launch -- stage 0 = d0a [and d0b, an independent
 * second problem of the same size: BasicMotionEncoder's convc2 | convf2, core/update.py:78-79,84-85], stage 1 = d1, which
 * reads what stage 0 writes [encoder.conv, core/update.py:80,85; or ConvGRU's q convolution behind its z|r convolution,
 * core/update.py:27-31].  A stage-1 tile waits for the flags of the stage-0 tiles under its patch instead of a kernel
 * boundary (gru_c8.hip's protocol).  cfg0 / cfg1: tile shapes of the stages, (4, 3) or (4, 4).
 *   flags       : dkt_conv2d_c8_chain_flag_words(d0a, cfg0, 1 or 2) zero-initialised words owned by this pair of layers;
 *   err_word    : optional, bit 1 (value 2) is OR-ed in when a wait timed out (results invalid);
 *   max_blocks  : bound on the launch's blocks (0 = what the device holds).  Every block of a chain launch must stay
 *                 resident while it may wait: chains that can run AT THE SAME TIME on different streams must fit the device
 *                 together (256 each for two chains on MI355X);
 *   timing_only : 1 = no waits, no publishes -- the upper bound of what the fusion can buy; results are WRONG. */
long dkt_conv2d_c8_chain_flag_words(const dkt_conv_c8_desc *d0, int cfg0, int nprob);
int dkt_conv2d_c8_chain(const dkt_conv_c8_desc *d0a, const dkt_conv_c8_desc *d0b, int cfg0, const dkt_conv_c8_desc *d1, int cfg1,
                        unsigned *flags, unsigned *err_word, int max_blocks, int timing_only, int device, void *stream);

/* Round 4: one whole ConvGRU step (core/update.py:23-32 == meta_arch/igev_stereo/update.py:32-41) in ONE launch on C8S
 * operands (csrc/gru_c8.hip):
 *     z = sigmoid(convz([h, x]) + cz);  r = sigmoid(convr([h, x]) + cr);
 *     q = tanh(convq([r*h, x]) + cq);   h <- (1 - z) h + z q          (fp32 NCHW `h` and its C8S twin `h_c8`, both in place)
 * z never leaves the registers; r*h goes through the C8S scratch `rh_c8`; the q convolution of a tile waits for the
 * flags of its 3x3 neighbour tiles instead of a kernel boundary.  hidden must be 128.
 *   w_zr : dkt_conv_c8_pack_weights image of the 256-output layer whose output channel 64*k + 32*m + i is
 *          (m == 0 ? convz : convr) channel 32*k + i, input channels in the reference's order [h | x...];
 *   w_q  : image of convq with its input channels REORDERED to [x... | r*h] (the x chunks are consumed first);
 *   bz, br, bq : the three biases in the reference's channel order;  scale_* : 1 / (weight scale * activation scale);
 *   flags : dkt_gru_c8_flag_words(B, H, W) zero-initialised 32-bit words owned by this (operator, shape) pair -- every
 *           launch increments them, they must not be shared with a launch of another shape or written by the caller;
 *   err_word (optional, device memory): bit 0 (value 1) is OR-ed in if a neighbour wait timed out (results are then invalid).
 * DKT_E_UNSUPPORTED when the device cannot hold the launch's tiles the way the flags need (fall back to two
 * dkt_conv2d_c8 launches with epilogues 1 and 2). */
typedef struct dkt_gru_c8_desc {
    void *h_c8; long h_c8_bstride;              /* C8S hidden state (hidden channels), bytes per batch item */
    const void *x[3]; long x_bstride[3]; int x_channels[3]; int nx;   /* the reference's x_list as C8S tensors */
    void *rh_c8; long rh_c8_bstride;            /* C8S scratch for r*h (zero border, as every C8S tensor) */
    const void *w_zr, *w_q;
    const float *bz, *br, *bq;
    const float *cz, *cr, *cq; long cz_bstride, cr_bstride, cq_bstride;   /* context terms, fp32 NCHW, strides in floats */
    float *h; long h_bstride;                   /* fp32 NCHW hidden state */
    float scale_zr, scale_q, act_scale;
    int B, H, W, hidden;
    unsigned *flags;
    int passes;                                 /* as dkt_conv_c8_desc.passes (0 / 3, 2, 1); both steps of a pair launch alike */
} dkt_gru_c8_desc;
long dkt_gru_c8_flag_words(int B, int H, int W);
int dkt_gru_c8(const dkt_gru_c8_desc *d, unsigned *err_word, int device, void *stream);
/* two independent ConvGRU steps (the finest level and the coarsest one of the next iteration) in one launch */
int dkt_gru_c8_pair(const dkt_gru_c8_desc *d0, const dkt_gru_c8_desc *d1, unsigned *err_word, int device, void *stream);

/* Producers of C8S operands besides the convolution epilogues (same arithmetic as their fp32 twins):
 *   dkt_pool2x_c8 / dkt_interp_c8 : pool2x / interp of core/update.py:87-95, fp32 NCHW in (B,C,H,W);
 *   dkt_conv2d_stem7_c8           : the 7x7 stem (convf1, core/update.py:75);
 *   dkt_corr1d_lookup_conv1x1_c8  : lookup fused with convc1 (core/corr.py:127-146 + core/update.py:76), L = 4 only.
 * ch0 (multiple of 8) = first channel written, act_scale the C8S tensor's power-of-two scale. */
int dkt_pool2x_c8(const float *x, long x_bstride, void *dst, long dst_bstride_bytes, int B, int C, int H, int W,
                  int ch0, float scale, int device, void *stream);
int dkt_interp_c8(const float *x, long x_bstride, void *dst, long dst_bstride_bytes, int B, int C, int H, int W,
                  int Ho, int Wo, int ch0, float scale, int device, void *stream);
/* The motion encoder's front as one launch (round 4): the coordinate update that closes the flow head (what dkt_head_finish
 * does: coords1[:, 0] += conv2(relu(conv1(h)))[:, 0], flow[:, 0] = coords1[:, 0] - coords0[:, 0]; raft_stereo.py:165-168),
 * dkt_corr1d_lookup_conv1x1_c8 at the NEW coordinate (core/corr.py:127-146 + core/update.py:76,84) and dkt_conv2d_stem7_c8 on
 * the NEW flow (core/update.py:77,85).  x_new must be another buffer than x_old (the caller alternates two).  Same arithmetic,
 * bit for bit, as the three launches.  L = 4, r = 3 or 4, cor_channels <= 64. */
typedef struct dkt_motion_front_desc {
    const float *const *skew;                   /* dkt_corr1d_skew pyramids (L pointers) */
    const float *planes; long planes_bstride; int n_co;   /* epilogue-3 planes of the head's first layer, (B, n_co*9, H, W1) */
    const float *head_bias;                     /* bias of the head's second layer (first output) or null */
    const float *x_old; long x_old_bstride;     /* coords1[:, 0] in */
    float *x_new; long x_new_bstride;           /* coords1[:, 0] out */
    const float *x0; long x0_bstride;           /* coords0[:, 0] */
    float *flow; long flow_bstride;             /* (B, stem_cin, H, W1): channel 0 written, the others read */
    const float *w_cor, *b_cor; int cor_channels;   /* convc1: weight k-major (L*K, cor_channels), bias or null */
    void *cor_c8; long cor_c8_bstride_bytes; int cor_c8_ch0; float cor_act_scale;
    const void *stem_w_hi, *stem_w_lo; const float *stem_bias;   /* dkt_conv2d_stem7_pack images of convf1 */
    float stem_out_scale, stem_in_scale; int stem_cin, stem_cout;
    void *flo_c8; long flo_c8_bstride_bytes; int flo_c8_ch0; float flo_act_scale;
    int B, H, W1, W2, L, r;
} dkt_motion_front_desc;
int dkt_motion_front_c8(const dkt_motion_front_desc *d, int device, void *stream);
/* two of the resampling jobs above in one launch (round 4: the loop's pool2x(net[0]) | interp(net[2]) in front of the
 * middle ConvGRU and interp(net[1]) | pool2x(net[1]) behind it, core/update.py:120-132); kind 0 = pool2x (Ho, Wo ignored),
 * kind 1 = interp to (Ho, Wo).  Same arithmetic as the single launches. */
typedef struct dkt_resample_c8_job {
    const float *x; long x_bstride;             /* fp32 NCHW source (B, C, H, W), batch stride in floats */
    void *dst; long dst_bstride_bytes;          /* C8S destination */
    int B, C, H, W, Ho, Wo, ch0, kind;
    float scale;                                /* the destination's power-of-two scale */
} dkt_resample_c8_job;
int dkt_resample_pair_c8(const dkt_resample_c8_job *job0, const dkt_resample_c8_job *job1, int device, void *stream);
int dkt_conv2d_stem7_c8(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                        const float *bias, float out_scale, float in_scale, void *y_c8, long y_c8_bstride_bytes,
                        int y_c8_ch0, float act_scale, int B, int Cin, int Cout, int H, int W, int relu,
                        int device, void *stream);
/* the same stem writing fp32 NCHW and C8S at once (the encoders' first layer: the fp32 copy is the residual operand of
 * layer1, core/extractor.py:167-171) */
int dkt_conv2d_stem7_dual(const float *x, long x_bstride, const void *w_hi, const void *w_lo,
                          const float *bias, float out_scale, float in_scale, float *y, long y_bstride,
                          void *y_c8, long y_c8_bstride_bytes, int y_c8_ch0, float act_scale,
                          int B, int Cin, int Cout, int H, int W, int relu, int device, void *stream);
/* Instance-norm glue of the feature encoder (core/extractor.py:21-60 with norm_fn='instance') producing C8S operands:
 *   t = (c - mean_c) * invstd_c;  if c_relu: t = relu(t);
 *   if a:  t = relu(a' + t),  a' = a or [relu]((a - mean_a) * invstd_a) when a_mean_invstd is given
 * c, a: fp32 NCHW (B, C, H, W), dense per batch item; *_mean_invstd: (B*C, 2) from dkt_instance_norm_finalize.
 * Writes y (fp32 NCHW, optional) and / or channels [ch0, ch0+C) of the C8S tensor dst (optional).  Same arithmetic as
 * dkt_instance_norm / dkt_instance_norm_add_relu_lazy. */
int dkt_instance_norm_join_c8(const float *c, const float *c_mean_invstd, int c_relu,
                              const float *a, const float *a_mean_invstd, int a_relu,
                              float *y, void *dst, long dst_bstride_bytes, int ch0, float act_scale,
                              int B, int C, int H, int W, int device, void *stream);
/* Input normalisation of a stereo pair in one pass (meta_arch/raft_stereo/raft_stereo.py:91-92):
 * out[0:B] = 2*(image1/255) - 1, out[B:2B] = 2*(image2/255) - 1, out = (2B, C, H, W) dense -- the feature encoder's
 * concatenated batch (core/extractor.py:180-183).  image*_bstride / per_image in floats (per_image = C*H*W, dense per item). */
int dkt_normalize_pair(const float *image1, long image1_bstride, const float *image2, long image2_bstride,
                       float *out, int B, long per_image, int device, void *stream);
/* IGEV's geometry-encoding lookup (dkt_geo_lookup; meta_arch/igev_stereo/geometry.py:29-69) fused with the motion
 * encoder's 1x1 layer (convc1, igev_stereo/update.py:78,86): out[b, co] = [relu](bias[co] + sum_k weight_t[k][co] *
 * lookup[b, k]) -- the L*(2r+1)*(C+1)-channel lookup is never written.  weight_t: (L*(2r+1)*(C+1), Cout) k-major, k in the
 * lookup's channel order.  disp: (B,1,H,W) with batch stride disp_bstride (floats); coords: (B,H,W) dense.  Destinations:
 * out (fp32 NCHW) and / or out_c8 (C8S, first channel out_c8_ch0); tap (optional) receives the lookup itself, bit-identical
 * to dkt_geo_lookup.  Supported: L = 2, C = 8, r = 4, Cout <= 64; otherwise DKT_E_UNSUPPORTED. */
int dkt_geo_lookup_conv1x1(const float *const *geo_pyr, const float *const *init_pyr,
                           const float *disp, long disp_bstride, const float *coords,
                           const float *weight_t, const float *bias,
                           float *out, long out_bstride, void *out_c8, long out_c8_bstride_bytes, int out_c8_ch0,
                           float act_scale, float *tap, long tap_bstride,
                           int B, int C, int D, int H, int W, int W2, int L, int r, int Cout, int relu,
                           int device, void *stream);
int dkt_corr1d_lookup_conv1x1_c8(const float *const *skew, const float *coords_x, long coords_bstride,
                                 const float *weight, const float *bias, void *out_c8, long out_c8_bstride_bytes,
                                 int out_c8_ch0, float act_scale, int B, int H, int W1, int W2, int L, int r, int Cout,
                                 int relu, int device, void *stream);

/* ---- round 6: the post-conditions of one pair, in one launch ----------------------------------------------------------
 * The reference's forward (meta_arch/raft_stereo/raft_stereo.py:85-187, igev_stereo.py:192-210) is plain fp32 and needs no
 * check; this implementation equals it only while (i) nothing overflowed the split-fp16 operands, (ii) no fused ConvGRU /
 * chain launch timed out on a neighbour flag (dkt_gru_c8: `err`), (iii) the C8S tensors that follow the input's magnitude
 * still sit in the window their scales were picked for.  dkt_loop_status gathers all three for the caller's ONE host read:
 *   status[0]         = *err_word (0 when err_word is NULL); the word is cleared (atomic exchange)
 *   status[1]         = 1 when finite_src[0 .. finite_n) holds an Inf or a NaN
 *   status[2 + 2 j]   = max over the body channels of C8S tensor j of the fp16 BIT PATTERN of |hi| (0x7c00 = Inf, above = NaN)
 *   status[3 + 2 j]   = the same over its `tail` trailing channels (0 when tail = 0)
 * status: 2 + 2 njobs words, zeroed by the call (stream-ordered).  njobs <= DKT_STATUS_MAX_JOBS. */
#define DKT_STATUS_MAX_JOBS 8
typedef struct dkt_c8_range_job {
    const void *t; long bstride_bytes;          /* C8S tensor (B, C channels, H x W) */
    int B, C, H, W;
    int tail;                                   /* trailing channels reported separately (their own scale) */
} dkt_c8_range_job;
int dkt_loop_status(const dkt_c8_range_job *jobs, int njobs, const float *finite_src, long finite_n, int *err_word,
                    unsigned *status, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DKTSTEREO_H */
