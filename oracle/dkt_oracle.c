/*
 * dkt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the arithmetic of DKT-Stereo's
 * stereo-inference hot path, used only as the parity checker for the HIP
 * kernels in dkt_stereo_amd/csrc (tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg are the only callers).  Nothing under
 * dkt_stereo_amd/ may link, import or call this file.
 *
 * Pinning: every function here is checked against the reference itself
 * (imported from /root/reference in the build container) by
 * tests/golden/make_golden.py, and against the committed outputs of that
 * script (tests/golden/ *.npz) by tests/test_oracle.py.  The reference has no
 * golden vectors of its own, so reference outputs
 * generated here are the pin.
 *
 * All tensors are float32, NCHW-contiguous unless stated.  Every function
 * cites the reference file:line it restates (paths relative to the reference
 * tree).  Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORC_MAX_LEVELS 8

/* ------------------------------------------------------------------------
 * bilinear_sampler on a 1-row image: core/utils/utils.py:59-74 feeding
 * F.grid_sample(align_corners=True, bilinear, zero padding).
 *   xg = 2*x/(W-1) - 1                     (utils.py:63, python-side fp32)
 *   ix = (xg + 1) * ((W-1)/2)              (ATen unnormalize, align_corners)
 *   x0 = floor(ix); w = ix - x0; e = 1 - w
 *   out = fma(v[x0+1], w, v[x0]*e)         (each tap 0 outside [0,W-1])
 * The y coordinate is 0 and H==1 so the two "south" taps have weight 0.
 * fp32 throughout.  The single fused multiply-add in the last line is what
 * ATen's vectorised CPU kernel executes for `nw_val*nw + ne_val*ne` (its
 * build contracts that expression); with it this function is BIT-IDENTICAL
 * to the reference on 20k random samples incl. out-of-range taps
 * (make_golden.py re-checks that).  Everything else must not be contracted
 * (compile with -ffp-contract=off).
 * ---------------------------------------------------------------------- */
static inline float orc_sample_row(const float *row, int W, float x)
{
    const float wm1 = (float)(W - 1);
    float xg = (2.0f * x) / wm1 - 1.0f;
    float ix = (xg + 1.0f) * (wm1 / 2.0f);
    float fl = floorf(ix);
    float w = ix - fl;
    float e = 1.0f - w;
    /* guard the int conversion for wild coordinates */
    float v0 = 0.0f, v1 = 0.0f;
    if (fl >= 0.0f && fl <= wm1)
        v0 = row[(int)fl];
    if (fl + 1.0f >= 0.0f && fl + 1.0f <= wm1)
        v1 = row[(int)fl + 1];
    return fmaf(v1, w, v0 * e);
}

/* exposed for unit pinning against F.grid_sample */
float orc_bilinear_1d(const float *row, int W, float x)
{
    return orc_sample_row(row, W, x);
}

/* ------------------------------------------------------------------------
 * CorrBlock1D.corr: core/corr.py:148-156 (all-pairs 1-D correlation)
 *   corr[b,h,w1,w2] = sum_c f1[b,c,h,w1]*f2[b,c,h,w2] * scale
 * scale = 1/sqrt(C) for RAFT (corr.py:156), 1 for IGEV (geometry.py:62-69).
 * The reference contracts through BLAS whose summation order is unspecified;
 * this restatement accumulates in double and rounds once, i.e. it is the
 * value both orders approximate (documented tolerance in tests: 2e-5 rel).
 * Note the reference divides by sqrt(C); for C a power of 4 the product with
 * the reciprocal is identical, otherwise pass divide=1 to divide instead.
 * Then CorrBlock1D.__init__: core/corr.py:119-125, pyramid by
 * avg_pool2d([1,2],[1,2]) = (x[2k]+x[2k+1])*0.5, floor on odd widths.
 * pyr[i] has shape (B*H*W1, W2>>i).
 * ---------------------------------------------------------------------- */
void orc_corr1d_build(const float *f1, const float *f2, float *const *pyr,
                      int B, int C, int H, int W1, int W2, int L, float sqrtC_or_0)
{
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int w1 = 0; w1 < W1; ++w1) {
                size_t n = ((size_t)b * H + h) * W1 + w1;
                float *row0 = pyr[0] + n * (size_t)W2;
                for (int w2 = 0; w2 < W2; ++w2) {
                    double acc = 0.0;
                    for (int c = 0; c < C; ++c) {
                        size_t base = ((size_t)b * C + c) * H + h;
                        acc += (double)f1[base * W1 + w1] * (double)f2[base * W2 + w2];
                    }
                    float v = (float)acc;
                    if (sqrtC_or_0 != 0.0f)
                        v = v / sqrtC_or_0;
                    row0[w2] = v;
                }
                int wprev = W2;
                const float *prev = row0;
                for (int i = 1; i < L; ++i) {
                    int wi = wprev / 2;
                    float *row = pyr[i] + n * (size_t)wi;
                    for (int k = 0; k < wi; ++k)
                        row[k] = (prev[2 * k] + prev[2 * k + 1]) * 0.5f;
                    prev = row;
                    wprev = wi;
                }
            }
}

/* pyramid only, from an existing level-0 volume (used to pin pooling exactly) */
void orc_pool_pyramid(const float *lvl0, float *const *pyr_out, size_t N, int W2, int L)
{
    for (size_t n = 0; n < N; ++n) {
        const float *prev = lvl0 + n * (size_t)W2;
        int wprev = W2;
        for (int i = 1; i < L; ++i) {
            int wi = wprev / 2;
            float *row = pyr_out[i - 1] + n * (size_t)wi;
            for (int k = 0; k < wi; ++k)
                row[k] = (prev[2 * k] + prev[2 * k + 1]) * 0.5f;
            prev = row;
            wprev = wi;
        }
    }
}

/* ------------------------------------------------------------------------
 * CorrBlock1D.__call__: core/corr.py:127-146.
 *   coords: (B,2,H,W1), only channel 0 (x) is used (corr.py:129)
 *   for level i, tap k:  x = dx[k] + coords_x / 2^i, dx = linspace(-r,r,2r+1)
 *   out[b, i*K+k, h, w1] = sample(pyr[i][n,:], x)
 * ---------------------------------------------------------------------- */
void orc_corr1d_lookup(const float *const *pyr, const float *coords, float *out,
                       int B, int H, int W1, int W2, int L, int r)
{
    const int K = 2 * r + 1;
    const size_t HW = (size_t)H * W1;
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            size_t n = (size_t)b * HW + p;
            float cx = coords[(size_t)b * 2 * HW + p];
            int wi = W2;
            float div = 1.0f;
            for (int i = 0; i < L; ++i) {
                const float *row = pyr[i] + n * (size_t)wi;
                float xc = cx / div;
                for (int k = 0; k < K; ++k) {
                    float x = (float)(k - r) + xc;
                    out[((size_t)b * L * K + (size_t)i * K + k) * HW + p] =
                        orc_sample_row(row, wi, x);
                }
                wi /= 2;
                div *= 2.0f;
            }
        }
}

/* ------------------------------------------------------------------------
 * PytorchAlternateCorrBlock1D: core/corr.py:64-107 ("alt", on the fly).
 * Per level i the right feature map has been avg-pooled i times along W
 * (corr.py:104); the x coordinate is coords_x/2^i + dx, the y coordinate is
 * the pixel's own row (exactly representable, so the y taps collapse to the
 * row itself with weight 1 -- see note), the sampled C-vector is dotted with
 * fmap1 and divided by sqrt(C) (corr.py:83-87).
 * Note: y goes through 2*y/(H-1)-1 and back, which is not always exact in
 * fp32; the restatement keeps the full 2-D bilinear form for that reason.
 * f2pyr[i]: (B,C,H,W2>>i) pooled right features (built by caller with
 * orc_pool_rows).  Accumulation over C is sequential fp32 like torch.sum over
 * dim=1 on a contiguous tensor is not guaranteed to be; tolerance applies.
 * ---------------------------------------------------------------------- */
void orc_pool_rows(const float *src, float *dst, size_t rows, int W)
{
    int wo = W / 2;
    for (size_t n = 0; n < rows; ++n)
        for (int k = 0; k < wo; ++k)
            dst[n * wo + k] = (src[n * W + 2 * k] + src[n * W + 2 * k + 1]) * 0.5f;
}

void orc_corr1d_lookup_alt(const float *f1, const float *const *f2pyr, const float *coords,
                           float *out, int B, int C, int H, int W1, int W2, int L, int r)
{
    const int K = 2 * r + 1;
    const size_t HW = (size_t)H * W1;
    const float sqrtC = sqrtf((float)C);
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int w1 = 0; w1 < W1; ++w1) {
                size_t p = (size_t)h * W1 + w1;
                float cx = coords[(size_t)b * 2 * HW + p];
                float cy = coords[(size_t)b * 2 * HW + HW + p];
                int wi = W2;
                float div = 1.0f;
                for (int i = 0; i < L; ++i) {
                    const float wm1 = (float)(wi - 1), hm1 = (float)(H - 1);
                    float yg = (2.0f * cy) / hm1 - 1.0f;
                    float iy = (yg + 1.0f) * (hm1 / 2.0f);
                    float fy = floorf(iy);
                    float wy = iy - fy, ey = 1.0f - wy;
                    float xc = cx / div;
                    for (int k = 0; k < K; ++k) {
                        float x = xc + (float)(k - r);
                        float xg = (2.0f * x) / wm1 - 1.0f;
                        float ix = (xg + 1.0f) * (wm1 / 2.0f);
                        float fx = floorf(ix);
                        float wx = ix - fx, ex = 1.0f - wx;
                        float nw = ey * ex, ne = ey * wx, sw = wy * ex, se = wy * wx;
                        int x0ok = (fx >= 0.0f && fx <= wm1), x1ok = (fx + 1.0f >= 0.0f && fx + 1.0f <= wm1);
                        int y0ok = (fy >= 0.0f && fy <= hm1), y1ok = (fy + 1.0f >= 0.0f && fy + 1.0f <= hm1);
                        double acc = 0.0;
                        for (int c = 0; c < C; ++c) {
                            const float *img = f2pyr[i] + ((size_t)b * C + c) * H * (size_t)wi;
                            float vnw = (x0ok && y0ok) ? img[(size_t)(int)fy * wi + (int)fx] : 0.0f;
                            float vne = (x1ok && y0ok) ? img[(size_t)(int)fy * wi + (int)fx + 1] : 0.0f;
                            float vsw = (x0ok && y1ok) ? img[(size_t)((int)fy + 1) * wi + (int)fx] : 0.0f;
                            float vse = (x1ok && y1ok) ? img[(size_t)((int)fy + 1) * wi + (int)fx + 1] : 0.0f;
                            float s = fmaf(vse, se, fmaf(vsw, sw, fmaf(vne, ne, vnw * nw)));
                            acc += (double)(s * f1[(((size_t)b * C + c) * H + h) * W1 + w1]);
                        }
                        out[((size_t)b * L * K + (size_t)i * K + k) * HW + p] = (float)acc / sqrtC;
                    }
                    wi /= 2;
                    div *= 2.0f;
                }
            }
}

/* ------------------------------------------------------------------------
 * Combined_Geo_Encoding_Volume.__call__: meta_arch/igev_stereo/geometry.py:34-58
 *   geo pyramid level i: (N, C, D>>i)  [reference layout after permute, :18]
 *   init pyramid level i: (N, W2>>i)
 *   per level i:
 *     geo taps:  x = dx[k] + disp/2^i                   (geometry.py:42)
 *     init taps: x = (coords/2^i - disp/2^i) + dx[k]    (geometry.py:50)
 *   out channel order per level: [c*K+k for c<C] ++ [init k]   (:55-57)
 *   out: (B, L*K*(C+1), H, W)
 * Here the geo pyramid is passed in the *reference's* (N,C,D_i) layout so
 * that the oracle is layout-for-layout the reference; the product reads the
 * (B,C,D,H,W) tensor directly.
 * ---------------------------------------------------------------------- */
void orc_geo_lookup(const float *const *geo_pyr, const float *const *init_pyr,
                    const float *disp, const float *coords, float *out,
                    int B, int C, int D, int H, int W, int W2, int L, int r)
{
    const int K = 2 * r + 1;
    const size_t HW = (size_t)H * W;
    const int per_level = K * (C + 1);
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            size_t n = (size_t)b * HW + p;
            float d = disp[n];
            float cx = coords[n];
            int di = D, wi = W2;
            float div = 1.0f;
            for (int i = 0; i < L; ++i) {
                float dl = d / div;
                float cl = cx / div;
                for (int c = 0; c < C; ++c) {
                    const float *row = geo_pyr[i] + (n * C + c) * (size_t)di;
                    for (int k = 0; k < K; ++k) {
                        float x = (float)(k - r) + dl;
                        out[((size_t)b * L * per_level + (size_t)i * per_level + (size_t)c * K + k) * HW + p] =
                            orc_sample_row(row, di, x);
                    }
                }
                const float *irow = init_pyr[i] + n * (size_t)wi;
                for (int k = 0; k < K; ++k) {
                    float x = (cl - dl) + (float)(k - r);
                    out[((size_t)b * L * per_level + (size_t)i * per_level + (size_t)C * K + k) * HW + p] =
                        orc_sample_row(irow, wi, x);
                }
                di /= 2;
                wi /= 2;
                div *= 2.0f;
            }
        }
}

/* geo volume (B,C,D,H,W) -> reference pyramid level 0 layout (N,C,D):
 * geometry.py:18  permute(0,3,4,1,2).reshape(b*h*w, c, 1, d) */
void orc_geo_permute(const float *geo, float *out, int B, int C, int D, int H, int W)
{
    const size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (size_t p = 0; p < HW; ++p)
                    out[(((size_t)b * HW + p) * C + c) * D + d] =
                        geo[(((size_t)b * C + c) * D + d) * HW + p];
}

/* ------------------------------------------------------------------------
 * build_gwc_volume + groupwise_correlation:
 *   meta_arch/igev_stereo/submodule.py:152-170 == meta_arch/gwcnet/submodules.py:39-58
 *   vol[b,g,d,h,w] = mean_{c in group g} ref[b,c,h,w]*tgt[b,c,h,w-d]  (w>=d) else 0
 * mean = sequential fp32 sum of the fp32 products divided by cpg.
 * ---------------------------------------------------------------------- */
void orc_gwc_volume(const float *ref, const float *tgt, float *vol,
                    int B, int C, int H, int W, int D, int G)
{
    const int cpg = C / G;
    const size_t HW = (size_t)H * W;
    memset(vol, 0, sizeof(float) * (size_t)B * G * D * HW);
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            for (int d = 0; d < D && d < W; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = d; w < W; ++w) {
                        float s = 0.0f;
                        for (int j = 0; j < cpg; ++j) {
                            size_t ch = ((size_t)b * C + (size_t)g * cpg + j) * HW + (size_t)h * W;
                            s += ref[ch + w] * tgt[ch + w - d];
                        }
                        vol[((((size_t)b * G + g) * D + d) * H + h) * W + w] = s / (float)cpg;
                    }
}

/* ------------------------------------------------------------------------
 * build_concat_volume, two definitions:
 *   ref_masked=1: meta_arch/gwcnet/submodules.py:25-36 (ref half only for w>=d)
 *   ref_masked=0: meta_arch/igev_stereo/submodule.py:207-218 (ref half for all w)
 * target half: tgt[..., w-d] for w>=d else 0.  vol: (B,2C,D,H,W)
 * ---------------------------------------------------------------------- */
void orc_concat_volume(const float *ref, const float *tgt, float *vol,
                       int B, int C, int H, int W, int D, int ref_masked)
{
    const size_t HW = (size_t)H * W;
    memset(vol, 0, sizeof(float) * (size_t)B * 2 * C * D * HW);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        size_t src = (((size_t)b * C + c) * H + h) * W;
                        size_t o_ref = ((((size_t)b * 2 * C + c) * D + d) * H + h) * W + w;
                        size_t o_tgt = ((((size_t)b * 2 * C + C + c) * D + d) * H + h) * W + w;
                        /* d >= W: the python slices [d:] and [:-d] are empty, so w >= d
                         * never holds; the IGEV copy still assigns the whole ref plane */
                        if (w >= d) {
                            vol[o_ref] = ref[src + w];
                            vol[o_tgt] = tgt[src + w - d];
                        } else if (!ref_masked) {
                            vol[o_ref] = ref[src + w];
                        }
                    }
}

/* ------------------------------------------------------------------------
 * ConvGRU gate arithmetic: core/update.py:24-31 (== igev_stereo/update.py:36-40)
 * given the raw conv outputs:
 *   z = sigmoid(az + cz); r = sigmoid(ar + cr); rh = r*h
 *   q = tanh(aq + cq);    h' = (1-z)*h + z*q
 * sigmoid(x) = 1/(1+exp(-x)).  (torch's vectorised sigmoid/tanh are within
 * 2 ulp of these libm forms; the tests carry that tolerance.)
 * ---------------------------------------------------------------------- */
void orc_gru_gate_zr(const float *az, const float *ar, const float *cz, const float *cr,
                     const float *h, float *z, float *rh, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        float zz = 1.0f / (1.0f + expf(-(az[i] + cz[i])));
        float rr = 1.0f / (1.0f + expf(-(ar[i] + cr[i])));
        z[i] = zz;
        rh[i] = rr * h[i];
    }
}

void orc_gru_gate_out(const float *aq, const float *cq, const float *z, const float *h,
                      float *hout, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        float q = tanhf(aq[i] + cq[i]);
        hout[i] = (1.0f - z[i]) * h[i] + z[i] * q;
    }
}

/* ------------------------------------------------------------------------
 * Direct 3x3/KxK convolution, stride 1, zero "same" padding, NCHW, with
 * bias: the arithmetic of nn.Conv2d as used at core/update.py:19-21,72-76.
 * Double accumulation, one rounding (backend summation order unspecified).
 * Small shapes only -- used to pin the product's own conv kernels.
 * ---------------------------------------------------------------------- */
void orc_conv2d_same(const float *x, const float *w, const float *bias, float *y,
                     int B, int Cin, int H, int W, int Cout, int KH, int KW)
{
    const int ph = KH / 2, pw = KW / 2;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int kh = 0; kh < KH; ++kh) {
                            int hh = h + kh - ph;
                            if (hh < 0 || hh >= H)
                                continue;
                            for (int kw = 0; kw < KW; ++kw) {
                                int wx = ww + kw - pw;
                                if (wx < 0 || wx >= W)
                                    continue;
                                acc += (double)x[(((size_t)b * Cin + ci) * H + hh) * W + wx] *
                                       (double)w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw];
                            }
                        }
                    y[(((size_t)b * Cout + co) * H + h) * W + ww] = (float)acc;
                }
}


/* ------------------------------------------------------------------------
 * PCVNet correlation block: meta_arch/pcvnet/corr.py:18-61.
 * Pyramid (corr.py:27-31): level i+1 = F.avg_pool2d(level i, [1,f], stride=[1,f]) with
 * f = compress_factor (4 when n_downsample == 2, else 2): the window is summed left to
 * right and divided by f, floor on widths that f does not divide.  All num_levels levels
 * are used (unlike core/corr.py there is no surplus level).
 * ---------------------------------------------------------------------- */
void orc_pool_rows_f(const float *src, float *dst, size_t rows, int W, int f)
{
    const int wo = W / f;
    for (size_t n = 0; n < rows; ++n)
        for (int k = 0; k < wo; ++k) {
            float s = 0.0f;
            for (int j = 0; j < f; ++j)
                s += src[n * (size_t)W + (size_t)k * f + j];
            dst[n * (size_t)wo + k] = s / (float)f;
        }
}

/* __call__(coords, sigma): corr.py:33-51.  coords, sigma: (B,G,H,W1) (G gaussians per pixel);
 *   x = dx*sigma + coords,  dx = -(S/2)..(S/2)  (torch.range, S samples; product and sum are
 *   two separate roundings);  level i samples at x / f^i;
 *   out[b, i*G*S + g*S + s, h, w1] = sample(pyr[i][n,:], x / f^i)
 * with the sampler of pcvnet/utils/utils.py:59-74 (same function as core/utils/utils.py). */
void orc_pcv_lookup(const float *const *pyr, const float *coords, const float *sigma, float *out,
                    int B, int G, int H, int W1, int W2, int L, int S, int f)
{
    const size_t HW = (size_t)H * W1;
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            const size_t n = (size_t)b * HW + p;
            int wi = W2;
            float div = 1.0f;
            for (int i = 0; i < L; ++i) {
                const float *row = pyr[i] + n * (size_t)wi;
                for (int g = 0; g < G; ++g) {
                    const float c = coords[((size_t)b * G + g) * HW + p];
                    const float sg = sigma[((size_t)b * G + g) * HW + p];
                    for (int s = 0; s < S; ++s) {
                        const float dx = (float)(s - S / 2);
                        const float prod = dx * sg;
                        const float x = prod + c;
                        out[((size_t)b * L * G * S + (size_t)i * G * S + (size_t)g * S + s) * HW + p] =
                            orc_sample_row(row, wi, x / div);
                    }
                }
                wi /= f;
                div *= (float)f;
            }
        }
}

/* ------------------------------------------------------------------------
 * CGI normalised correlation: meta_arch/cgi/submodule.py:143-180
 * (build_norm_correlation_volume is also igev_stereo/submodule.py:179).
 *   xn[b,c,h,w] = x[b,c,h,w] / (||x[b, group(c), h, w]||_2 + 1e-05)
 *   vol[b,g,d,h,w] = mean_{c in g} refn[b,c,h,w] * tgtn[b,c,h,w-d]   (w >= d) else 0
 * norm_correlation is the single-group case.  The norm of a width-sliced tensor equals the
 * slice of the norm (it reduces over channels only), so normalising once up front is the
 * same computation.  Sum of squares: sequential fp32 (torch.norm's order is unspecified).
 * ---------------------------------------------------------------------- */
void orc_group_l2norm(const float *x, float *y, int B, int C, int H, int W, int G)
{
    const int cpg = C / G;
    const size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            for (size_t p = 0; p < HW; ++p) {
                const size_t base = ((size_t)b * C + (size_t)g * cpg) * HW + p;
                float s = 0.0f;
                for (int j = 0; j < cpg; ++j) {
                    const float v = x[base + (size_t)j * HW];
                    s += v * v;
                }
                const float d = sqrtf(s) + 1e-05f;
                for (int j = 0; j < cpg; ++j)
                    y[base + (size_t)j * HW] = x[base + (size_t)j * HW] / d;
            }
}


/* ------------------------------------------------------------------------
 * Backward of the RAFT lookup and pyramid (what autograd derives for
 * core/corr.py:127-146 and :119-125).  coords are detached by the caller
 * (raft_stereo.py:152), so only the gradient w.r.t. the volume exists.
 *
 * orc_corr1d_lookup_bwd: gpyr[i][n, x0] += g*e, gpyr[i][n, x0+1] += g*w for every tap
 * (zero-padding taps contribute nothing), taps visited in k order like
 * grid_sampler_2d_backward visits the sample points.  gpyr must be zeroed by the caller.
 * ---------------------------------------------------------------------- */
void orc_corr1d_lookup_bwd(const float *gout, const float *coords, float *const *gpyr,
                           int B, int H, int W1, int W2, int L, int r)
{
    const int K = 2 * r + 1;
    const size_t HW = (size_t)H * W1;
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            const size_t n = (size_t)b * HW + p;
            const float cx = coords[(size_t)b * 2 * HW + p];
            int wi = W2;
            float div = 1.0f;
            for (int i = 0; i < L; ++i) {
                float *row = gpyr[i] + n * (size_t)wi;
                const float wm1 = (float)(wi - 1);
                const float xc = cx / div;
                for (int k = 0; k < K; ++k) {
                    const float x = (float)(k - r) + xc;
                    const float xg = (2.0f * x) / wm1 - 1.0f;
                    const float ix = (xg + 1.0f) * (wm1 / 2.0f);
                    const float fl = floorf(ix);
                    const float w = ix - fl, e = 1.0f - w;
                    const float g = gout[((size_t)b * L * K + (size_t)i * K + k) * HW + p];
                    if (fl >= 0.0f && fl <= wm1) row[(int)fl] += g * e;
                    if (fl + 1.0f >= 0.0f && fl + 1.0f <= wm1) row[(int)fl + 1] += g * w;
                }
                wi /= 2;
                div *= 2.0f;
            }
        }
}

/* Total gradient of the level-0 volume from the per-level gradients, i.e. the chain of
 * avg_pool2d backward passes (each hands grad/2 to both inputs; the odd last column of a
 * level receives nothing):  T_{L-1} = g_{L-1};  T_i[c] = g_i[c] + T_{i+1}[c/2] / 2;
 * g0_total = T_0 / divisor   (divisor = sqrt(C): the backward of corr / sqrt(C)). */
void orc_corr1d_pool_bwd(const float *const *gpyr, float *g0, size_t N, int W2, int L, float divisor)
{
    for (size_t n = 0; n < N; ++n)
        for (int c = 0; c < W2; ++c) {
            int deepest = 0;
            for (int i = 1; i < L; ++i) {
                if ((c >> i) < (W2 >> i)) deepest = i; else break;
            }
            float t = gpyr[deepest][n * (size_t)(W2 >> deepest) + (c >> deepest)];
            for (int i = deepest - 1; i >= 0; --i)
                t = gpyr[i][n * (size_t)(W2 >> i) + (c >> i)] + t / 2.0f;
            g0[n * (size_t)W2 + c] = t / divisor;
        }
}


/* ------------------------------------------------------------------------
 * Convex up-sampling, RAFTStereo.upsample_flow: meta_arch/raft_stereo/raft_stereo.py:70-82
 *   mask (N, 9*f*f, H, W) viewed (N,1,9,f,f,H,W), softmax over the 9; flow (N,D,H,W)
 *   out[n,d,f*h+i,f*w+j] = sum_k softmax_k(mask[n,(k*f+i)*f+j,h,w]) * f*flow[n,d,h+ky-1,w+kx-1]
 * (k = 3*ky+kx, zero outside: F.unfold padding).  softmax = exp(x - max) / sum, sums in k order.
 * ---------------------------------------------------------------------- */
void orc_convex_upsample(const float *flow, const float *mask, float *out, int N, int D, int H, int W, int f)
{
    const size_t HW = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                for (int i = 0; i < f; ++i)
                    for (int j = 0; j < f; ++j) {
                        float m[9], mx = -INFINITY, sum = 0.0f;
                        for (int k = 0; k < 9; ++k) {
                            m[k] = mask[((size_t)n * 9 * f * f + ((size_t)k * f + i) * f + j) * HW + (size_t)h * W + w];
                            if (m[k] > mx) mx = m[k];
                        }
                        for (int k = 0; k < 9; ++k) {
                            m[k] = expf(m[k] - mx);
                            sum += m[k];
                        }
                        for (int d = 0; d < D; ++d) {
                            float acc = 0.0f;
                            for (int k = 0; k < 9; ++k) {
                                const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
                                float v = 0.0f;
                                if (hh >= 0 && hh < H && ww >= 0 && ww < W)
                                    v = (float)f * flow[((size_t)n * D + d) * HW + (size_t)hh * W + ww];
                                acc += (m[k] / sum) * v;
                            }
                            out[(((size_t)n * D + d) * H * f + (size_t)h * f + i) * (size_t)W * f + (size_t)w * f + j] = acc;
                        }
                    }
}

/* context_upsample: meta_arch/igev_stereo/submodule.py:242-254
 *   disp_low (B,1,h,w), up_weights (B,9,4h,4w) -> (B,4h,4w):
 *   out[b,Y,X] = sum_k unfold3x3(disp_low)[k, Y/4, X/4] * up_weights[b,k,Y,X]  (nearest up-sampling by 4) */
void orc_context_upsample(const float *disp, const float *wts, float *out, int B, int h, int w)
{
    const int H4 = 4 * h, W4 = 4 * w;
    for (int b = 0; b < B; ++b)
        for (int Y = 0; Y < H4; ++Y)
            for (int X = 0; X < W4; ++X) {
                float acc = 0.0f;
                for (int k = 0; k < 9; ++k) {
                    const int hh = Y / 4 + k / 3 - 1, ww = X / 4 + k % 3 - 1;
                    float v = 0.0f;
                    if (hh >= 0 && hh < h && ww >= 0 && ww < w) v = disp[((size_t)b * h + hh) * w + ww];
                    acc += v * wts[(((size_t)b * 9 + k) * H4 + Y) * W4 + X];
                }
                out[((size_t)b * H4 + Y) * W4 + X] = acc;
            }
}


/* Backward of the IGEV geometry-volume lookup (autograd of geometry.py:34-58 w.r.t. the two pyramids;
 * disp is detached by the caller, igev_stereo.py:200).  Reference layouts: ggeo[i] rows (n*C + c, D>>i),
 * ginit[i] rows (n, W2>>i); both zeroed by the caller.  Taps in k order, like orc_corr1d_lookup_bwd. */
void orc_geo_lookup_bwd(const float *gout, const float *disp, const float *coords,
                        float *const *ggeo, float *const *ginit,
                        int B, int C, int D, int H, int W, int W2, int L, int r)
{
    const int K = 2 * r + 1;
    const size_t HW = (size_t)H * W;
    const int per_level = K * (C + 1);
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            const size_t n = (size_t)b * HW + p;
            const float d = disp[n], cx = coords[n];
            int di = D, wi = W2;
            float div = 1.0f;
            for (int i = 0; i < L; ++i) {
                const float dl = d / div, cl = cx / div;
                for (int c = 0; c <= C; ++c) {
                    const int width = c < C ? di : wi;
                    float *row = c < C ? ggeo[i] + (n * C + c) * (size_t)di : ginit[i] + n * (size_t)wi;
                    const float wm1 = (float)(width - 1);
                    for (int k = 0; k < K; ++k) {
                        const float x = c < C ? (float)(k - r) + dl : (cl - dl) + (float)(k - r);
                        const float xg = (2.0f * x) / wm1 - 1.0f;
                        const float ix = (xg + 1.0f) * (wm1 / 2.0f);
                        const float fl = floorf(ix);
                        const float w = ix - fl, e = 1.0f - w;
                        const float g = gout[((size_t)b * L * per_level + (size_t)i * per_level + (size_t)c * K + k) * HW + p];
                        if (fl >= 0.0f && fl <= wm1) row[(int)fl] += g * e;
                        if (fl + 1.0f >= 0.0f && fl + 1.0f <= wm1) row[(int)fl + 1] += g * w;
                    }
                }
                di /= 2;
                wi /= 2;
                div *= 2.0f;
            }
        }
}

int orc_version(void) { return 5; }
