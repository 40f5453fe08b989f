/*
 * ofx_oracle.c -- CPU ORACLE (test infrastructure only, see ofx_oracle.h).
 *
 * Plain-C99 restatement of the reference's dense pyramidal Lucas-Kanade path.
 * Written from the semantics recorded in SURVEY.md section 8a; each function
 * names the reference lines it restates.  Build with -ffp-contract=off so the
 * float expressions round exactly as the reference's x86-64 build does.
 *
 * Loop structure deliberately differs from the reference: clipped tap ranges
 * are computed once per pixel instead of testing every tap, which is the same
 * set of taps in the same (row-major) order.
 */
#include "ofx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_E
#define M_E 2.7182818284590452354
#endif

/* kernels.cpp:6-10, :15-19, :20-24, :61-64 */
const float orc_Dx_3x3[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1};
const float orc_Dy_3x3[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
const float orc_Dt_3x3[9] = {1, 2, 1, 2, 3, 2, 1, 2, 1};
const float orc_GAUS_3x3[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* tap index range [lo,hi) of a length-`len` window starting at `start` that
 * falls inside [0,extent) */
static inline void clip_taps(int start, int len, int extent, int *lo, int *hi)
{
    *lo = imax(0, -start);
    *hi = imin(len, extent - start);
}

/* ------------------------------------------------------------------------ */
void orc_sub_u8(const uint8_t *a, const uint8_t *b, int n, uint8_t *dst)
{
    /* OptFlowCPU.cpp:15 : unsigned char difference, wraps mod 256 */
    for (int i = 0; i < n; ++i) dst[i] = (uint8_t)(a[i] - b[i]);
}

void orc_sub_f32(const float *a, const float *b, int n, float *dst)
{
    /* OptFlowUtils.hpp:25 */
    for (int i = 0; i < n; ++i) dst[i] = a[i] - b[i];
}

void orc_grayscale_avg(const uint8_t *src3, uint8_t *dst3, int w, int h)
{
    /* OptFlowCPU.cpp:27-28 : integer mean of the three bytes, replicated */
    const size_t n = (size_t)w * (size_t)h;
    for (size_t p = 0; p < n; ++p) {
        const uint8_t *s = src3 + 3 * p;
        uint8_t g = (uint8_t)(((int)s[0] + (int)s[1] + (int)s[2]) / 3);
        dst3[3 * p] = dst3[3 * p + 1] = dst3[3 * p + 2] = g;
    }
}

/* The reference accumulates `int tmp += uchar * float`: the sum is formed in
 * float and truncated back to int after EVERY tap (OptFlowCPU.cpp:62,102). */
static inline int acc_trunc(int acc, uint8_t px, float m) { return (int)((float)acc + (float)px * m); }

void orc_conv_3ch(const uint8_t *src3, const float *mask, uint8_t *dst3, int w, int h, int mw, int mh)
{
    /* OptFlowCPU.cpp:33-73 */
    const int ox = mw >> 1, oy = mh >> 1;
    for (int y = 0; y < h; ++y) {
        int i0, i1;
        clip_taps(y - oy, mh, h, &i0, &i1);
        for (int x = 0; x < w; ++x) {
            int j0, j1, acc[3] = {0, 0, 0};
            clip_taps(x - ox, mw, w, &j0, &j1);
            for (int i = i0; i < i1; ++i)
                for (int j = j0; j < j1; ++j) {
                    const uint8_t *s = src3 + 3 * ((size_t)(y - oy + i) * w + (x - ox + j));
                    const float m = mask[i * mw + j];
                    acc[0] = acc_trunc(acc[0], s[0], m);
                    acc[1] = acc_trunc(acc[1], s[1], m);
                    acc[2] = acc_trunc(acc[2], s[2], m);
                }
            uint8_t *d = dst3 + 3 * ((size_t)y * w + x);
            d[0] = (uint8_t)acc[0];
            d[1] = (uint8_t)acc[1];
            d[2] = (uint8_t)acc[2];
        }
    }
}

void orc_conv_3ch_to_1ch(const uint8_t *src3, int w, int h, uint8_t *dst, const float *mask, int mw, int mh)
{
    /* OptFlowCPU.cpp:75-109 : reads byte 0 of every pixel (:102), wraps (:106) */
    const int ox = mw >> 1, oy = mh >> 1;
    for (int y = 0; y < h; ++y) {
        int i0, i1;
        clip_taps(y - oy, mh, h, &i0, &i1);
        for (int x = 0; x < w; ++x) {
            int j0, j1, acc = 0;
            clip_taps(x - ox, mw, w, &j0, &j1);
            for (int i = i0; i < i1; ++i)
                for (int j = j0; j < j1; ++j)
                    acc = acc_trunc(acc, src3[3 * ((size_t)(y - oy + i) * w + (x - ox + j))], mask[i * mw + j]);
            dst[(size_t)y * w + x] = (uint8_t)acc;
        }
    }
}

void orc_conv_3ch_to_1ch_f32(const uint8_t *src3, int w, int h, float *dst, const float *mask, int mw, int mh)
{
    /* OptFlowGpu.cu:1040-1090 : float accumulator, taps with a zero weight are
     * skipped (:1075), result stored unrounded */
    const int ox = mw >> 1, oy = mh >> 1;
    for (int y = 0; y < h; ++y) {
        int i0, i1;
        clip_taps(y - oy, mh, h, &i0, &i1);
        for (int x = 0; x < w; ++x) {
            int j0, j1;
            float acc = 0.0f;
            clip_taps(x - ox, mw, w, &j0, &j1);
            for (int i = i0; i < i1; ++i)
                for (int j = j0; j < j1; ++j) {
                    const float m = mask[i * mw + j];
                    if (m == 0.0f) continue;
                    acc += (float)src3[3 * ((size_t)(y - oy + i) * w + (x - ox + j))] * m;
                }
            dst[(size_t)y * w + x] = acc;
        }
    }
}

/* ------------------------------------------------------------------------ */
void orc_downscale_gaussian(const uint8_t *src3, int w, int h, uint8_t *dst3, const float *mask, int mw, int mh)
{
    /* OptFlowCPU.cpp:112-148 : destination (w,h); source is (2w,2h) with row
     * stride exactly 2w (:117,:136); float accumulators (:124), float->uchar
     * truncation (:143-145); taps outside the source are skipped (:133) */
    const int ox = mw >> 1, oy = mh >> 1;
    const int sw = w << 1, sh = h << 1;
    for (int y = 0; y < h; ++y) {
        int p0, p1;
        clip_taps(2 * y - oy, mh, sh, &p0, &p1);
        for (int x = 0; x < w; ++x) {
            int q0, q1;
            float acc[3] = {0.0f, 0.0f, 0.0f};
            clip_taps(2 * x - ox, mw, sw, &q0, &q1);
            for (int p = p0; p < p1; ++p)
                for (int q = q0; q < q1; ++q) {
                    const uint8_t *s = src3 + 3 * ((size_t)(2 * y - oy + p) * sw + (2 * x - ox + q));
                    const float m = mask[p * mw + q];
                    acc[0] += m * (float)s[0];
                    acc[1] += m * (float)s[1];
                    acc[2] += m * (float)s[2];
                }
            uint8_t *d = dst3 + 3 * ((size_t)y * w + x);
            d[0] = (uint8_t)acc[0];
            d[1] = (uint8_t)acc[1];
            d[2] = (uint8_t)acc[2];
        }
    }
}

void orc_gauss_pyramid(uint8_t **pyr3, int w, int h, int n, const float *mask, int mw, int mh)
{
    /* OptFlowCPU.cpp:151-160 : level k from level k-1, dims (w>>k, h>>k) */
    for (int k = 1; k < n; ++k) orc_downscale_gaussian(pyr3[k - 1], w >> k, h >> k, pyr3[k], mask, mw, mh);
}

/* ------------------------------------------------------------------------ */
void orc_srm_1ch(const uint8_t *a, const uint8_t *b, int w, int h, int ww, int wh, int32_t *dst)
{
    /* OptFlowCPU.cpp:162-200 : window clipped at the image border, not padded */
    const int ox = ww >> 1, oy = wh >> 1;
    for (int y = 0; y < h; ++y) {
        int p0, p1;
        clip_taps(y - oy, wh, h, &p0, &p1);
        for (int x = 0; x < w; ++x) {
            int q0, q1;
            int32_t acc = 0;
            clip_taps(x - ox, ww, w, &q0, &q1);
            for (int p = p0; p < p1; ++p) {
                const size_t row = (size_t)(y - oy + p) * w + (x - ox);
                for (int q = q0; q < q1; ++q) acc += (int32_t)a[row + q] * (int32_t)b[row + q];
            }
            dst[(size_t)y * w + x] = acc;
        }
    }
}

void orc_srm_1ch_f32(const float *a, const float *b, int w, int h, int ww, int wh, float *dst)
{
    /* OptFlowGpu.cu:1549-1588 : float accumulator, row-major tap order */
    const int ox = ww >> 1, oy = wh >> 1;
    for (int y = 0; y < h; ++y) {
        int p0, p1;
        clip_taps(y - oy, wh, h, &p0, &p1);
        for (int x = 0; x < w; ++x) {
            int q0, q1;
            float acc = 0.0f;
            clip_taps(x - ox, ww, w, &q0, &q1);
            for (int p = p0; p < p1; ++p) {
                const size_t row = (size_t)(y - oy + p) * w + (x - ox);
                for (int q = q0; q < q1; ++q) acc += a[row + q] * b[row + q];
            }
            dst[(size_t)y * w + x] = acc;
        }
    }
}

void orc_srm_1ch_f32_exact(const float *a, const float *b, int w, int h, int ww, int wh, float *dst)
{
    /* same taps as orc_srm_1ch_f32; products and sum formed in double (exact
     * for integer-valued planes with |a*b| sums < 2^53), one rounding to float */
    const int ox = ww >> 1, oy = wh >> 1;
    for (int y = 0; y < h; ++y) {
        int p0, p1;
        clip_taps(y - oy, wh, h, &p0, &p1);
        for (int x = 0; x < w; ++x) {
            int q0, q1;
            double acc = 0.0;
            clip_taps(x - ox, ww, w, &q0, &q1);
            for (int p = p0; p < p1; ++p) {
                const size_t row = (size_t)(y - oy + p) * w + (x - ox);
                for (int q = q0; q < q1; ++q) acc += (double)a[row + q] * (double)b[row + q];
            }
            dst[(size_t)y * w + x] = (float)acc;
        }
    }
}

/* ------------------------------------------------------------------------ */
void orc_shift_back_pyramid(const uint8_t *src3, int w, int h, int level, int max_level,
                            float *const *flow_pyr, uint8_t *dst3)
{
    /* OptFlowCPU.cpp:241-282.
     *  :247      only w*h BYTES (a third of the image) are copied first.
     *  :260-262  `i * (1 >> offset)` is 0 for every offset >= 1, so every pixel
     *            reads flow element 0 of each coarser level: one translation.
     *  :264-265  float accumulation, coarsest level first.
     *  :268-273  (int)(j + u): float add, truncation toward zero; out-of-image
     *            targets leave dst untouched.  A non-finite or huge sum is UB
     *            in the reference and lands out-of-image on x86 (INT_MIN);
     *            here it is out-of-image by definition. */
    memcpy(dst3, src3, (size_t)w * (size_t)h);
    float u = 0.0f, v = 0.0f;
    for (int k = max_level - 1; k > level; --k) {
        const float mult = (float)(1 << (k - level));
        u += mult * flow_pyr[k][0];
        v += mult * flow_pyr[k][1];
    }
    for (int i = 0; i < h; ++i) {
        const float ty = (float)i + v;
        if (!(ty > -1.0f && ty < (float)h)) continue;
        const int ny = (int)ty;
        for (int j = 0; j < w; ++j) {
            const float tx = (float)j + u;
            if (!(tx > -1.0f && tx < (float)w)) continue;
            const int nx = (int)tx;
            const uint8_t *s = src3 + 3 * ((size_t)ny * w + nx);
            uint8_t *d = dst3 + 3 * ((size_t)i * w + j);
            d[0] = s[0];
            d[1] = s[1];
            d[2] = s[2];
        }
    }
}

/* ------------------------------------------------------------------------ */
void orc_inverse_matrix_f32arith(const int32_t *sxx, const int32_t *syy, const int32_t *sxy,
                                 const int32_t *sxt, const int32_t *syt, float *flow, int w, int h)
{
    /* OptFlowCPU.cpp:285-309 : everything in float; int sums are converted
     * to float by the usual arithmetic conversions at :303-304 */
    const size_t n = (size_t)w * (size_t)h;
    for (size_t p = 0; p < n; ++p) {
        float a = (float)sxx[p], b = (float)sxy[p], c = b, d = (float)syy[p];
        const float pre = 1 / (a * d - b * c);
        a *= pre;
        b *= pre;
        c *= pre;
        d *= pre;
        flow[2 * p] = -d * (float)sxt[p] + b * (float)syt[p];
        flow[2 * p + 1] = c * (float)sxt[p] - a * (float)syt[p];
    }
}

static inline void solve_f64(double a, double b, double d, double xt, double yt, int scale_c, float *uv)
{
    /* OptFlowGpu.cu:1737-1754 (scale_c=1) / OptFlowCPU.cpp:369-382 (scale_c=0) */
    double c = b;
    const double pre = 1 / (a * d - b * c);
    a *= pre;
    b *= pre;
    if (scale_c) c *= pre;
    d *= pre;
    uv[0] = (float)(-d * xt + b * yt);
    uv[1] = (float)(c * xt - a * yt);
}

void orc_inverse_matrix_i32(const int32_t *sxx, const int32_t *syy, const int32_t *sxy,
                            const int32_t *sxt, const int32_t *syt, float *flow, int w, int h)
{
    const size_t n = (size_t)w * (size_t)h;
    for (size_t p = 0; p < n; ++p)
        solve_f64((double)sxx[p], (double)sxy[p], (double)syy[p], (double)sxt[p], (double)syt[p], 1, flow + 2 * p);
}

void orc_inverse_matrix_f32(const float *sxx, const float *syy, const float *sxy,
                            const float *sxt, const float *syt, float *flow, int w, int h)
{
    const size_t n = (size_t)w * (size_t)h;
    for (size_t p = 0; p < n; ++p)
        solve_f64((double)sxx[p], (double)sxy[p], (double)syy[p], (double)sxt[p], (double)syt[p], 1, flow + 2 * p);
}

void orc_inverse_matrix_inline_cpu(const int32_t *sxx, const int32_t *syy, const int32_t *sxy,
                                   const int32_t *sxt, const int32_t *syt, float *flow, int w, int h)
{
    const size_t n = (size_t)w * (size_t)h;
    for (size_t p = 0; p < n; ++p)
        solve_f64((double)sxx[p], (double)sxy[p], (double)syy[p], (double)sxt[p], (double)syt[p], 0, flow + 2 * p);
}

/* ------------------------------------------------------------------------ */
void orc_calc_optical_flow_cpu(const uint8_t *prev3, const uint8_t *next3, int w, int h,
                               float **flow_pyr, int level, int max_level, int window)
{
    /* OptFlowCPU.cpp:312-399 */
    const size_t n = (size_t)w * (size_t)h;
    uint8_t *shifted = (uint8_t *)calloc(3 * n, 1); /* :320, pinned as zero-filled */
    if (level != max_level - 1) {
        orc_shift_back_pyramid(next3, w, h, level, max_level, flow_pyr, shifted); /* :323 */
        next3 = shifted;
    }
    uint8_t *ix = (uint8_t *)malloc(n), *iy = (uint8_t *)malloc(n), *t1 = (uint8_t *)malloc(n), *t2 = (uint8_t *)malloc(n);
    orc_conv_3ch_to_1ch(prev3, w, h, ix, orc_Dx_3x3, 3, 3);   /* :330 */
    orc_conv_3ch_to_1ch(prev3, w, h, iy, orc_Dy_3x3, 3, 3);   /* :333 */
    orc_conv_3ch_to_1ch(prev3, w, h, t1, orc_GAUS_3x3, 3, 3); /* :336 */
    orc_conv_3ch_to_1ch(next3, w, h, t2, orc_GAUS_3x3, 3, 3); /* :338 */
    orc_sub_u8(t2, t1, (int)n, t1);                           /* :340, It in place */
    int32_t *s = (int32_t *)malloc(5 * n * sizeof(int32_t));
    orc_srm_1ch(ix, ix, w, h, window, window, s);         /* :347 */
    orc_srm_1ch(iy, iy, w, h, window, window, s + n);     /* :350 */
    orc_srm_1ch(ix, iy, w, h, window, window, s + 2 * n); /* :353 */
    orc_srm_1ch(ix, t1, w, h, window, window, s + 3 * n); /* :356 */
    orc_srm_1ch(iy, t1, w, h, window, window, s + 4 * n); /* :358 */
    orc_inverse_matrix_inline_cpu(s, s + n, s + 2 * n, s + 3 * n, s + 4 * n, flow_pyr[level], w, h); /* :363-384 */
    free(s);
    free(ix);
    free(iy);
    free(t1);
    free(t2);
    free(shifted);
}

void orc_calc_opt_flow_gpu(const uint8_t *prev3, const uint8_t *next3, int w, int h,
                           float **flow_pyr, int level, int max_level, int window, int exact_sums)
{
    /* OptFlowGpu.cu:1909-1979 */
    const size_t n = (size_t)w * (size_t)h;
    uint8_t *shifted = (uint8_t *)calloc(3 * n, 1); /* :1917 */
    if (level != max_level - 1) {
        orc_shift_back_pyramid(next3, w, h, level, max_level, flow_pyr, shifted); /* :1920 */
        next3 = shifted;
    }
    float *ix = (float *)malloc(n * sizeof(float)), *iy = (float *)malloc(n * sizeof(float));
    float *t1 = (float *)malloc(n * sizeof(float)), *t2 = (float *)malloc(n * sizeof(float));
    orc_conv_3ch_to_1ch_f32(prev3, w, h, ix, orc_Dx_3x3, 3, 3); /* :1930 */
    orc_conv_3ch_to_1ch_f32(prev3, w, h, iy, orc_Dy_3x3, 3, 3); /* :1933 */
    orc_conv_3ch_to_1ch_f32(prev3, w, h, t1, orc_Dt_3x3, 3, 3); /* :1936 */
    orc_conv_3ch_to_1ch_f32(next3, w, h, t2, orc_Dt_3x3, 3, 3); /* :1938 */
    orc_sub_f32(t2, t1, (int)n, t1);                            /* :1940 */
    float *s = (float *)malloc(5 * n * sizeof(float));
    void (*srm)(const float *, const float *, int, int, int, int, float *) =
        exact_sums ? orc_srm_1ch_f32_exact : orc_srm_1ch_f32;
    srm(ix, ix, w, h, window, window, s);         /* :1948 */
    srm(iy, iy, w, h, window, window, s + n);     /* :1951 */
    srm(ix, iy, w, h, window, window, s + 2 * n); /* :1954 */
    srm(ix, t1, w, h, window, window, s + 3 * n); /* :1957 */
    srm(iy, t1, w, h, window, window, s + 4 * n); /* :1960 */
    orc_inverse_matrix_f32(s, s + n, s + 2 * n, s + 3 * n, s + 4 * n, flow_pyr[level], w, h); /* :1964 */
    free(s);
    free(ix);
    free(iy);
    free(t1);
    free(t2);
    free(shifted);
}

void orc_level_planes(const uint8_t *prev3, const uint8_t *next3, int w, int h, int window, int mode,
                      int exact_sums, float *ix_o, float *iy_o, float *it_o, double *sums5)
{
    const size_t n = (size_t)w * (size_t)h;
    if (mode == 0) {
        uint8_t *ix = (uint8_t *)malloc(n), *iy = (uint8_t *)malloc(n), *t1 = (uint8_t *)malloc(n), *t2 = (uint8_t *)malloc(n);
        orc_conv_3ch_to_1ch(prev3, w, h, ix, orc_Dx_3x3, 3, 3);
        orc_conv_3ch_to_1ch(prev3, w, h, iy, orc_Dy_3x3, 3, 3);
        orc_conv_3ch_to_1ch(prev3, w, h, t1, orc_GAUS_3x3, 3, 3);
        orc_conv_3ch_to_1ch(next3, w, h, t2, orc_GAUS_3x3, 3, 3);
        orc_sub_u8(t2, t1, (int)n, t1);
        for (size_t p = 0; p < n; ++p) {
            if (ix_o) ix_o[p] = ix[p];
            if (iy_o) iy_o[p] = iy[p];
            if (it_o) it_o[p] = t1[p];
        }
        if (sums5) {
            int32_t *s = (int32_t *)malloc(n * sizeof(int32_t));
            const uint8_t *A[5] = {ix, iy, ix, ix, iy}, *B[5] = {ix, iy, iy, t1, t1};
            for (int k = 0; k < 5; ++k) {
                orc_srm_1ch(A[k], B[k], w, h, window, window, s);
                for (size_t p = 0; p < n; ++p) sums5[k * n + p] = s[p];
            }
            free(s);
        }
        free(ix);
        free(iy);
        free(t1);
        free(t2);
    } else {
        float *ix = (float *)malloc(n * sizeof(float)), *iy = (float *)malloc(n * sizeof(float));
        float *t1 = (float *)malloc(n * sizeof(float)), *t2 = (float *)malloc(n * sizeof(float));
        orc_conv_3ch_to_1ch_f32(prev3, w, h, ix, orc_Dx_3x3, 3, 3);
        orc_conv_3ch_to_1ch_f32(prev3, w, h, iy, orc_Dy_3x3, 3, 3);
        orc_conv_3ch_to_1ch_f32(prev3, w, h, t1, orc_Dt_3x3, 3, 3);
        orc_conv_3ch_to_1ch_f32(next3, w, h, t2, orc_Dt_3x3, 3, 3);
        orc_sub_f32(t2, t1, (int)n, t1);
        if (ix_o) memcpy(ix_o, ix, n * sizeof(float));
        if (iy_o) memcpy(iy_o, iy, n * sizeof(float));
        if (it_o) memcpy(it_o, t1, n * sizeof(float));
        if (sums5) {
            float *s = (float *)malloc(n * sizeof(float));
            const float *A[5] = {ix, iy, ix, ix, iy}, *B[5] = {ix, iy, iy, t1, t1};
            for (int k = 0; k < 5; ++k) {
                if (exact_sums) orc_srm_1ch_f32_exact(A[k], B[k], w, h, window, window, s);
                else orc_srm_1ch_f32(A[k], B[k], w, h, window, window, s);
                for (size_t p = 0; p < n; ++p) sums5[k * n + p] = s[p];
            }
            free(s);
        }
        free(ix);
        free(iy);
        free(t1);
        free(t2);
    }
}

void orc_compose_flow(float *const *flow_pyr, int w, int h, int levels, int level, float *dst_uv)
{
    /* main.cu:138-147 : (w,h) are the dims AT `level`; each coarser level is
     * sampled at (i>>scale, j>>scale) and weighted 2^scale; float u,v updated
     * through a double product (:145-146) */
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            float u = 0.0f, v = 0.0f;
            for (int k = levels - 1; k >= level; --k) {
                const int sc = k - level;
                const size_t pos = (size_t)(i >> sc) * (size_t)(w >> sc) + (size_t)(j >> sc);
                const double m = (double)(1 << sc);
                u = (float)((double)u + m * (double)flow_pyr[k][2 * pos]);
                v = (float)((double)v + m * (double)flow_pyr[k][2 * pos + 1]);
            }
            dst_uv[2 * ((size_t)i * w + j)] = u;
            dst_uv[2 * ((size_t)i * w + j) + 1] = v;
        }
}

/* ------------------------------------------------------------------------ */
void orc_generate_gaussian_kernel(double sigma_s, int ks, double *dst)
{
    /* OptFlowUtils.cpp:68-114 : size -1 -> 2*pi*sigma (:70-73), even -> +1
     * (:74-77); value depends only on (|i-c|,|j-c|) (:92-97); normalised by the
     * row-major sum (:100-113) */
    if (ks == -1) ks = (int)(2.0 * M_PI * sigma_s);
    if (ks % 2 == 0) ks += 1;
    const int c = ks >> 1;
    const double s2 = sigma_s * sigma_s;
    for (int i = 0; i < ks; ++i)
        for (int j = 0; j < ks; ++j) {
            const double m = (double)abs(i - c), n = (double)abs(j - c);
            dst[i * ks + j] = 1.0 / (2.0 * M_PI * s2) * pow(M_E, -0.5 * (n * n + m * m) / s2);
        }
    double sum = 0;
    for (int i = 0; i < ks * ks; ++i) sum += dst[i];
    for (int i = 0; i < ks * ks; ++i) dst[i] /= sum;
}

void orc_bilateral_3ch(const uint8_t *src3, const uint8_t *gray3, uint8_t *dst3, int w, int h,
                       int ww, int wh, double sigma_s, double sigma_b)
{
    /* OptFlowCPU.cpp:401-465 : spatial mask generated with ww only (:404);
     * range weight from channel 0 of `gray` (:419,:442-446); clipped window;
     * weights multiplied as (src*n_b)*n_s (:450-452); truncation to u8 */
    double *gm = (double *)malloc((size_t)(ww | 1) * (size_t)(ww | 1) * sizeof(double) + 64);
    orc_generate_gaussian_kernel(sigma_s, ww, gm);
    const int ox = ww >> 1, oy = wh >> 1;
    const double sb2 = sigma_b * sigma_b;
    for (int i = 0; i < h; ++i) {
        int m0, m1;
        clip_taps(i - oy, wh, h, &m0, &m1);
        for (int j = 0; j < w; ++j) {
            int n0, n1;
            clip_taps(j - ox, ww, w, &n0, &n1);
            const double f0 = gray3[3 * ((size_t)i * w + j)];
            double wsum = 0, acc[3] = {0, 0, 0};
            for (int m = m0; m < m1; ++m)
                for (int n = n0; n < n1; ++n) {
                    const size_t q = (size_t)(i - oy + m) * w + (j - ox + n);
                    const double k = (double)gray3[3 * q] - f0;
                    const double nb = 1.0 / (2.0 * M_PI * sb2) * pow(M_E, -0.5 * (k * k) / sb2);
                    const double ns = gm[m * ww + n];
                    wsum += nb * ns;
                    acc[0] += src3[3 * q] * nb * ns;
                    acc[1] += src3[3 * q + 1] * nb * ns;
                    acc[2] += src3[3 * q + 2] * nb * ns;
                }
            uint8_t *d = dst3 + 3 * ((size_t)i * w + j);
            d[0] = (uint8_t)(acc[0] / wsum);
            d[1] = (uint8_t)(acc[1] / wsum);
            d[2] = (uint8_t)(acc[2] / wsum);
        }
    }
    free(gm);
}

/* ------------------------------------------------------------------------ */
void orc_replicate_1ch_to_3ch(const uint8_t *src1, uint8_t *dst3, int n)
{
    for (int i = 0; i < n; ++i) dst3[3 * i] = dst3[3 * i + 1] = dst3[3 * i + 2] = src1[i];
}

void orc_extract_ch0(const uint8_t *src3, uint8_t *dst1, int n)
{
    for (int i = 0; i < n; ++i) dst1[i] = src3[3 * i];
}

/* ------------------------------------------------------------------------ */
/* Extension (no reference twin): see ofx_oracle.h.  Every operation is a single-rounding float op in the order
 * written (build with -ffp-contract=off); the HIP warp kernel performs the same sequence. */
void orc_warp_bilinear_u8(const uint8_t *src1, int w, int h, const float *flow_uv, float scale, uint8_t *dst1)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t p = (size_t)y * w + x;
            float sx = (float)x + scale * flow_uv[2 * p];
            float sy = (float)y + scale * flow_uv[2 * p + 1];
            if (!(sx >= -1e9f && sx <= 1e9f) || !(sy >= -1e9f && sy <= 1e9f)) { /* non-finite flow: no warp */
                dst1[p] = src1[p];
                continue;
            }
            /* replicate border */
            sx = sx < 0.0f ? 0.0f : (sx > (float)(w - 1) ? (float)(w - 1) : sx);
            sy = sy < 0.0f ? 0.0f : (sy > (float)(h - 1) ? (float)(h - 1) : sy);
            const int x0 = (int)sx, y0 = (int)sy; /* sx, sy >= 0: truncation == floor */
            const int x1 = x0 + 1 < w ? x0 + 1 : w - 1, y1 = y0 + 1 < h ? y0 + 1 : h - 1;
            const float fx = sx - (float)x0, fy = sy - (float)y0;
            const float p00 = src1[(size_t)y0 * w + x0], p01 = src1[(size_t)y0 * w + x1];
            const float p10 = src1[(size_t)y1 * w + x0], p11 = src1[(size_t)y1 * w + x1];
            const float a = p00 + fx * (p01 - p00);
            const float b = p10 + fx * (p11 - p10);
            const float v = a + fy * (b - a);
            dst1[p] = (uint8_t)(int)(v + 0.5f);
        }
}

void orc_lk_iter_level(const uint8_t *prev1, const uint8_t *next1_shifted, int w, int h, int window, int iters,
                       float *flow_uv, float *flow_first)
{
    const size_t n = (size_t)w * (size_t)h;
    uint8_t *p3 = (uint8_t *)malloc(3 * n), *n3 = (uint8_t *)malloc(3 * n), *warped = (uint8_t *)malloc(n);
    float *delta = (float *)malloc(2 * n * sizeof(float));
    float *levels[1] = {delta};
    orc_replicate_1ch_to_3ch(prev1, p3, (int)n);
    for (int it = 0; it < iters; ++it) {
        if (it == 0) {
            orc_replicate_1ch_to_3ch(next1_shifted, n3, (int)n);
        } else {
            orc_warp_bilinear_u8(next1_shifted, w, h, flow_uv, ORC_ITER_SCALE, warped);
            orc_replicate_1ch_to_3ch(warped, n3, (int)n);
        }
        /* one level, already shifted: level == max_level - 1 skips the shift inside */
        orc_calc_opt_flow_gpu(p3, n3, w, h, levels, 0, 1, window, 1);
        if (it == 0) {
            memcpy(flow_uv, delta, 2 * n * sizeof(float));
            if (flow_first) memcpy(flow_first, delta, 2 * n * sizeof(float));
        } else {
            for (size_t i = 0; i < 2 * n; ++i) flow_uv[i] = flow_uv[i] + delta[i];
        }
    }
    free(delta);
    free(warped);
    free(n3);
    free(p3);
}

/* ------------------------------------------------------------------------ */
/* The rest of the reference's exported surface: link-compat functions the flow path does not use. */

void orc_srm_3ch(const uint8_t *a3, const uint8_t *b3, int w, int h, int ww, int wh, int32_t *dst3)
{
    /* OptFlowCPU.cpp:202-238.  The window test is `cx < 0 || cy < 0 || cx > w || cy > h` (:222): column w and row h pass,
     * so a tap one past the right edge lands on the first pixel of the next row (pos = cy*w + w) and a tap on row h reads
     * past the image.  Positions inside the buffer (pos < w*h) are read exactly as the reference does; positions past its
     * end -- undefined in the reference -- contribute nothing. */
    const int ox = ww >> 1, oy = wh >> 1;
    const size_t n = (size_t)w * (size_t)h;
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            int acc[3] = {0, 0, 0};
            for (int y = 0; y < wh; ++y) {
                const int cy = i - oy + y;
                if (cy < 0 || cy > h) continue;
                for (int x = 0; x < ww; ++x) {
                    const int cx = j - ox + x;
                    if (cx < 0 || cx > w) continue;
                    const size_t t = (size_t)cy * (size_t)w + (size_t)cx;
                    if (t >= n) continue;
                    acc[0] += (int)a3[3 * t] * (int)b3[3 * t];
                    acc[1] += (int)a3[3 * t + 1] * (int)b3[3 * t + 1];
                    acc[2] += (int)a3[3 * t + 2] * (int)b3[3 * t + 2];
                }
            }
            int32_t *d = dst3 + 3 * ((size_t)i * w + j);
            d[0] = acc[0];
            d[1] = acc[1];
            d[2] = acc[2];
        }
}

void orc_cleanup_outliers(uint8_t *img1, int w, int h)
{
    /* OptFlowUtils.cpp:5-19: values in [20, 240) -> 255, the rest -> 0 */
    const size_t n = (size_t)w * (size_t)h;
    for (size_t p = 0; p < n; ++p) img1[p] = (img1[p] >= 240 || img1[p] < 20) ? 0 : 255;
}

void orc_upscale(const uint8_t *src, int w, int h, int n, int channels, uint8_t *dst)
{
    /* OptFlowUtils.cpp:21-61 (channels = 3 / 1): every source pixel becomes a 2^n x 2^n block */
    const size_t ow = (size_t)w << n, oh = (size_t)h << n;
    for (size_t oy = 0; oy < oh; ++oy)
        for (size_t ox = 0; ox < ow; ++ox)
            for (int c = 0; c < channels; ++c)
                dst[channels * (oy * ow + ox) + c] = src[channels * ((oy >> n) * (size_t)w + (ox >> n)) + c];
}

void orc_conv_1d_3ch(const uint8_t *src3, int w, int h, uint8_t *dst3)
{
    /* OptFlowGpu.cu:1134-1189: 9 taps (0.1 .. 0.5 .. 0.1, :1164) along the PIXEL sequence, int accumulators that are
     * truncated after every tap (`temp += (float)src * mask`, :1152-1154), result as unsigned char.  The reference passes
     * the byte size as the element count (:1184), so its taps run up to 4 pixels past the end of the buffer; here taps
     * outside [0, w*h) are skipped (the library's documented deviation, include/OptFlowGpu.cuh). */
    static const float wgt[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.4f, 0.3f, 0.2f, 0.1f};
    const long npix = (long)w * (long)h;
    for (long x = 0; x < npix; ++x) {
        int acc[3] = {0, 0, 0};
        for (int i = 0; i < 9; ++i) {
            const long t = x - 4 + i;
            if (t < 0 || t >= npix) continue;
            for (int c = 0; c < 3; ++c) acc[c] = (int)((float)acc[c] + (float)src3[3 * t + c] * wgt[i]);
        }
        for (int c = 0; c < 3; ++c) dst3[3 * x + c] = (uint8_t)acc[c];
    }
}
