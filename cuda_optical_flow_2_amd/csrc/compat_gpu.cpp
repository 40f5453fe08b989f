// namespace gpu / namespace utils / stencil tables: the reference's C++ call surface (include/OptFlowGpu.cuh,
// include/OptFlowUtils.hpp, include/kernels.hpp) implemented on top of the C ABI of ofx.h.
//
// Contract kept from the reference: host pointers owned by the caller, synchronous calls, void
// returns, nothing retained.  What is new: every HIP call is checked, failures are recorded
// (gpu_compat_last_status / ofx_last_error) and the output buffers are left untouched on failure.
#include <vector>

#include "OptFlowGpu.cuh"
#include "OptFlowUtils.hpp"
#include "compat_scratch.h"
#include "kernels.hpp"
#include "ofx_internal.h"

// ---- stencil tables (values of the reference's kernels.cpp:6-64) ---------------------------------------------------
extern const float Dx_3x3[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1};
extern const float Dx_3x3_t[9] = {1.0 * 1.0 / 3.0, 0, -1.0 * 1.0 / 3.0, 2.0 * 1.0 / 3.0, 0, -2.0 * 1.0 / 3.0,
                                  1.0 * 1.0 / 3.0, 0, -1.0 * 1.0 / 3.0};
extern const float Dy_3x3[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
extern const float Dt_3x3[9] = {1, 2, 1, 2, 3, 2, 1, 2, 1};
extern const float Dt_3x3_n[9] = {0.0666, 0.1333, 0.0666, 0.1333, 0.2, 0.1333, 0.0666, 0.1333, 0.0666};
extern const float Dy_DIAGONAL_2x2[9] = {1, 0, 0, 0, -1, 0, 0, 0, 0};
extern const float Dy_2x2[9] = {-1, -1, 0, 1, 1, 0, 0, 0, 0};
extern const float Dz_2x2[9] = {1, 1, 0, 1, 1, 0, 0, 0, 0};
extern const float Dx_5x5[25] = {-1, -2, 0, 1, 2, -2, -3, 0, 2, 3, -3, -5, 0, 3, 5, -2, -3, 0, 3, 2, -1, -2, 0, 2, 1};
extern const float GAUS_KERNEL_5x5[25] = {0.00366, 0.01465, 0.02564, 0.01465, 0.00366, 0.01465, 0.05860, 0.09523, 0.05860,
                                          0.01465, 0.02564, 0.09523, 0.15018, 0.09523, 0.02564, 0.01465, 0.05860, 0.09523,
                                          0.05860, 0.01465, 0.00366, 0.01465, 0.02564, 0.01465, 0.00366};
extern const float GAUS_KERNEL_3x3[9] = {0.0625, 0.125, 0.0625, 0.125, 0.25, 0.125, 0.0625, 0.125, 0.0625};

namespace ofx_compat {
int &status()
{
    static thread_local int st = OFX_OK;
    return st;
}
} // namespace ofx_compat

using ofx_compat::args_ok;
using ofx_compat::Scratch;

namespace {

// 9-tap 1-D filter over the pixel sequence (reference OptFlowGpu.cu:1134-1159, weights from :1164): int accumulators
// truncated after every tap.  The reference lets taps run up to 4 pixels past the end of the buffer; here taps
// outside [0, npix) are skipped.
__global__ void conv_1d_3ch_kernel(const unsigned char *src, unsigned char *dst, int npix)
{
    const float wgt[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.4f, 0.3f, 0.2f, 0.1f};
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= npix) return;
    int acc[3] = {0, 0, 0};
    for (int i = 0; i < 9; ++i) {
        const int t = x - 4 + i;
        if (t < 0 || t >= npix) continue;
        for (int c = 0; c < 3; ++c) acc[c] = (int)((float)acc[c] + (float)src[3 * t + c] * wgt[i]);
    }
    for (int c = 0; c < 3; ++c) dst[3 * x + c] = (unsigned char)acc[c];
}

} // namespace

extern "C" int gpu_compat_last_status(void) { return ofx_compat::status(); }

namespace gpu {

void grayscale_avg(const unsigned char *rgb, unsigned char *gray3, int rows, int cols)
{
    if (!args_ok(rgb && gray3 && rows > 0 && cols > 0, "gpu::grayscale_avg")) return;
    const size_t n = (size_t)rows * cols * 3;
    Scratch s;
    unsigned char *d_in = s.upload(rgb, n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_grayscale_avg_3ch(d_in, d_out, cols, rows, nullptr));
    s.download(gray3, d_out, n);
    ofx_compat::status() = s.rc();
}

static void conv3(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh, int float_acc,
                  const char *who)
{
    if (!args_ok(img3 && out3 && mask && w > 0 && h > 0, who)) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(img3, n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_conv_3ch(d_in, d_out, w, h, mask, mw, mh, float_acc, nullptr));
    s.download(out3, d_out, n);
    ofx_compat::status() = s.rc();
}

void conv_3ch_2d(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh)
{
    conv3(img3, out3, w, h, mask, mw, mh, 0, "gpu::conv_3ch_2d");
}

void conv_3ch_2d_constant(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh)
{
    conv3(img3, out3, w, h, mask, mw, mh, 0, "gpu::conv_3ch_2d_constant");
}

void conv_3ch_tiled(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh)
{
    conv3(img3, out3, w, h, mask, mw, mh, 1, "gpu::conv_3ch_tiled");
}

static void conv1_u8(const unsigned char *img3, int w, int h, unsigned char *out1, const float *mask, int mw, int mh, const char *who)
{
    if (!args_ok(img3 && out1 && mask && w > 0 && h > 0, who)) return;
    const size_t n = (size_t)w * h;
    Scratch s;
    unsigned char *d_in = s.upload(img3, 3 * n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_conv_3ch_1ch_u8(d_in, w, h, d_out, mask, mw, mh, nullptr));
    s.download(out1, d_out, n);
    ofx_compat::status() = s.rc();
}

void conv_3ch_1ch_constant(const unsigned char *img3, int w, int h, unsigned char *out1, const float *mask, int mw, int mh)
{
    conv1_u8(img3, w, h, out1, mask, mw, mh, "gpu::conv_3ch_1ch_constant");
}

void conv_3ch_1ch_tiled(const unsigned char *img3, int w, int h, unsigned char *out1, const float *mask, int mw, int mh)
{
    conv1_u8(img3, w, h, out1, mask, mw, mh, "gpu::conv_3ch_1ch_tiled");
}

void conv_3ch_1ch_tiled_uchar_float(const unsigned char *img3, int w, int h, float *out1, const float *mask, int mw, int mh)
{
    if (!args_ok(img3 && out1 && mask && w > 0 && h > 0, "gpu::conv_3ch_1ch_tiled_uchar_float")) return;
    const size_t n = (size_t)w * h;
    Scratch s;
    unsigned char *d_in = s.upload(img3, 3 * n);
    float *d_out = s.alloc<float>(n);
    if (s.ok()) s.run(ofx_conv_3ch_1ch_f32(d_in, w, h, d_out, mask, mw, mh, nullptr));
    s.download(out1, d_out, n);
    ofx_compat::status() = s.rc();
}

void conv_1d_3ch(unsigned char *img3, int w, int h, unsigned char *out3)
{
    if (!args_ok(img3 && out3 && w > 0 && h > 0, "gpu::conv_1d_3ch")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(img3, n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) {
        hipLaunchKernelGGL(conv_1d_3ch_kernel, dim3((unsigned)((w * h + 255) / 256)), dim3(256), 0, nullptr, d_in, d_out, w * h);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            ofx_set_error("gpu::conv_1d_3ch launch: %s", hipGetErrorString(e));
            s.run(OFX_E_HIP);
        }
    }
    s.download(out3, d_out, n);
    ofx_compat::status() = s.rc();
}

void gauss_pyramid(unsigned char **pyramid, int w, int h, int levels, const float *mask, int mw, int mh)
{
    (void)mask; // the reference ignores its mask too: g_gauss_pyramid is hard-wired to GAUS_KERNEL_3x3 (OptFlowGpu.cu:1193)
    (void)mw;
    (void)mh;
    if (!args_ok(pyramid && w > 0 && h > 0 && levels >= 1, "gpu::gauss_pyramid")) return;
    // one upload of level 0, every coarser level produced on the device, one download per level
    Scratch s;
    std::vector<unsigned char *> d(levels, nullptr);
    d[0] = s.upload(pyramid[0], (size_t)w * h * 3);
    for (int k = 1; k < levels && s.ok(); ++k) {
        const int dw = w >> k, dh = h >> k;
        if (dw <= 0 || dh <= 0) {
            ofx_set_error("gpu::gauss_pyramid: level %d is empty", k);
            s.run(OFX_E_INVALID);
            break;
        }
        d[k] = s.alloc<unsigned char>((size_t)dw * dh * 3);
        if (s.ok()) s.run(ofx_downsample_3ch(d[k - 1], d[k], dw, dh, nullptr));
    }
    for (int k = 1; k < levels && s.ok(); ++k) s.download(pyramid[k], d[k], (size_t)(w >> k) * (h >> k) * 3);
    ofx_compat::status() = s.rc();
}

static void srm_u8(const unsigned char *a, const unsigned char *b, int w, int h, int ww, int wh, int *out, const char *who)
{
    if (!args_ok(a && b && out && w > 0 && h > 0, who)) return;
    const size_t n = (size_t)w * h;
    Scratch s;
    unsigned char *d_a = s.upload(a, n), *d_b = (a == b) ? d_a : s.upload(b, n);
    int *d_o = s.alloc<int>(n);
    if (s.ok()) s.run(ofx_srm_u8(d_a, d_b, w, h, ww, wh, d_o, nullptr));
    s.download(out, d_o, n);
    ofx_compat::status() = s.rc();
}

void srm_1ch(const unsigned char *a, const unsigned char *b, int w, int h, int ww, int wh, int *out)
{
    srm_u8(a, b, w, h, ww, wh, out, "gpu::srm_1ch");
}

void srm_1ch_tiled(const unsigned char *a, const unsigned char *b, int w, int h, int ww, int wh, int *out)
{
    srm_u8(a, b, w, h, ww, wh, out, "gpu::srm_1ch_tiled");
}

void srm_1ch_float(const float *a, const float *b, int w, int h, int ww, int wh, float *out)
{
    if (!args_ok(a && b && out && w > 0 && h > 0, "gpu::srm_1ch_float")) return;
    const size_t n = (size_t)w * h;
    Scratch s;
    float *d_a = s.upload(a, n), *d_b = (a == b) ? d_a : s.upload(b, n);
    float *d_o = s.alloc<float>(n);
    if (s.ok()) s.run(ofx_srm_f32(d_a, d_b, w, h, ww, wh, d_o, nullptr));
    s.download(out, d_o, n);
    ofx_compat::status() = s.rc();
}

void inverse_matrix(int *sumIx2, int *sumIy2, int *sumIxIy, int *sumIxIt, int *sumIyIt, float **optFlowPyramid, int level, int w, int h)
{
    if (!args_ok(sumIx2 && sumIy2 && sumIxIy && sumIxIt && sumIyIt && optFlowPyramid && level >= 0 && optFlowPyramid[level] && w > 0 && h > 0,
                 "gpu::inverse_matrix"))
        return;
    const size_t n = (size_t)w * h;
    Scratch s;
    int *xx = s.upload(sumIx2, n), *yy = s.upload(sumIy2, n), *xy = s.upload(sumIxIy, n), *xt = s.upload(sumIxIt, n),
        *yt = s.upload(sumIyIt, n);
    float *d_f = s.alloc<float>(2 * n);
    if (s.ok()) s.run(ofx_solve_i32(xx, yy, xy, xt, yt, d_f, w, h, OFX_SOLVE_F64, nullptr));
    s.download(optFlowPyramid[level], d_f, 2 * n);
    ofx_compat::status() = s.rc();
}

void inverse_matrix_float(float *sumIx2, float *sumIy2, float *sumIxIy, float *sumIxIt, float *sumIyIt, float **optFlowPyramid, int level, int w,
                          int h)
{
    if (!args_ok(sumIx2 && sumIy2 && sumIxIy && sumIxIt && sumIyIt && optFlowPyramid && level >= 0 && optFlowPyramid[level] && w > 0 && h > 0,
                 "gpu::inverse_matrix_float"))
        return;
    const size_t n = (size_t)w * h;
    Scratch s;
    float *xx = s.upload(sumIx2, n), *yy = s.upload(sumIy2, n), *xy = s.upload(sumIxIy, n), *xt = s.upload(sumIxIt, n),
          *yt = s.upload(sumIyIt, n);
    float *d_f = s.alloc<float>(2 * n);
    if (s.ok()) s.run(ofx_solve_f32(xx, yy, xy, xt, yt, d_f, w, h, nullptr));
    s.download(optFlowPyramid[level], d_f, 2 * n);
    ofx_compat::status() = s.rc();
}

void calc_opt_flow(const unsigned char *prev3, unsigned char *next3, int w, int h, float **optFlowPyramid, int level, int maxLevel)
{
    // window 19x19 and the Dt_3x3 temporal mask are the reference's constants (OptFlowGpu.cu:1936-1945)
    ofx_compat::status() = ofx_calc_opt_flow_host(prev3, next3, w, h, optFlowPyramid, level, maxLevel, 19, OFX_MODE_LK_FLOAT);
}

void bilinear_filter(unsigned char *img3, unsigned char *gray3, unsigned char *out3, int w, int h, int ww, int wh, double sigmaS, double sigmaB)
{
    if (!args_ok(img3 && gray3 && out3 && w > 0 && h > 0, "gpu::bilinear_filter")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(img3, n), *d_g = (gray3 == img3) ? d_in : s.upload(gray3, n), *d_out = s.alloc<unsigned char>(n);
    // (the reference's signature has no mode: ofx_bilateral_wrappers_fast / OFX_BILATERAL_FAST select the +-1 LSB kernel)
    const bool fast = ofx_bilateral_wrappers_fast(-1) != 0 && (ww & 1) && (wh & 1) && ww <= 13 && wh <= ww;
    if (s.ok()) s.run(fast ? ofx_bilateral_3ch_fast(d_in, d_g, d_out, w, h, ww, wh, sigmaS, sigmaB, nullptr)
                           : ofx_bilateral_3ch(d_in, d_g, d_out, w, h, ww, wh, sigmaS, sigmaB, nullptr));
    s.download(out3, d_out, n);
    ofx_compat::status() = s.rc();
}

} // namespace gpu

// ---- utils (host helpers; reference OptFlowUtils.cpp) --------------------------------------------------------------
namespace utils {

void cleanup_outliers(unsigned char *img1, int w, int h)
{
    // OptFlowUtils.cpp:5-19
    const size_t n = (size_t)w * h;
    for (size_t p = 0; p < n; ++p) img1[p] = (img1[p] >= 240 || img1[p] < 20) ? 0 : 255;
}

template <int CH>
static void upscale(const unsigned char *src, int w, int h, int n, unsigned char *dst)
{
    // OptFlowUtils.cpp:21-61: every source pixel becomes a 2^n x 2^n block
    const int f = 1 << n;
    const size_t ow = (size_t)w * f;
    for (size_t oy = 0; oy < (size_t)h * f; ++oy)
        for (size_t ox = 0; ox < ow; ++ox) {
            const unsigned char *s = src + CH * ((oy >> n) * (size_t)w + (ox >> n));
            unsigned char *d = dst + CH * (oy * ow + ox);
            for (int c = 0; c < CH; ++c) d[c] = s[c];
        }
}

void upscale_3ch(unsigned char *img3, int w, int h, int n, unsigned char *out3) { upscale<3>(img3, w, h, n, out3); }
void upscale_1ch(unsigned char *img1, int w, int h, int n, unsigned char *out1) { upscale<1>(img1, w, h, n, out1); }

void generate_gaussian_kernel(double sigmaS, int kernel_size, double *dest) { ofx_generate_gaussian_kernel(sigmaS, kernel_size, dest); }

} // namespace utils
