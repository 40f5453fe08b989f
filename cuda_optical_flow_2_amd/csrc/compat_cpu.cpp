// namespace cpu: the reference's CPU call surface (include/OptFlowCpu.hpp, reference OptFlowCpu.hpp:3-184) as host-pointer
// wrappers over the device kernels of this library.
//
// main.cu treats the cpu:: functions as line-for-line alternatives of the gpu:: ones (main.cu:199,239,248-251,261) and calls
// cpu::sub_arr outright (main.cu:64), so a drop-in has to export them.  They are NOT a CPU fallback: every function stages
// its host buffers in HBM and runs on the MI355X like its gpu:: twin -- with the cpu:: functions' own arithmetic where
// the two differ (cpu::gauss_pyramid honours its mask, cpu::inverse_matrix solves in float, cpu::calc_optical_flow is the
// bug-for-bug compat_cpu mode with its 9x9 window).  Contract as in compat_gpu.cpp: host pointers owned by the caller,
// synchronous, void returns, failures readable through gpu_compat_last_status() / ofx_last_error().
#include <vector>

#include "OptFlowCpu.hpp"
#include "compat_scratch.h"
#include "ofx_internal.h"

using ofx_compat::args_ok;
using ofx_compat::Scratch;
using ofx_compat::status;

namespace cpu {

void sub_arr(unsigned char *arr1, unsigned char *arr2, int n, unsigned char *dest)
{
    if (!args_ok(arr1 && arr2 && dest && n >= 0, "cpu::sub_arr")) return;
    Scratch s;
    unsigned char *d_a = s.upload(arr1, (size_t)n), *d_b = (arr2 == arr1) ? d_a : s.upload(arr2, (size_t)n);
    unsigned char *d_o = s.alloc<unsigned char>((size_t)n);
    if (s.ok()) s.run(ofx_sub_u8(d_a, d_b, (size_t)n, d_o, nullptr));
    s.download(dest, d_o, (size_t)n);
    status() = s.rc();
}

void grayscale_avg_cpu(const unsigned char *src, unsigned char *dest, int w, int h)
{
    if (!args_ok(src && dest && w > 0 && h > 0, "cpu::grayscale_avg_cpu")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(src, n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_grayscale_avg_3ch(d_in, d_out, w, h, nullptr));
    s.download(dest, d_out, n);
    status() = s.rc();
}

void conv_3ch(const unsigned char *src, const float *mask, unsigned char *dest, int w, int h, int mw, int mh)
{
    if (!args_ok(src && mask && dest && w > 0 && h > 0, "cpu::conv_3ch")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(src, n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_conv_3ch(d_in, d_out, w, h, mask, mw, mh, 0, nullptr)); // int accumulators, per-tap truncation (:62)
    s.download(dest, d_out, n);
    status() = s.rc();
}

void conv_3ch_to_1ch(const unsigned char *src, int w, int h, unsigned char *dest, const float *mask, int mw, int mh)
{
    if (!args_ok(src && mask && dest && w > 0 && h > 0, "cpu::conv_3ch_to_1ch")) return;
    const size_t n = (size_t)w * h;
    Scratch s;
    unsigned char *d_in = s.upload(src, 3 * n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_conv_3ch_1ch_u8(d_in, w, h, d_out, mask, mw, mh, nullptr));
    s.download(dest, d_out, n);
    status() = s.rc();
}

void downscale_gaussian(unsigned char *src, int w, int h, unsigned char *dest, const float *mask, int mw, int mh)
{
    // (w, h) is the DESTINATION size; the source is 2w x 2h (OptFlowCPU.cpp:117-118)
    if (!args_ok(src && mask && dest && w > 0 && h > 0, "cpu::downscale_gaussian")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(src, 4 * n), *d_out = s.alloc<unsigned char>(n);
    if (s.ok()) s.run(ofx_downscale_mask_3ch(d_in, d_out, w, h, mask, mw, mh, nullptr));
    s.download(dest, d_out, n);
    status() = s.rc();
}

void gauss_pyramid(unsigned char **pyramid, int w, int h, int n, const float *mask, int mw, int mh)
{
    if (!args_ok(pyramid && mask && w > 0 && h > 0 && n >= 1, "cpu::gauss_pyramid")) return;
    // one upload of level 0, every coarser level produced on the device from the one before, one download per level
    Scratch s;
    std::vector<unsigned char *> d((size_t)n, nullptr);
    d[0] = s.upload(pyramid[0], (size_t)w * h * 3);
    for (int k = 1; k < n && s.ok(); ++k) {
        const int dw = w >> k, dh = h >> k;
        if (dw <= 0 || dh <= 0) {
            ofx_set_error("cpu::gauss_pyramid: level %d is empty", k);
            s.run(OFX_E_INVALID);
            break;
        }
        d[k] = s.alloc<unsigned char>((size_t)dw * dh * 3);
        if (s.ok()) s.run(ofx_downscale_mask_3ch(d[k - 1], d[k], dw, dh, mask, mw, mh, nullptr));
    }
    for (int k = 1; k < n && s.ok(); ++k) s.download(pyramid[k], d[k], (size_t)(w >> k) * (h >> k) * 3);
    status() = s.rc();
}

void srm_1ch(const unsigned char *arr1, const unsigned char *arr2, int w, int h, int ww, int wh, int *dest)
{
    if (!args_ok(arr1 && arr2 && dest && w > 0 && h > 0, "cpu::srm_1ch")) return;
    const size_t n = (size_t)w * h;
    Scratch s;
    unsigned char *d_a = s.upload(arr1, n), *d_b = (arr1 == arr2) ? d_a : s.upload(arr2, n);
    int *d_o = s.alloc<int>(n);
    if (s.ok()) s.run(ofx_srm_u8(d_a, d_b, w, h, ww, wh, d_o, nullptr));
    s.download(dest, d_o, n);
    status() = s.rc();
}

void srm_3ch(unsigned char *arr1, unsigned char *arr2, int w, int h, int ww, int wh, int *dest)
{
    if (!args_ok(arr1 && arr2 && dest && w > 0 && h > 0, "cpu::srm_3ch")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_a = s.upload(arr1, n), *d_b = (arr1 == arr2) ? d_a : s.upload(arr2, n);
    int *d_o = s.alloc<int>(n);
    if (s.ok()) s.run(ofx_srm_3ch_u8(d_a, d_b, w, h, ww, wh, d_o, nullptr));
    s.download(dest, d_o, n);
    status() = s.rc();
}

void shift_back_pyramid(const unsigned char *src, int w, int h, int level, int maxLevel, float **optFlowPyramid, unsigned char *dest)
{
    if (!args_ok(src && dest && optFlowPyramid && w > 0 && h > 0 && level >= 0 && level < maxLevel && maxLevel <= OFX_MAX_LEVELS,
                 "cpu::shift_back_pyramid"))
        return;
    // the translation: float accumulation over the coarser levels' PIXEL 0, coarsest first (OptFlowCPU.cpp:255-266, where
    // `i * (1 >> offset)` is 0 for every offset >= 1)
    float uv[2] = {0.0f, 0.0f};
    for (int k = maxLevel - 1; k > level; --k) {
        if (!args_ok(optFlowPyramid[k] != nullptr, "cpu::shift_back_pyramid (a coarser flow level)")) return;
        const float mult = (float)(1 << (k - level));
        uv[0] += mult * optFlowPyramid[k][0];
        uv[1] += mult * optFlowPyramid[k][1];
    }
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(src, n);
    unsigned char *d_out = s.upload(dest, n); // pixels whose target leaves the image keep what the caller's buffer held
    float *d_uv = s.upload(uv, 2);
    if (s.ok()) s.run(ofx_shift_3ch(d_in, d_out, w, h, d_uv, nullptr));
    s.download(dest, d_out, n);
    status() = s.rc();
}

void inverse_matrix(int *sumIx2, int *sumIy2, int *sumIxIy, int *sumIxIt, int *sumIyIt, float **optFlowPyramid, int level, int w, int h)
{
    if (!args_ok(sumIx2 && sumIy2 && sumIxIy && sumIxIt && sumIyIt && optFlowPyramid && level >= 0 && optFlowPyramid[level] && w > 0 && h > 0,
                 "cpu::inverse_matrix"))
        return;
    const size_t n = (size_t)w * h;
    Scratch s;
    int *xx = s.upload(sumIx2, n), *yy = s.upload(sumIy2, n), *xy = s.upload(sumIxIy, n), *xt = s.upload(sumIxIt, n),
        *yt = s.upload(sumIyIt, n);
    float *d_f = s.alloc<float>(2 * n);
    if (s.ok()) s.run(ofx_solve_i32(xx, yy, xy, xt, yt, d_f, w, h, OFX_SOLVE_F32, nullptr)); // float arithmetic, OptFlowCPU.cpp:293-304
    s.download(optFlowPyramid[level], d_f, 2 * n);
    status() = s.rc();
}

void calc_optical_flow(const unsigned char *prev, unsigned char *next, int w, int h, float **optFlowPyramid, int level, int maxLevel)
{
    // window 9x9, wrapped-u8 derivatives, Gaussian It and the inline solve are the reference's (OptFlowCPU.cpp:329-384)
    status() = ofx_calc_opt_flow_host(prev, next, w, h, optFlowPyramid, level, maxLevel, 9, OFX_MODE_COMPAT_CPU);
}

void bilinear_filter_3ch(unsigned char *src, unsigned char *gray, unsigned char *dest, int w, int h, int ww, int wh, double sigmaS, double sigmaB)
{
    if (!args_ok(src && gray && dest && w > 0 && h > 0, "cpu::bilinear_filter_3ch")) return;
    const size_t n = (size_t)w * h * 3;
    Scratch s;
    unsigned char *d_in = s.upload(src, n), *d_g = (gray == src) ? d_in : s.upload(gray, n), *d_out = s.alloc<unsigned char>(n);
    // (the reference's signature has no mode: ofx_bilateral_wrappers_fast / OFX_BILATERAL_FAST select the +-1 LSB kernel)
    const bool fast = ofx_bilateral_wrappers_fast(-1) != 0 && (ww & 1) && (wh & 1) && ww <= 13 && wh <= ww;
    if (s.ok()) s.run(fast ? ofx_bilateral_3ch_fast(d_in, d_g, d_out, w, h, ww, wh, sigmaS, sigmaB, nullptr)
                           : ofx_bilateral_3ch(d_in, d_g, d_out, w, h, ww, wh, sigmaS, sigmaB, nullptr));
    s.download(dest, d_out, n);
    status() = s.rc();
}

} // namespace cpu
