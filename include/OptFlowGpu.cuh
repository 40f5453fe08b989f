// Drop-in declaration of the reference's GPU call surface (namespace gpu), implemented by libofx_hip.so on MI355X.
//
// The signatures are the reference's (its OptFlowGpu.cuh:5-35) because main.cu calls them; everything behind them
// is new: each wrapper validates its arguments, stages the host buffers in HBM and runs the hand-written HIP kernels
// of this repo through the C ABI in ofx.h (cuda_optical_flow_2_amd/csrc/compat_gpu.cpp).  All pointers are HOST
// pointers owned by the caller, calls are synchronous, nothing is retained -- as in the reference.  Functions return
// void there, so failures are reported through ofx_last_error() / gpu_compat_last_status() instead.
//
// Differences from the reference's behaviour, all of them places where the reference is broken:
//   * every launch covers the whole image (the reference swaps grid and block at nine call sites and silently
//     computes nothing above 640x480);
//   * inverse_matrix / inverse_matrix_float write every pixel (the reference's integer-division grid skips the
//     last partial 32x32 blocks);
//   * calc_opt_flow's scratch image is zero-initialised (the reference reads uninitialised malloc memory).
#pragma once

namespace gpu {

// (r+g+b)/3 replicated into three channels.  Note the (rows, cols) argument order.
void grayscale_avg(const unsigned char *rgb, unsigned char *gray3, int rows, int cols);

// 3-channel correlation with an mw x mh mask, zero padding by skipping, per-tap integer truncation.
void conv_3ch_2d(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh);
void conv_3ch_2d_constant(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh);
// Same stencil with float accumulators (the reference's tiled kernel accumulates in float).
void conv_3ch_tiled(const unsigned char *img3, unsigned char *out3, int w, int h, const float *mask, int mw, int mh);

// Channel 0 -> one u8 plane; integer accumulator, result wraps modulo 256.
void conv_3ch_1ch_constant(const unsigned char *img3, int w, int h, unsigned char *out1, const float *mask, int mw, int mh);
void conv_3ch_1ch_tiled(const unsigned char *img3, int w, int h, unsigned char *out1, const float *mask, int mw, int mh);
// Channel 0 -> one float plane; float accumulator, no rounding or wrap.
void conv_3ch_1ch_tiled_uchar_float(const unsigned char *img3, int w, int h, float *out1, const float *mask, int mw, int mh);

// The reference's 9-tap 1-D practice kernel; kept for link compatibility (horizontal box of 9 over the byte stream).
void conv_1d_3ch(unsigned char *img3, int w, int h, unsigned char *out3);

// pyramid[k] (k = 1..levels-1) from pyramid[k-1]: 2x decimation with the fixed 3x3 Gaussian.  As in the reference
// the mask arguments are ignored.
void gauss_pyramid(unsigned char **pyramid, int w, int h, int levels, const float *mask, int mw, int mh);

// Windowed sum of a*b, window clipped at the image border.
void srm_1ch(const unsigned char *a, const unsigned char *b, int w, int h, int ww, int wh, int *out);
void srm_1ch_float(const float *a, const float *b, int w, int h, int ww, int wh, float *out);
void srm_1ch_tiled(const unsigned char *a, const unsigned char *b, int w, int h, int ww, int wh, int *out);

// Per-pixel 2x2 solve in double; writes interleaved (u,v) into optFlowPyramid[level].
void inverse_matrix(int *sumIx2, int *sumIy2, int *sumIxIy, int *sumIxIt, int *sumIyIt, float **optFlowPyramid, int level, int w, int h);
void inverse_matrix_float(float *sumIx2, float *sumIy2, float *sumIxIy, float *sumIxIt, float *sumIyIt, float **optFlowPyramid, int level, int w, int h);

// One pyramid level of dense Lucas-Kanade (window 19x19, Dt_3x3 temporal mask, double solve).
void calc_opt_flow(const unsigned char *prev3, unsigned char *next3, int w, int h, float **optFlowPyramid, int level, int maxLevel);

// Bilateral filter (the reference's name for it).
void bilinear_filter(unsigned char *img3, unsigned char *gray3, unsigned char *out3, int w, int h, int ww, int wh, double sigmaS, double sigmaB);

} // namespace gpu

// 0 when the calling thread's last gpu:: call succeeded, otherwise the OFX_E_* code (message: ofx_last_error()).
extern "C" int gpu_compat_last_status(void);
