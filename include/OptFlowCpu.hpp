// Drop-in declaration of the reference's CPU call surface (namespace cpu, its OptFlowCpu.hpp:3-184), implemented by
// libofx_hip.so.  main.cu includes this header (main.cu:2), calls cpu::sub_arr (main.cu:64) and keeps the other cpu::
// functions as commented-out alternatives of the gpu:: calls (main.cu:199,239,248-251,261): with this header and the
// library those lines can be swapped in and out as in the reference.
//
// Signatures are the reference's; the implementation is not its CPU code: every function stages the caller's host
// buffers in HBM and runs hand-written HIP kernels on the MI355X (cuda_optical_flow_2_amd/csrc/compat_cpu.cpp),
// reproducing the arithmetic of the reference's cpu:: functions bit for bit -- including where it differs from the
// gpu:: twins (gauss_pyramid honours its mask; inverse_matrix solves in float; calc_optical_flow uses the 9x9 window,
// wrapped 8-bit derivatives and the unscaled `c` of OptFlowCPU.cpp:312-399).  All pointers are HOST pointers owned by the
// caller, calls are synchronous, nothing is retained.  The functions return void, so failures are reported through
// gpu_compat_last_status() / ofx_last_error().
#pragma once

namespace cpu {

// dest[i] = arr1[i] - arr2[i] on bytes (wraps modulo 256).
void sub_arr(unsigned char *arr1, unsigned char *arr2, int n, unsigned char *dest);

// (c0 + c1 + c2) / 3 written to all three channels.  3-channel interleaved 8-bit images of w*h*3 bytes.
void grayscale_avg_cpu(const unsigned char *src, unsigned char *dest, int w, int h);

// 3-channel correlation with an mw x mh mask; taps outside the image are skipped; the accumulator is an int that is
// truncated after every tap; the result wraps modulo 256.
void conv_3ch(const unsigned char *src, const float *mask, unsigned char *dest, int w, int h, int mw, int mh);

// The same stencil on channel 0 only, one 8-bit output plane (src is expected to be grey).
void conv_3ch_to_1ch(const unsigned char *src, int w, int h, unsigned char *dest, const float *mask, int mw, int mh);

// One pyramid step: (w, h) is the size of dest, src is 2w x 2h; dest(x,y) = mask applied around src(2x, 2y), float
// accumulators, truncated to 8 bits.
void downscale_gaussian(unsigned char *src, int w, int h, unsigned char *dest, const float *mask, int mw, int mh);

// pyramid[i] from pyramid[i-1] for i = 1..n-1 with downscale_gaussian; the caller allocates every level
// ((w >> i) x (h >> i) x 3 bytes).
void gauss_pyramid(unsigned char **pyramid, int w, int h, int n, const float *mask, int mw, int mh);

// dest(x,y) = sum over the ww x wh window around (x,y), clipped to the image, of arr1 * arr2 (one channel).
void srm_1ch(const unsigned char *arr1, const unsigned char *arr2, int w, int h, int ww, int wh, int *dest);

// The same per channel of 3-channel inputs, 3 ints per pixel.  (The reference's window test admits one column / row
// past the far edges; positions past the end of the buffer contribute nothing here.)
void srm_3ch(unsigned char *arr1, unsigned char *arr2, int w, int h, int ww, int wh, int *dest);

// Shift `src` back by the flow found on the coarser levels (levels level+1 .. maxLevel-1 of optFlowPyramid must be
// filled).  As in the reference the shift is one translation formed from pixel 0 of those levels, only the first w*h
// bytes of dest are initialised from src, and pixels whose target leaves the image keep what dest held.
void shift_back_pyramid(const unsigned char *src, int w, int h, int level, int maxLevel, float **optFlowPyramid, unsigned char *dest);

// Per-pixel 2x2 solve in float; writes interleaved (u,v) into optFlowPyramid[level].
void inverse_matrix(int *sumIx2, int *sumIy2, int *sumIxIy, int *sumIxIt, int *sumIyIt, float **optFlowPyramid, int level, int w, int h);

// One pyramid level of dense Lucas-Kanade the way the reference's CPU path computes it (9x9 window).
void calc_optical_flow(const unsigned char *prev, unsigned char *next, int w, int h, float **optFlowPyramid, int level, int maxLevel);

// Bilateral filter (spatial Gaussian sigmaS of size ww, range Gaussian sigmaB on the channel-0 difference of `gray`).
void bilinear_filter_3ch(unsigned char *src, unsigned char *gray, unsigned char *dest, int w, int h, int ww, int wh, double sigmaS, double sigmaB);

} // namespace cpu
