/*
 * ofx_oracle.h -- CPU ORACLE for the dense pyramidal Lucas-Kanade path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library under
 * cuda_optical_flow_2_amd/csrc, the C-ABI in include/ofx.h, the gpu:: compat
 * surface) may include, link or call this.  Allowed users: tests/,
 * __graft_entry__.smoke(), and bench.py's cpu_baseline leg.
 *
 * It is a plain-C restatement of the reference's algorithm.  Every function
 * cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Parity status: PINNED.  The reference ships no golden vectors or tests
 * (SURVEY.md section 4), so this restatement is pinned against the reference's
 * own CPU sources compiled in the build container (oracle/Makefile target
 * `ref`, output oracle/_ref/libref_cpu.so) by tests/test_oracle_vs_ref.py, and
 * against fixtures generated from that build (tests/golden/, generator
 * tests/golden/make_golden.py).  The float ("GPU semantics") functions follow
 * OptFlowGpu.cu, which cannot be compiled or run here (CUDA); they are pinned
 * only by the known-answer vectors in SURVEY.md section 8c and by their shared
 * integer sub-steps.
 *
 * Image layout conventions are the reference's: "3ch" = HWC interleaved
 * unsigned char with 3 bytes per pixel; "1ch" = HW plane; flow = interleaved
 * (u,v) float pairs, 2*w*h floats per level.
 */
#ifndef OFX_ORACLE_H
#define OFX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stencil tables, same values as kernels.cpp:6-64 */
extern const float orc_Dx_3x3[9];
extern const float orc_Dy_3x3[9];
extern const float orc_Dt_3x3[9];
extern const float orc_GAUS_3x3[9];

/* ---- element-wise ------------------------------------------------------ */
/* OptFlowCPU.cpp:11-17 */
void orc_sub_u8(const uint8_t *a, const uint8_t *b, int n, uint8_t *dst);
/* OptFlowUtils.hpp:21-31 */
void orc_sub_f32(const float *a, const float *b, int n, float *dst);
/* OptFlowCPU.cpp:19-31 */
void orc_grayscale_avg(const uint8_t *src3, uint8_t *dst3, int w, int h);

/* ---- small correlations -------------------------------------------------- */
/* OptFlowCPU.cpp:33-73 : 3ch -> 3ch, int accumulator truncated after every tap */
void orc_conv_3ch(const uint8_t *src3, const float *mask, uint8_t *dst3, int w, int h, int mw, int mh);
/* OptFlowCPU.cpp:75-109 : channel 0 -> 1ch u8, int accumulator, wrap mod 256 */
void orc_conv_3ch_to_1ch(const uint8_t *src3, int w, int h, uint8_t *dst, const float *mask, int mw, int mh);
/* OptFlowGpu.cu:1040-1090 : channel 0 -> 1ch f32, float accumulator, zero taps skipped */
void orc_conv_3ch_to_1ch_f32(const uint8_t *src3, int w, int h, float *dst, const float *mask, int mw, int mh);

/* ---- pyramid ------------------------------------------------------------ */
/* OptFlowCPU.cpp:112-148 : (w,h) are the DESTINATION dims, source is 2w x 2h */
void orc_downscale_gaussian(const uint8_t *src3, int w, int h, uint8_t *dst3, const float *mask, int mw, int mh);
/* OptFlowCPU.cpp:151-160 */
void orc_gauss_pyramid(uint8_t **pyr3, int w, int h, int n, const float *mask, int mw, int mh);

/* ---- window sums of products ------------------------------------------- */
/* OptFlowCPU.cpp:162-200 (== OptFlowGpu.cu:1463-1502) */
void orc_srm_1ch(const uint8_t *a, const uint8_t *b, int w, int h, int ww, int wh, int32_t *dst);
/* OptFlowGpu.cu:1549-1588 : float accumulator, taps visited row-major */
void orc_srm_1ch_f32(const float *a, const float *b, int w, int h, int ww, int wh, float *dst);
/* same window, but accumulated exactly (double) and rounded ONCE to float.
 * Not a reference function: it is the order-independent definition the HIP
 * path implements for integer-valued derivative planes. */
void orc_srm_1ch_f32_exact(const float *a, const float *b, int w, int h, int ww, int wh, float *dst);

/* ---- global shift ("warp") ---------------------------------------------- */
/* OptFlowCPU.cpp:241-282.  dst3 must be pre-initialised by the caller; the
 * reference leaves it uninitialised (malloc) and the pinned behaviour is
 * all-zero (zero-fill shim). */
void orc_shift_back_pyramid(const uint8_t *src3, int w, int h, int level, int max_level,
                            float *const *flow_pyr, uint8_t *dst3);

/* ---- 2x2 solve, three reference variants -------------------------------- */
/* (ii) OptFlowCPU.cpp:285-309 : float arithmetic */
void orc_inverse_matrix_f32arith(const int32_t *sxx, const int32_t *syy, const int32_t *sxy,
                                 const int32_t *sxt, const int32_t *syt, float *flow, int w, int h);
/* (iii) OptFlowGpu.cu:1727-1755 (int sums) : double arithmetic, correct */
void orc_inverse_matrix_i32(const int32_t *sxx, const int32_t *syy, const int32_t *sxy,
                            const int32_t *sxt, const int32_t *syt, float *flow, int w, int h);
/* (iii) OptFlowGpu.cu:1819-1846 (float sums) */
void orc_inverse_matrix_f32(const float *sxx, const float *syy, const float *sxy,
                            const float *sxt, const float *syt, float *flow, int w, int h);
/* (i) OptFlowCPU.cpp:363-384 : double arithmetic, c NOT scaled by the prefix */
void orc_inverse_matrix_inline_cpu(const int32_t *sxx, const int32_t *syy, const int32_t *sxy,
                                   const int32_t *sxt, const int32_t *syt, float *flow, int w, int h);

/* ---- per-level compositions --------------------------------------------- */
/* OptFlowCPU.cpp:312-399 with the window made a parameter (reference: 9).
 * "compat_cpu" mode of the engine. */
void orc_calc_optical_flow_cpu(const uint8_t *prev3, const uint8_t *next3, int w, int h,
                               float **flow_pyr, int level, int max_level, int window);
/* OptFlowGpu.cu:1909-1979 with the window made a parameter (reference: 19).
 * "lk_float" mode of the engine.  exact_sums != 0 selects orc_srm_1ch_f32_exact. */
void orc_calc_opt_flow_gpu(const uint8_t *prev3, const uint8_t *next3, int w, int h,
                           float **flow_pyr, int level, int max_level, int window, int exact_sums);

/* intermediate planes of one level, for stage-by-stage parity tests.  Any
 * pointer may be NULL.  mode 0 = compat_cpu (u8 derivatives widened to float
 * for inspection, int sums widened), mode 1 = lk_float. */
void orc_level_planes(const uint8_t *prev3, const uint8_t *next3_shifted, int w, int h, int window, int mode,
                      int exact_sums, float *ix, float *iy, float *it, double *sums5);

/* main.cu:138-147 : flow composed down to `level`, dense */
void orc_compose_flow(float *const *flow_pyr, int w, int h, int levels, int level, float *dst_uv);

/* ---- bilateral pre-filter ("bilinear_filter" in the reference) ---------- */
/* OptFlowUtils.cpp:68-114 */
void orc_generate_gaussian_kernel(double sigma_s, int kernel_size, double *dst);
/* OptFlowCPU.cpp:401-465 */
void orc_bilateral_3ch(const uint8_t *src3, const uint8_t *gray3, uint8_t *dst3, int w, int h,
                       int ww, int wh, double sigma_s, double sigma_b);

/* ---- extension: iterative refinement with bilinear warp (NO reference twin) ------------------------
 * Defined by DESIGN.md section "lk_iter"; pinned only by this restatement and by known-answer translations.
 *   iteration 1   = the reference level (orc_calc_opt_flow_gpu semantics on the globally shifted next image)
 *   iteration i>1 : W = round_u8(bilinear(next', x + s*u, y + s*v)), s = 8/15 (Sobel gain 8 / Dt_3x3 gain 15),
 *                   F += level(prev, W)  with exact window sums
 * The shift vectors between levels stay the reference's (pixel 0 of the FIRST iteration's flow). */
#define ORC_ITER_SCALE 0.533333361148834228515625f /* (float)(8.0/15.0) */
void orc_warp_bilinear_u8(const uint8_t *src1, int w, int h, const float *flow_uv, float scale, uint8_t *dst1);
void orc_lk_iter_level(const uint8_t *prev1, const uint8_t *next1_shifted, int w, int h, int window, int iters,
                       float *flow_uv /* out: accumulated */, float *flow_first /* out, may be NULL: iteration 1 only */);

/* ---- helpers for 1-channel pipelines (layout only, no arithmetic) ------- */
void orc_replicate_1ch_to_3ch(const uint8_t *src1, uint8_t *dst3, int n);
void orc_extract_ch0(const uint8_t *src3, uint8_t *dst1, int n);

/* link-compat leftovers */
/* cpu::srm_3ch, OptFlowCPU.cpp:202-238 (positions past the end of the buffer contribute nothing) */
void orc_srm_3ch(const uint8_t *a3, const uint8_t *b3, int w, int h, int ww, int wh, int32_t *dst3);
/* utils::cleanup_outliers, OptFlowUtils.cpp:5-19 */
void orc_cleanup_outliers(uint8_t *img1, int w, int h);
/* utils::upscale_3ch / upscale_1ch, OptFlowUtils.cpp:21-61 */
void orc_upscale(const uint8_t *src, int w, int h, int n, int channels, uint8_t *dst);
/* gpu::conv_1d_3ch, OptFlowGpu.cu:1134-1189 (taps outside the buffer skipped) */
void orc_conv_1d_3ch(const uint8_t *src3, int w, int h, uint8_t *dst3);

#ifdef __cplusplus
}
#endif
#endif
