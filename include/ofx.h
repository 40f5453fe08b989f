/*
 * ofx.h -- C ABI of the MI355X dense pyramidal Lucas-Kanade engine (libofx_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path: plain pointers and
 * sizes, no C++ or torch types.  The reference exposes the same path as C++
 * free functions (OptFlowGpu.cuh:5-35); include/OptFlowGpu.cuh in this repo
 * re-declares that surface and implements it on top of the entry points below
 * (INTEGRATION.md shows the binding).  Every entry point names the reference
 * interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, an OFX_E_* code otherwise, and never
 *     throws; ofx_last_error() returns the message of the calling thread's last
 *     failure (the reference returns void and checks nothing).
 *   - "d_" pointers are DEVICE pointers, "h_" pointers are HOST pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Device
 *     entry points only enqueue work; they never synchronise or allocate.
 *   - "3ch" images are HWC interleaved u8, 3 bytes per pixel, tightly packed
 *     (the reference's layout, main.cu:95-104).  "1ch" planes are u8 with a row
 *     pitch in bytes that is a multiple of 4 and >= w.
 *   - flow is interleaved (u,v) float32, 2*w floats per row, tightly packed
 *     (OptFlowGpu.cu:1844-1845).
 *   - window sizes are the reference's ww/wh (full size, not radius).
 */
#ifndef OFX_H
#define OFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFX_OK 0
#define OFX_E_INVALID 1     /* bad argument (size, alignment, unsupported window ...) */
#define OFX_E_HIP 2         /* a HIP runtime call failed; message holds hipGetErrorString */
#define OFX_E_UNSUPPORTED 3 /* valid request this build does not implement */
#define OFX_E_STATE 4       /* session used out of order */

/* which reference semantics a flow level follows */
#define OFX_MODE_COMPAT_CPU 0 /* cpu::calc_optical_flow, OptFlowCPU.cpp:312-399: wrapped-u8 derivatives,
                                 Gaussian It, inline solve (c unscaled) */
#define OFX_MODE_LK_FLOAT 1   /* gpu::calc_opt_flow, OptFlowGpu.cu:1909-1979: float derivatives, Dt_3x3,
                                 double solve */

#define OFX_MODE_LK_FLOAT_FAST 2 /* lk_float with the solve in its <= 1 ulp(float) formulation (SURVEY 8c's tolerance for the
                                    solve; identical NaN / Inf positions) instead of the replay of the reference's
                                    operation order: numerators first, reciprocal to 2^-44 -- 11 double operations per
                                    pixel instead of 19.  Everything else (window sums, shift, pyramid) is bit-exact. */

#define OFX_MAX_LEVELS 12
#ifndef OFX_MAX_LK_ITEMS /* (build experiments only: the shipped library and this header must agree) */
#define OFX_MAX_LK_ITEMS 80
#endif /* (level, pair) items one fused LK launch can carry (ofx_lk_levels, ofx_stream_launch) */

const char *ofx_last_error(void);
/* library/ABI version, bumped when a signature changes */
int ofx_abi_version(void);
/* number of visible HIP devices, or a negative OFX_E code */
int ofx_device_count(void);

/* ------------------------------------------------------------------------
 * Geometry of one pyramid level as seen by one rank.  Unsharded use:
 * row0 = 0, rows = h, out_y0 = 0, out_y1 = h.  Row-sharded use:
 * the plane buffers hold global rows [row0, row0+rows) -- the rank's block plus
 * halo -- and the kernels produce global rows [out_y0, out_y1).  Rows outside
 * [0,h) are the image border (taps skipped, as the reference does); rows inside
 * [0,h) that a stencil needs must be present in the buffer or the call fails
 * with OFX_E_INVALID.
 * ---------------------------------------------------------------------- */
typedef struct ofx_geom {
    int w, h;   /* global width / height of this level */
    int pitch;  /* bytes per row of the 1ch planes (multiple of 4, >= w) */
    int row0;   /* global row index held in buffer row 0 */
    int rows;   /* rows present in the buffers */
    int out_y0; /* first global row to produce */
    int out_y1; /* one past the last global row to produce */
} ofx_geom;

/* ---- hot path, device-resident ------------------------------------------ */

/* One level of dense LK, fully fused: 3x3 derivative stencils, the five
 * windowed sums of products and the 2x2 solve in one kernel.
 * Replaces the device work of gpu::calc_opt_flow (OptFlowGpu.cu:1930-1964) /
 * cpu::calc_optical_flow steps 1-3 (OptFlowCPU.cpp:329-384) for one level.
 * d_next must already be shifted (ofx_shift_1ch) when level != top (or use ofx_lk_levels with d_uv).
 * d_flow receives rows [out_y0,out_y1) at row offset (y - flow_row0).
 * window: odd, 3..23 (lk_float) / 3..25 (compat_cpu). */
int ofx_lk_level(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode,
                 float *d_flow, int flow_row0, void *stream);

/* Several levels in ONE launch (the levels of a pyramid are independent once the shift vectors are known -- see
 * ofx_corner_flows).  Descriptors are processed in the order given; coarse levels first is the useful order. */
typedef struct ofx_lk_desc {
    const uint8_t *d_prev;
    const uint8_t *d_next;
    ofx_geom geom;
    float *d_flow;
    int flow_row0;
    /* NULL: d_next is used as it is (top level, or already shifted with ofx_shift_1ch).  Non-NULL: 2 floats on the
     * device; the kernel reads d_next THROUGH the reference's global shift by that vector (cpu::shift_back_pyramid,
     * OptFlowCPU.cpp:241-282, fused into the row loads) -- same bits as shifting first, without the extra pass. */
    const float *d_uv;
    int accumulate; /* non-zero: d_flow += result (refinement iterations, ofx_warp_levels) instead of d_flow = result */
    /* Extension (the reference divides unguarded, OptFlowGpu.cu:1833-1838 / OptFlowCPU.cpp:369-372, and writes
     * NaN / Inf / huge vectors where a window has no texture or a single edge): > 0 = a pixel whose determinant Sxx*Syy - Sxy^2,
     * rounded to float, is below min_det gets the flow (0, 0).  0 = the reference.  One value per launch: the first
     * descriptor's. */
    float min_det;
    /* Extension (lk_iter; csrc/lk_body_warp.h).  d_warp_out non-NULL, with accumulate set: the launch also writes the warped image
     * the NEXT refinement iteration reads as its d_next -- d_warp_out = ofx_warp_levels(d_warp_src, the flow this launch leaves in
     * d_flow, warp_scale), same bytes -- so that only a pair's first refinement iteration needs ofx_warp_levels.  d_warp_src is the
     * warp source (the globally shifted next image), d_warp_out a plane of the level's geometry other than d_next and d_warp_src.
     * d_warp_src MUST BE FOLLOWED BY THREE READABLE BYTES (ABI v10: the launch fetches a tap's dword at the tap's own byte, also in
     * the last three columns of the plane's last row; the bytes beyond the plane are never looked at.  Every plane of an ofx_session
     * is followed by 64.)
     * For all descriptors of a launch or for none.  On a row window (a shard: geom.row0 / rows / out rows are not the whole level)
     * d_warp_out receives the out rows, a tap row outside [row0, row0 + rows) is replaced by the nearest row held, and bit
     * warp_status_bit of *d_warp_status (optional) is set when a pixel with a finite flow needed such a row. */
    const uint8_t *d_warp_src;
    uint8_t *d_warp_out;
    float warp_scale;
    int *d_warp_status;
    int warp_status_bit;
} ofx_lk_desc;
int ofx_lk_levels(const ofx_lk_desc *levels, int n, int window, int mode, void *stream);

/* descriptor of one level's shift (ofx_shift_levels, ofx_stream_launch) */
typedef struct ofx_shift_desc {
    const uint8_t *d_src;
    uint8_t *d_dst;
    ofx_geom geom;
    const float *d_uv;
} ofx_shift_desc;

/* One tick of the frame-stream pipeline in ONE launch: the pyramids of the newest frame(s) | the corner flows of earlier
 * pair(s) | the fused LK of still earlier pair(s), as disjoint block ranges of one grid.  Every stage only reads what
 * earlier launches wrote, so the stages need no synchronisation; a stage is skipped when its count is 0.  A tick may
 * carry up to OFX_STREAM_MAX_BATCH frames / pairs per stage (ofx_params.stream_batch): the LK items of B pairs then
 * share one launch, which multiplies the strip height by B (1/B of the priming rows per output row) and divides the
 * number of launches by B.
 * ofx_session_stream_submit drives this; it is exposed for callers that manage their own buffers. */
#ifndef OFX_STREAM_MAX_BATCH
#define OFX_STREAM_MAX_BATCH 16
#endif
typedef struct ofx_pyramid_stage {
    /* levels 1..levels-1 from d_frame, plus a copy of level 0 into d_levels[0] */
    const uint8_t *d_frame;
    int frame_pitch, w, h, levels;
    uint8_t *d_levels[OFX_MAX_LEVELS];
    int pitches[OFX_MAX_LEVELS];
    /* row windows of the destination planes (row-sharded callers): with windowed != 0, d_levels[k] holds the global rows
     * [row0[k], row0[k] + rows[k]) of level k and only those are written; d_frame is always the whole frame. */
    int windowed;
    int row0[OFX_MAX_LEVELS], rows[OFX_MAX_LEVELS];
    /* a second, small pyramid of the same frame's top-left patch_w x patch_h corner (patch_levels = 0: none).  A
     * pyramid of such a patch equals the top-left part of the frame's pyramid at every level (the stencil 2x-1..2x+1
     * never reaches past column/row 2*w_k-1), which lets a rank that does not hold row 0 compute the corner flows. */
    int patch_w, patch_h, patch_levels;
    uint8_t *d_patch_levels[OFX_MAX_LEVELS];
    int patch_pitches[OFX_MAX_LEVELS];
} ofx_pyramid_stage;
typedef struct ofx_corner_stage {
    /* descriptors as for ofx_corner_flows.  cols[k] > 0: the planes of level k are a patch holding columns [0, cols[k])
     * and rows [0, geom.rows) of the geom.w x geom.h level; d_status (may be NULL): bit k is OR-ed in when level k needed
     * a pixel inside the image but outside its planes (the shift left the patch: that pair's result is not the
     * reference's).  shard_rows[k] = {need0, need1, valid0, valid1} (all zero: unchecked), for a row-sharded caller: the
     * image rows [need0, need1) of level k that the LK stencils of this shard touch and the rows [valid0, valid1) its
     * buffers hold; bit 8 + k is OR-ed in when level k's vertical shift sends those reads to rows that exist in the image
     * but not in the shard (the rows of that pair next to the shard's edge are then not the reference's). */
    ofx_lk_desc level[OFX_MAX_LEVELS];
    int levels;
    float *d_uv;
    int cols[OFX_MAX_LEVELS];
    int *d_status;
    int shard_rows[OFX_MAX_LEVELS][4];
    /* build_patch != 0: the stage first builds the patch pyramids it reads -- levels 1 .. levels-1 of the top-left
     * patch_w x patch_h corner of BOTH frames, from d_patch_src[0] (the pair's previous frame) and d_patch_src[1] (its next
     * frame) into d_patch[0][k] / d_patch[1][k] (pitch patch_pitch[k]) -- and then walks the chain.  level[k] must
     * describe exactly those planes for k >= 1 and the frames themselves for k = 0.  With it a pair's shift vectors need
     * nothing but the two frames, i.e. they can be formed in the tick in which the pair's second frame arrives
     * (ofx_params.stream_two_stage). */
    int build_patch, patch_w, patch_h;
    const uint8_t *d_patch_src[2];
    int patch_src_pitch[2];
    uint8_t *d_patch[2][OFX_MAX_LEVELS];
    int patch_pitch[OFX_MAX_LEVELS];
    /* REPAIR of a shift that leaves the patch (cpu::shift_back_pyramid defines the shift for every input,
     * OptFlowCPU.cpp:255-273: a flat or nearly singular corner sends the next level's shifted corner anywhere in the image).
     * d_patch_reloc[1 .. levels-1] != NULL: a third set of patch planes (same sizes and pitches as d_patch[f]); when level k
     * needs a pixel of the next frame's pyramid that the top-left patch does not hold, the stage's block rebuilds that small
     * pyramid AROUND the shifted corner from the whole next frame and runs the level again on it -- no status bit, no host
     * round trip, the reference's result.  Needs level[0] to describe the WHOLE frames (geom.rows = h, cols[0] = 0 or w) and
     * patch_w / patch_h / patch_pitch to be set (with or without build_patch).  NULL: bit k of the status words is raised
     * instead, as before.
     * d_pair_status (may be NULL): THIS pair's status word is WRITTEN there when the chain ends: the bits raised in
     * *d_status by this pair, plus OFX_STATUS_REPAIRED when a relocated patch was used (the pair is exact all the same). */
    uint8_t *d_patch_reloc[OFX_MAX_LEVELS];
    int *d_pair_status;
} ofx_corner_stage;
#define OFX_STATUS_REPAIRED (1 << 24)
typedef struct ofx_stream_stages {
    ofx_pyramid_stage pyr[OFX_STREAM_MAX_BATCH];
    int n_pyr;
    ofx_corner_stage corner[OFX_STREAM_MAX_BATCH];
    int n_corner;
    ofx_lk_desc lk[OFX_MAX_LK_ITEMS]; /* the LK items of every pair of the tick (levels x pairs <= OFX_MAX_LK_ITEMS) */
    int n_lk;
    int deep_fetch; /* 0 = by size, +1 / -1 = deep / one-step row fetch of the LK stage (see ofx_params.deep_fetch; OFX_LK_DMA overrides) */
} ofx_stream_stages;
int ofx_stream_launch(const ofx_stream_stages *stages, int window, int mode, void *stream);
/* Measurement hook: with a device buffer of 8 * capacity_blocks uint64 set, every wave of every later ofx_stream_launch
 * records its start and end time (100 MHz wall clock) at [2 * (4 * block + wave)]; first (2 * OFX_STREAM_MAX_BATCH + 1 ints, may be NULL) receives
 * the block ranges of the last launch (corner blocks first, then LK up to first[0], then the pyramid stages).
 * d_buf = NULL switches it off.  Process-global, not thread-safe. */
int ofx_debug_stream_trace(unsigned long long *d_buf, int capacity_blocks, int *first);

/* Same level, but stopping before the solve: writes the five window sums
 * (Sxx, Syy, Sxy, Sxt, Syt) as int32 planes of w ints per row.  Test/inspection
 * entry point; equals five gpu::srm_1ch(_float) calls (OptFlowGpu.cu:1944-1960). */
int ofx_lk_level_sums(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode,
                      int32_t *d_sums5, int flow_row0, void *stream);

/* 2x decimating 3x3 Gaussian on a 1ch plane: dst(x,y) = trunc(sum G[p][q] *
 * src(2x-1+q, 2y-1+p)), taps outside the source skipped.  Replaces one level
 * of gpu::gauss_pyramid (OptFlowGpu.cu:1198-1232) / cpu::downscale_gaussian
 * (OptFlowCPU.cpp:112-148) for grey images.  `dst` describes the destination
 * level; the source level is (2*dst->w) x (2*dst->h) and src_row0/src_rows say
 * which of its rows d_src holds. */
int ofx_downsample_1ch(const uint8_t *d_src, int src_pitch, int src_row0, int src_rows,
                       uint8_t *d_dst, const ofx_geom *dst, void *stream);

/* All levels of a grey pyramid from level 0 in ONE launch (same arithmetic as ofx_downsample_1ch level by level;
 * whole, unsharded levels only).  d_levels[k] / pitches[k] for k = 1..levels-1 (index 0 unused). */
int ofx_pyramid_1ch(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches,
                    int levels, void *stream);

/* Translation the reference applies to `next` below the top level:
 * (u,v) = sum_{k=top..level+1} 2^(k-level) * flow_k[pixel 0]  in float, coarsest
 * first (OptFlowCPU.cpp:255-266, where `i * (1 >> offset)` is always 0).
 * d_flow_levels[k] = device pointer to level k's flow (only k > level are read).
 * Writes 2 floats to d_uv. */
int ofx_shift_vector(const float *const *d_flow_levels, int level, int max_level, float *d_uv, void *stream);

/* The shift of level k only needs PIXEL 0 of every coarser flow level (OptFlowCPU.cpp:260-262), and pixel 0's flow
 * only needs the (radius+2)^2 top-left corner of its level.  This entry point walks the pyramid coarse to fine in
 * one tiny launch: for each level it forms the shift vector from the corner flows found so far, evaluates pixel 0's
 * window sums on the shifted corner, solves, and stores d_uv[2k..2k+1] (and flow level k's pixel 0 when the
 * descriptor's flow_row0 is 0).  After it every level's shift and LK launch is independent of the others.
 * Descriptors: index k = pyramid level k, d_next = the UNSHIFTED next plane, geom.row0 must be 0. */
int ofx_corner_flows(const ofx_lk_desc *levels, int n_levels, int window, int mode, float *d_uv, void *stream);

/* Several ofx_shift_1ch calls in one launch (ofx_shift_desc is declared above). */
int ofx_shift_levels(const ofx_shift_desc *levels, int n, void *stream);

/* dst(x,y) = src((int)(x+u), (int)(y+v)) when that lands inside the image,
 * else src(x,y) if 3*(y*w+x) < w*h else 0 -- cpu::shift_back_pyramid
 * (OptFlowCPU.cpp:241-282) on channel 0 with the destination zero-initialised.
 * (u,v) is read from d_uv on the device.  Source rows needed but absent from
 * the buffer make the affected pixels undefined; the caller sizes halos. */
int ofx_shift_1ch(const uint8_t *d_src, uint8_t *d_dst, const ofx_geom *g, const float *d_uv, void *stream);

/* Extension (the reference has no iterations): d_dst(x,y) = round_u8(bilinear(d_src, x + scale*u, y + scale*v))
 * with (u,v) = d_flow at (x,y), replicate border, non-finite flow = no warp.  The planes may hold a row window of the level
 * (geom.row0 / rows; rows [out_y0, out_y1) are produced): a source row the warp needs that lies inside the image but outside
 * the window is replaced by the nearest row held and, when d_status is not NULL, bit status_bit is OR-ed into *d_status (a
 * row-sharded caller's flow reached beyond its halo).  One refinement
 * iteration of a level = this warp of the (shifted) next image by the flow so far, then ofx_lk_levels with
 * accumulate = 1.  scale = OFX_ITER_SCALE turns the reference's flow units into pixels (Sobel gain 8 / Dt_3x3 gain 15). */
#define OFX_ITER_SCALE 0.533333361148834228515625f
typedef struct ofx_warp_desc {
    const uint8_t *d_src;
    uint8_t *d_dst;
    ofx_geom geom;
    const float *d_flow;
    int flow_row0;
    float scale;
    int *d_status;  /* may be NULL */
    int status_bit;
} ofx_warp_desc;
int ofx_warp_levels(const ofx_warp_desc *levels, int n, void *stream);

/* main.cu:138-147: dense flow at `level` = sum_k 2^(k-level) flow_k(y>>s, x>>s). */
int ofx_compose_flow(const float *const *d_flow_levels, int w, int h, int levels, int level, float *d_dst,
                     void *stream);

/* ---- layout helpers ------------------------------------------------------ */
int ofx_extract_ch0(const uint8_t *d_src3, uint8_t *d_dst1, int w, int h, int dst_pitch, void *stream);
int ofx_replicate_3ch(const uint8_t *d_src1, int src_pitch, uint8_t *d_dst3, int w, int h, void *stream);

/* ---- API-compat primitives, device-resident -------------------------------
 * Generic (any mask / window size) counterparts of the reference kernels that
 * the flow path does not use in fused form. */
/* gpu::grayscale_avg, OptFlowGpu.cu:47-60 */
int ofx_grayscale_avg_3ch(const uint8_t *d_src3, uint8_t *d_dst3, int w, int h, void *stream);
/* gpu::conv_3ch_2d / _constant, OptFlowGpu.cu:108-147 (int accumulators, per-tap truncation) */
int ofx_conv_3ch(const uint8_t *d_src3, uint8_t *d_dst3, int w, int h, const float *h_mask, int mw, int mh,
                 int float_acc, void *stream);
/* gpu::conv_3ch_1ch_constant / _tiled, OptFlowGpu.cu:380-425 */
int ofx_conv_3ch_1ch_u8(const uint8_t *d_src3, int w, int h, uint8_t *d_dst, const float *h_mask, int mw, int mh,
                        void *stream);
/* gpu::conv_3ch_1ch_tiled_uchar_float, OptFlowGpu.cu:1040-1090 */
int ofx_conv_3ch_1ch_f32(const uint8_t *d_src3, int w, int h, float *d_dst, const float *h_mask, int mw, int mh,
                         void *stream);
/* gpu::gauss_pyramid one level on 3ch images, OptFlowGpu.cu:1198-1232 (mask fixed to GAUS_KERNEL_3x3) */
int ofx_downsample_3ch(const uint8_t *d_src3, uint8_t *d_dst3, int dw, int dh, void *stream);
/* gpu::srm_1ch, OptFlowGpu.cu:1463-1502 */
int ofx_srm_u8(const uint8_t *d_a, const uint8_t *d_b, int w, int h, int ww, int wh, int32_t *d_dst, void *stream);
/* gpu::srm_1ch_float, OptFlowGpu.cu:1549-1588 (row-major float accumulation) */
int ofx_srm_f32(const float *d_a, const float *d_b, int w, int h, int ww, int wh, float *d_dst, void *stream);
/* gpu::inverse_matrix, OptFlowGpu.cu:1727-1755 */
int ofx_solve_i32(const int32_t *d_sxx, const int32_t *d_syy, const int32_t *d_sxy, const int32_t *d_sxt,
                  const int32_t *d_syt, float *d_flow, int w, int h, int variant, void *stream);
/* gpu::inverse_matrix_float, OptFlowGpu.cu:1819-1846 */
int ofx_solve_f32(const float *d_sxx, const float *d_syy, const float *d_sxy, const float *d_sxt,
                  const float *d_syt, float *d_flow, int w, int h, void *stream);
/* solve variants for ofx_solve_i32 */
#define OFX_SOLVE_F64 0        /* OptFlowGpu.cu:1737-1754, double, correct */
#define OFX_SOLVE_INLINE_CPU 1 /* OptFlowCPU.cpp:369-382, double, c unscaled */
#define OFX_SOLVE_F32 2        /* OptFlowCPU.cpp:293-304, float */
/* utils::generate_gaussian_kernel, OptFlowUtils.cpp:68-114 (host arithmetic, double; dst holds ks*ks values,
 * (ks+1)^2 when ks is even, 2*pi*sigma rounded when ks == -1) */
void ofx_generate_gaussian_kernel(double sigma_s, int kernel_size, double *h_dst);
/* gpu::bilinear_filter (a bilateral filter), OptFlowGpu.cu:1984-2048 */
int ofx_bilateral_3ch(const uint8_t *d_src3, const uint8_t *d_gray3, uint8_t *d_dst3, int w, int h, int ww, int wh,
                      double sigma_s, double sigma_b, void *stream);

/* The same filter within SURVEY 8c's tolerance for this stage (+-1 LSB of the reference's byte), several times faster: float
 * accumulators, the range weight evaluated (v_exp_f32) instead of looked up, the quotient formed around the centre value for
 * grey images.  Odd ww <= 13, odd wh <= ww.  Opt-in: ofx_bilateral_3ch stays the bit-exact kernel. */
int ofx_bilateral_3ch_fast(const uint8_t *d_src3, const uint8_t *d_gray3, uint8_t *d_dst3, int w, int h, int ww, int wh,
                           double sigma_s, double sigma_b, void *stream);
/* gpu::bilinear_filter / cpu::bilinear_filter_3ch have the reference's signatures and carry no mode: this process-wide switch
 * makes them run ofx_bilateral_3ch_fast (on != 0) or the bit-exact kernel (on == 0, the default; environment
 * OFX_BILATERAL_FAST=1 starts with it on).  on < 0 only queries.  Returns the previous setting. */
int ofx_bilateral_wrappers_fast(int on);

/* the remaining functions of namespace cpu (OptFlowCpu.hpp:3-184), device-resident, so that the cpu:: call surface of
 * include/OptFlowCpu.hpp runs on the MI355X as well */
/* cpu::sub_arr, OptFlowCPU.cpp:11-17 (bytes, wrapping) */
int ofx_sub_u8(const uint8_t *d_a, const uint8_t *d_b, size_t n, uint8_t *d_dst, void *stream);
/* cpu::srm_3ch, OptFlowCPU.cpp:202-238 (bounds test `>` as there; taps past the end of the buffer contribute nothing) */
int ofx_srm_3ch_u8(const uint8_t *d_a3, const uint8_t *d_b3, int w, int h, int ww, int wh, int32_t *d_dst3, void *stream);
/* cpu::downscale_gaussian, OptFlowCPU.cpp:112-148: one pyramid level on 3ch images with the CALLER's mask (<= 81 taps) */
int ofx_downscale_mask_3ch(const uint8_t *d_src3, uint8_t *d_dst3, int dw, int dh, const float *h_mask, int mw, int mh, void *stream);
/* cpu::shift_back_pyramid on the 3ch image, OptFlowCPU.cpp:241-282: d_dst3 keeps its contents except for its first w*h bytes
 * (copied from d_src3) and the pixels whose shifted target lies inside the image */
int ofx_shift_3ch(const uint8_t *d_src3, uint8_t *d_dst3, int w, int h, const float *d_uv, void *stream);

/* ---- session: device-resident pyramids for a stream of frames -------------
 * Mirrors main.cu:192-272: the previous frame's pyramid is kept, each new
 * frame gets its pyramid built and every level is solved coarse to fine. */
typedef struct ofx_session ofx_session;

typedef struct ofx_params {
    int width, height; /* level-0 size; (width>>k, height>>k) must be even for k < levels-1 */
    int levels;        /* 1..OFX_MAX_LEVELS (main.cu:192 uses 4) */
    int window;        /* reference: 9 (CPU, OptFlowCPU.cpp:344) / 19 (GPU, OptFlowGpu.cu:1944) */
    int mode;          /* OFX_MODE_* */
    int device;        /* HIP device ordinal */
    /* Row sharding.  sharded == 0: this session holds whole levels.  sharded != 0: for level k this
     * rank OWNS global rows [own_y0[k], own_y1[k]) -- it computes the pyramid and the flow for them -- and its
     * plane buffers HOLD rows [buf_y0[k], buf_y1[k]) (own rows plus halo; the halo rows are filled by the
     * caller's exchange between ofx_session_downsample_level and ofx_session_run_levels, or recomputed: comp_y*). */
    int sharded;
    int own_y0[OFX_MAX_LEVELS], own_y1[OFX_MAX_LEVELS];
    int buf_y0[OFX_MAX_LEVELS], buf_y1[OFX_MAX_LEVELS];
    /* rows of level k (k >= 1) this rank produces itself when it builds the pyramid, own <= comp <= buf.
     * comp == own: halos come from the neighbours (exchange); comp == buf: halos are recomputed locally from a
     * wider halo one level below (no exchange).  Ignored (treated as own) when comp_y1[k] == 0. */
    int comp_y0[OFX_MAX_LEVELS], comp_y1[OFX_MAX_LEVELS];
    /* refinement iterations per level (extension, lk_float only): 0 or 1 = the reference (no refinement).  Unsharded
     * sessions run them pair at a time (ofx_session_run_flow) or through the stream pipeline (one warp + one accumulating LK
     * launch per iteration for all pairs of a tick); sharded sessions through the stream pipeline only (local_corner), with
     * buf_y* holding own rows +- ((window/2 + 1) * iters + slack for the shift and the warp): parallel.ShardPlan(iters=...). */
    int iters;
    /* Sharded sessions: compute the corner flows (the reference's shift vectors, formed from pixel 0 of every coarser
     * flow level) LOCALLY from a small top-left patch of every frame instead of receiving them from the rank that owns
     * row 0.  Every frame handed to the session is the whole frame, so the patch is always at hand; with it a rank
     * needs nothing from any other rank and can run the one-launch-per-frame stream pipeline
     * (ofx_session_stream_*).  patch_size: level-0 side of the square patch, 0 = automatic
     * (2^(levels-1) * (window/2 + 2 + 8), at least 256, clipped to the frame).  With borrow_frames the shift is exact for
     * every input (a shifted corner that leaves the patch is read through a relocated one, ofx_corner_stage.d_patch_reloc);
     * with copied frames it is exact while it stays inside the patch and ofx_session_corner_status reports when it did not. */
    int local_corner;
    int patch_size;
    /* frames per tick of the stream pipeline: 0 or 1 = one launch per frame, 2 .. 16 = one launch per that many frames
     * (see ofx_session_stream_submit); stream_batch * levels <= OFX_MAX_LK_ITEMS. */
    int stream_batch;
    /* Stream pipeline without its own copy of level 0: the LK and corner stages read level 0 straight from the frame
     * buffers handed to ofx_session_stream_submit, and the pyramid stage only writes levels 1 and up.
     * LIFETIME RULE (the one statement of it; INTEGRATION.md, DESIGN.md and engine.py quote it): the buffer of frame f
     * (frames counted from 0, B = stream_batch) is last read by the launch that the submit of frame f + dB enqueues, at the
     * latest, where d = 3, or 2 with stream_two_stage (below).  It may be rewritten (a) by work enqueued on the SAME stream after that submit call, or (b) from the host or
     * another stream once that launch has COMPLETED -- the submit call returning is not enough.  A producer that writes
     * frame g into its buffer on the stream, right before submitting it, therefore needs a ring of at least dB + 1
     * buffers; one that writes asynchronously needs as many more as it has launches in flight.
     * Saves 2 bytes per level-0 pixel of HBM traffic per frame and a third of the pipeline's cache working set
     * (DESIGN.md section 4.3).  0 = copy (any buffer lifetime).
     * Pair-at-a-time path (ofx_session_set_frame_device + build_pyramid + run_flow + swap): the flag makes
     * ofx_session_set_frame_device remember the buffer instead of copying it (no copy launch; the pyramid, corner and LK
     * launches read it in place).  There the buffer of a frame is read until the run_flow of the pair in which it is the
     * PREVIOUS frame has run, and its pitch must be the session's level-0 pitch (the width rounded up to 64 bytes,
     * ofx_session_plane reports it) at a 4-byte aligned address; host frames (ofx_session_set_frame_host*) and the staged
     * path still copy.  A single-level session with refinement iterations (levels == 1, iters > 1) copies its frames whatever
     * this flag says (pair at a time: the stream pipeline needs two levels): its level 0 would be the fused warp's source, which must be followed by three
     * readable bytes (ofx_lk_desc.d_warp_src). */
    int borrow_frames;
    /* determinant guard of the solve, see ofx_lk_desc.min_det (0 = the reference: flat regions are NaN) */
    float min_det;
    /* Stream pipeline in TWO stages instead of three: a tick runs pyramid(its frames) | corner(the pairs its frames complete)
     * | LK(the pairs of the tick before), the corner stage building the two small patch pyramids it needs itself
     * (ofx_corner_stage.build_patch) instead of waiting a tick for the pyramid stage.  A pair's flow is complete one tick
     * earlier, the pipeline holds 2B + 2 image sets instead of 3B + 2 and a borrowed frame f is read until the launch
     * enqueued by the submit of frame f + 2B (ring of >= 2B + 1 buffers) -- which is what lets eight 4K frames per launch
     * stay inside the Infinity Cache.  Needs borrow_frames.  The shift vectors are exact for EVERY input: a shifted corner that
     * leaves the patch (256 level-0 pixels or patch_size) is read through a patch pyramid the corner block rebuilds around it
     * (ofx_corner_stage.d_patch_reloc; ofx_session_pair_status says for which pairs that happened). */
    int stream_two_stage;
    /* The frame buffers handed to this (sharded, local_corner) session are PARTIAL: only the level-0 rows the plan holds
     * (buf_y0[0] .. buf_y1[0]) and the frame's top-left patch were ever written; every other byte is undefined (the ranks of
     * parallel.ShardedFlow's "stream_exchange" mode only ever receive those).  The corner chain then reads level 0 through the
     * patch's extent only and the repair of a shift that leaves the patch is OFF (it would rebuild from rows that are not
     * there): such a pair raises bit k of the status word, as with copied frames, and is an error.  Not with stream_two_stage. */
    int frames_partial;
    /* Where the stream tick's LK stage expects its image rows to come from (ABI v10).  0 = decide by the size of the largest level
     * (levels of 16 Mpx and more: deep fetch); +1 = the frames handed over are COLD -- they were last touched more than an
     * Infinity Cache (256 MiB) of traffic ago, e.g. a long pool of decoded surfaces that is consumed much later than it is
     * written: fetch rows two steps ahead straight into LDS (ofx_stream_stages.deep_fetch; measured on MI355X with a ring of 44
     * 4K frames: 272.8 vs 287.2 us per tick of eight pairs, 1080p 137.8 vs 142.8); -1 = they are warm (just written by a producer, or a
     * short ring): fetch one step ahead (2 % faster then).  A hint about speed only: every choice gives the same bits. */
    int deep_fetch;
} ofx_params;

int ofx_session_create(const ofx_params *p, ofx_session **out);
/* Sharded sessions: *h_status receives (and the session clears) the OR over all pairs so far of bit k = "level k's shift left
 * the top-left patch" (local_corner), bit 8 + k = "level k's vertical shift reached image rows beyond this shard's halo" (the
 * margin rows of the plan) and bit 16 + k = "a refinement iteration's warp at level k reached beyond the halo" (iters > 1);
 * 0 = every pair so far is exactly the unsharded result.  Synchronises `stream`. */
int ofx_session_corner_status(ofx_session *s, int *h_status, void *stream);
/* The same word for ONE pair of the stream pipeline (frames counted from 0, pair p = frame p-1 -> frame p), while it is one of
 * the newest 2 * stream_batch pairs whose corner stage has run: the bits that pair raised, plus OFX_STATUS_REPAIRED when its
 * shifted corner left the top-left patch and was read through a relocated one (informational: the flow is the reference's).
 * With the repair in place (stream_two_stage, or local_corner with borrow_frames and whole frames: not frames_partial) bit k can
 * no longer occur; bits 8 + k and
 * 16 + k (a shard's halo) stay errors.  Synchronises `stream`; does not clear anything. */
int ofx_session_pair_status(ofx_session *s, int pair, int *h_status, void *stream);
int ofx_session_destroy(ofx_session *s);
/* Load the NEXT frame's level 0 (1ch, tightly packed w bytes per row, full frame) from host / device memory. */
int ofx_session_set_frame_host(ofx_session *s, const uint8_t *h_gray1, void *stream);
int ofx_session_set_frame_host_3ch(ofx_session *s, const uint8_t *h_img3, void *stream);
int ofx_session_set_frame_device(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream);
/* Build the next frame's pyramid (gpu::gauss_pyramid, main.cu:250): ofx_session_downsample_level for k = 1..levels-1. */
int ofx_session_build_pyramid(ofx_session *s, void *stream);
/* Level k of the next frame's pyramid from level k-1, own rows only. */
int ofx_session_downsample_level(ofx_session *s, int level, void *stream);
/* All levels against the previous frame's pyramid (main.cu:256-262) in three launches: ofx_session_corner_flows,
 * then ofx_session_run_levels (one multi-level shift launch, one multi-level LK launch). */
int ofx_session_run_flow(ofx_session *s, void *stream);
int ofx_session_corner_flows(ofx_session *s, void *stream);
int ofx_session_run_levels(ofx_session *s, void *stream);
/* The same result the reference's way: per level ofx_session_compute_uv then ofx_session_run_level, coarse to fine. */
int ofx_session_run_flow_sequential(ofx_session *s, void *stream);
/* Shift vector of `level` from the coarser flows' pixel 0 into the session's uv slot (meaningful on the rank that
 * owns row 0; a sharded driver broadcasts the slot before ofx_session_run_level). */
int ofx_session_compute_uv(ofx_session *s, int level, void *stream);
/* Shift (below the top level) + fused LK of one level, own rows, using the uv slot as it stands. */
int ofx_session_run_level(ofx_session *s, int level, void *stream);
/* Pipelined pair (throughput path): the staging half of a pair -- frame load, pyramid, corner flows, shifts -- runs on
 * a session-owned auxiliary stream underneath the previous pair's LK launch; the solve half runs on `stream`.
 * ofx_session_submit_device does a whole pair and leaves the new frame as the previous one; results are bit-identical
 * to set_frame_device + build_pyramid + run_flow + swap.  d_gray1 must be complete in HBM at the call.
 * A sharded driver issues the halves itself: stage_frame; corner_flows (rank holding row 0) and the broadcast of the
 * shift vectors on the aux stream (ofx_session_aux_stream); stage_shift; solve_staged. */
int ofx_session_submit_device(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream);
int ofx_session_stage_frame(ofx_session *s, const uint8_t *d_gray1, int pitch, void *aux_stream);
int ofx_session_stage_shift(ofx_session *s, void *aux_stream);
int ofx_session_solve_staged(ofx_session *s, void *stream);
int ofx_session_aux_stream(ofx_session *s, void **stream);
/* Stream pipeline (highest throughput): ONE launch (ofx_stream_launch) per tick of B = ofx_params.stream_batch frames
 * (1 .. 16), in which the pyramids of those frames, the corner flows of the B pairs before and the fused LK (shift
 * included) of the B pairs before that run side by side.  With B = 1 every call launches and the flow of pair p (frame
 * p-1 -> frame p, frames counted from 0) is written by the launch of frame p+2.  (ofx_params.stream_two_stage: the corner
 * flows of the pairs the tick's own frames complete, the LK of the B pairs before -- everything one tick earlier.)  With B > 1 only every B-th call launches,
 * for the B frames received since the last launch (the others are remembered: their buffers must stay unmodified until
 * that launching call has returned); pairs complete B at a time, one tick later.  *completed_pair receives the HIGHEST
 * pair complete after the call in `stream` order (all lower ones are complete too; -1 when the call completed none); the
 * flows of the newest B pairs are at ofx_session_flow_of.  After the last frame call ofx_session_stream_drain until it
 * reports -2.  Results are bit-identical to the pair-at-a-time paths.  A frame buffer must stay valid until the launch
 * that received it has finished. */
int ofx_session_stream_begin(ofx_session *s);
int ofx_session_stream_submit(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream, int *completed_pair);
int ofx_session_stream_drain(ofx_session *s, void *stream, int *completed_pair);
/* The same for n consecutive frames in one call (n >= 1; pitches may be NULL when every frame's pitch is `pitch0`): exactly
 * what n calls of ofx_session_stream_submit do, for callers whose frames arrive in groups or whose call overhead counts
 * (an FFI call per frame is ~1.5 us; a rank of an 8-way sharded 4K pair spends ~5 us of GPU time per frame). */
int ofx_session_stream_submit_frames(ofx_session *s, const uint8_t *const *d_gray1, const int *pitches, int pitch0, int n, void *stream,
                                     int *completed_pair);
/* Flow of `pair` at `level` while it is one of the newest stream_batch completed pairs of the stream pipeline (pair p
 * lives in flow set p mod stream_batch).  Same outputs as ofx_session_flow. */
int ofx_session_flow_of(ofx_session *s, int pair, int level, float **d_ptr, int *row0, int *rows);
/* prev <- next (main.cu:270-272). */
int ofx_session_swap(ofx_session *s);
/* Device pointers / geometry of the session's buffers. which: 0 = prev, 1 = next, 2 = shifted scratch. */
int ofx_session_plane(ofx_session *s, int which, int level, uint8_t **d_ptr, ofx_geom *geom);
int ofx_session_flow(ofx_session *s, int level, float **d_ptr, int *row0, int *rows);
/* Shift vector slot of `level` for the pair IN PROGRESS (2 floats; levels are contiguous, 2 floats apart).  The session
 * alternates between two slots per pair, so query it again after every ofx_session_swap / solve_staged. */
int ofx_session_shift_uv(ofx_session *s, int level, float **d_uv);
/* Copy one level's flow (the rows this session owns, tightly packed) to host, synchronising `stream`. */
int ofx_session_get_flow_host(ofx_session *s, int level, float *h_dst, void *stream);

/* Time the fused LK launch (all levels in ofx_session_run_levels, level 0 in ofx_session_run_level) with HIP events recorded on the launch stream: arm for up to max_launches
 * launches (0 disarms); read returns the average/minimum duration in microseconds and re-arms. */
int ofx_session_timing(ofx_session *s, int max_launches);
int ofx_session_timing_read(ofx_session *s, double *avg_us, double *min_us, int *launches);
/* While armed, EVERY launch the session issues is bracketed and tagged with its kind; ofx_session_timing_read averages the
 * dominant ones (LK, LK_ACC, STREAM) and re-arms, ofx_session_timing_read_kind reads one kind and leaves the records in place
 * (call it before ofx_session_timing_read). */
#define OFX_TIME_LK 0      /* fused LK launch that writes the flow (all levels, or level 0 of run_level) */
#define OFX_TIME_LK_ACC 1  /* fused LK launch of a refinement iteration (flow += result) */
#define OFX_TIME_WARP 2    /* bilinear warp of a refinement iteration, all levels */
#define OFX_TIME_STREAM 3  /* one tick of the stream pipeline */
#define OFX_TIME_SHIFT 4   /* stand-alone global shift of all levels (refinement iterations only) */
#define OFX_TIME_CORNER 5  /* corner kernel */
#define OFX_TIME_PYRAMID 6 /* fused pyramid launch */
#define OFX_TIME_LK_ACC_WARP 7 /* the same launch when it also writes the next iteration's warped image (lk_body_warp.h) */
#define OFX_TIME_KINDS 8
int ofx_session_timing_read_kind(ofx_session *s, int kind, double *avg_us, double *min_us, int *launches);

/* ---- host-pointer convenience used by the gpu:: compat surface ------------ */
/* gpu::calc_opt_flow (OptFlowGpu.cuh:33): host 3ch images in, host flow pyramid in/out. */
int ofx_calc_opt_flow_host(const uint8_t *h_prev3, const uint8_t *h_next3, int w, int h, float **h_flow_pyr,
                           int level, int max_level, int window, int mode);

/* The host-pointer entry points (this one, ofx_compose_flow_host and the gpu:: / cpu:: wrappers) move transfers of 3 MB and
 * more in 2 MB chunks through pinned bounce buffers, on several threads at once: the caller and a small pool the library keeps
 * (environment OFX_STAGE_THREADS = pool size, default 3; -1 = plain hipMemcpy).  Returns the number of threads that share a
 * transfer.  The calls stay synchronous and retain nothing of the caller's. */
int ofx_stage_threads(void);

/* main.cu:138-147 (the dense field visualizeFlowField samples) with host pointers: h_flow_pyr[k] for k >= level are the
 * host flow levels gpu::calc_opt_flow filled, (w, h) is the size of `level`; h_dst receives 2*w*h floats. */
int ofx_compose_flow_host(float *const *h_flow_pyr, int w, int h, int levels, int level, float *h_dst);

#ifdef __cplusplus
}
#endif
#endif /* OFX_H */
