// Internal helpers shared by the HIP translation units of libofx_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ofx.h"

// thread-local last-error message behind ofx_last_error()
void ofx_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define OFX_HIP(call)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            ofx_set_error("%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);           \
            return OFX_E_HIP;                                                                            \
        }                                                                                                \
    } while (0)

#define OFX_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ofx_set_error(__VA_ARGS__); \
            return OFX_E_INVALID;       \
        }                               \
    } while (0)

#define OFX_TRY(expr)               \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != OFX_OK) return rc_; \
    } while (0)

static inline hipStream_t ofx_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int ofx_div_up(int a, int b) { return (a + b - 1) / b; }

// roctx range around a stage of the session (ofx_core.cpp; active with OFX_ROCTX=1, see there)
void ofx_range_push(const char *name);
void ofx_range_pop(void);
struct OfxRange {
    explicit OfxRange(const char *name) { ofx_range_push(name); }
    ~OfxRange() { ofx_range_pop(); }
    OfxRange(const OfxRange &) = delete;
    OfxRange &operator=(const OfxRange &) = delete;
};

// validates the parts of an ofx_geom every kernel relies on
int ofx_check_geom(const ofx_geom *g, const char *who);

// rows of the level that a stencil of vertical radius `halo` around [out_y0,out_y1) touches, clipped to the image,
// must be present in the buffer
int ofx_check_halo(const ofx_geom *g, int halo, const char *who);

// argument builders shared between the stand-alone stage launches and the stream (pipelined) launch; the structs live
// in stages_body.h / corner_body.h
namespace ofx_dev {
struct PyrArgs;
struct PyrMarchArgs;
struct ShiftTable;
struct CornerHead;
struct CornerLevel;
} // namespace ofx_dev
// row0/rows (NULL: whole levels): the global rows each destination plane (index 0 = the level-0 copy) holds
int ofx_pyramid_args(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches, int levels,
                     uint8_t *d_level0_copy, int copy_pitch, const int *row0, const int *rows, ofx_dev::PyrArgs *out,
                     size_t *lds_bytes, int *blocks_x, int *blocks_y);
// the stream kernel's pyramid stage (pyr_march.h); *items = waves needed
int ofx_pyramid_march_args(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches, int levels,
                           uint8_t *d_level0_copy, int copy_pitch, const int *row0, const int *rows, int target_waves,
                           ofx_dev::PyrMarchArgs *out, int *items);
int ofx_shift_table(const ofx_shift_desc *levels, int n, ofx_dev::ShiftTable *out, int *blocks_out);
// cols (NULL: full width): columns [0, cols[k]) each level's planes hold; d_status (NULL: none): see CornerHead::status;
// shard_rows (NULL: unchecked): 4 ints per level, CornerLevel::need0 .. valid1
int ofx_corner_args(const ofx_lk_desc *levels, int n_levels, int window, int mode, float *d_uv, const int *cols, int *d_status,
                    const int *shard_rows, ofx_dev::CornerHead *out, ofx_dev::CornerLevel *lv_out);
// the pair-at-a-time path's pyramid launch with the pair's corner chain aboard (pyr_corner.hip); C->build_patch must be set
int ofx_pyramid_corner_1ch(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches, int levels,
                           const ofx_corner_stage *C, int first, int window, int mode, void *stream);
// sharded sessions whose shift vectors come from another rank: raise status bit 8 + k when level k's vertical shift sends the
// shard's reads (rows [need0, need1) before the shift) to image rows outside [valid0, valid1); shard_rows = 4 ints per level
int ofx_shard_margin_check(const float *d_uv, int levels, const int *heights, const int *shard_rows, int *d_status, void *stream);
