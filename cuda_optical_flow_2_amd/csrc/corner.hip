// Corner kernel: the flow of pixel 0 at every pyramid level, coarse to fine, in one single-wave launch.
//
// The reference shifts `next` at level k by (u,v) = sum_{j>k} 2^(j-k) * flow_j[pixel 0] (OptFlowCPU.cpp:255-266:
// `i * (1 >> offset)` is 0 for every offset >= 1, so only element 0 of each coarser flow level is ever read).  Hence
// the only dependency between the levels of a frame pair is a chain through PIXEL 0, whose flow depends on the
// (radius+1)^2 clipped window at the top-left corner of its level.  This kernel walks that chain once -- per level:
// shift vector, derivatives of the shifted corner, five window sums, the 2x2 solve (lk_solve.h, the same code the
// fused level kernel uses, so the values are bit-identical) -- and publishes every level's shift vector.  Afterwards
// all levels can be shifted and solved concurrently (ofx_shift_levels, ofx_lk_levels).
#include "corner_body.h"

using namespace ofx_dev;

namespace {

template <int MODE, bool FAST>
__global__ __launch_bounds__(64) void corner_kernel(const CornerArgs A)
{
    __shared__ float f0[2 * OFX_MAX_LEVELS];
    __shared__ __attribute__((aligned(16))) uint8_t cache[kCornerTileBytes + OFX_MAX_LEVELS * kCornerCacheBytes];
    corner_wave<MODE, FAST>(A.hd, A.lv, (int)threadIdx.x, f0, cache);
}

// Row-sharded sessions that RECEIVE their shift vectors (rank 0's corner kernel + broadcast) check them here: the level
// kernel reads next at row (int)(y + v) for the rows y its stencils touch, and a target inside the image must be a row the
// shard's buffers hold.  Same rule and same status bit (8 + k) as the corner wave applies for local_corner sessions.
struct MarginArgs {
    const float *uv;
    int *status;
    int levels;
    int h[OFX_MAX_LEVELS];
    int rows[OFX_MAX_LEVELS][4]; // need0, need1, valid0, valid1
};

__global__ __launch_bounds__(64) void shard_margin_kernel(const MarginArgs A)
{
    const int k = (int)threadIdx.x;
    if (k >= A.levels - 1) return; // the top level is not shifted
    const float v = A.uv[2 * k + 1];
    const int need0 = A.rows[k][0], need1 = A.rows[k][1], valid0 = A.rows[k][2], valid1 = A.rows[k][3];
    if (!(need1 > need0) || v != v) return;
    const float t0 = (float)need0 + v, t1 = (float)(need1 - 1) + v;
    if (t1 > -1.0f && t0 < (float)A.h[k]) {
        const int lo = max(0, (int)floorf(t0)), hi = min(A.h[k] - 1, (int)floorf(t1));
        if (lo <= hi && (lo < valid0 || hi >= valid1)) atomicOr(A.status, 1 << (8 + k));
    }
}

} // namespace

int ofx_shard_margin_check(const float *d_uv, int levels, const int *heights, const int *shard_rows, int *d_status, void *stream)
{
    OFX_REQUIRE(d_uv && heights && shard_rows && d_status && levels >= 1 && levels <= OFX_MAX_LEVELS, "ofx_shard_margin_check: bad arguments");
    MarginArgs a{};
    a.uv = d_uv;
    a.status = d_status;
    a.levels = levels;
    for (int k = 0; k < levels; ++k) {
        a.h[k] = heights[k];
        for (int j = 0; j < 4; ++j) a.rows[k][j] = shard_rows[4 * k + j];
    }
    hipLaunchKernelGGL(shard_margin_kernel, dim3(1), dim3(64), 0, ofx_stream(stream), a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

int ofx_corner_args(const ofx_lk_desc *levels, int n_levels, int window, int mode, float *d_uv, const int *cols, int *d_status,
                    const int *shard_rows, CornerHead *out, CornerLevel *lv_out)
{
    OFX_REQUIRE(levels && d_uv && n_levels >= 1 && n_levels <= OFX_MAX_LEVELS, "ofx_corner_flows: bad arguments");
    OFX_REQUIRE(window >= 3 && (window & 1), "ofx_corner_flows: window must be odd and >= 3 (got %d)", window);
    OFX_REQUIRE((window >> 1) <= kCornerMaxRadius, "ofx_corner_flows: window %d not supported (at most %d)", window, 2 * kCornerMaxRadius + 1);
    OFX_REQUIRE(mode == OFX_MODE_COMPAT_CPU || mode == OFX_MODE_LK_FLOAT || mode == OFX_MODE_LK_FLOAT_FAST, "ofx_corner_flows: bad mode %d", mode);
    CornerHead a{};
    a.levels = n_levels;
    a.min_det = levels[0].min_det;
    a.radius = window >> 1;
    a.uv = d_uv;
    a.status = d_status;
    for (int k = 0; k < n_levels; ++k) {
        const ofx_geom *g = &levels[k].geom;
        ofx_geom gc = *g; // a patch's planes are cols[k] wide: that is what the pitch has to cover
        if (cols && cols[k] > 0 && cols[k] < g->w) gc.w = cols[k];
        OFX_TRY(ofx_check_geom(&gc, "ofx_corner_flows"));
        OFX_REQUIRE(levels[k].d_prev && levels[k].d_next, "ofx_corner_flows: null plane at level %d", k);
        OFX_REQUIRE(g->row0 == 0, "ofx_corner_flows: level %d buffer must start at row 0 (it holds rows from %d)", k, g->row0);
        const int need = a.radius + 2 < g->h ? a.radius + 2 : g->h;
        OFX_REQUIRE(g->rows >= need, "ofx_corner_flows: level %d holds %d rows, the corner needs %d", k, g->rows, need);
        const int col_end = cols && cols[k] > 0 ? cols[k] : g->w;
        OFX_REQUIRE(col_end <= g->pitch && col_end >= (a.radius + 2 < g->w ? a.radius + 2 : g->w),
                    "ofx_corner_flows: level %d holds %d columns, the corner needs %d", k, col_end, a.radius + 2);
        lv_out[k] = CornerLevel{levels[k].d_prev, levels[k].d_next, levels[k].d_flow, g->w, g->h, g->pitch, g->rows, levels[k].flow_row0, col_end,
                                shard_rows ? shard_rows[4 * k] : 0, shard_rows ? shard_rows[4 * k + 1] : 0, shard_rows ? shard_rows[4 * k + 2] : 0,
                                shard_rows ? shard_rows[4 * k + 3] : 0};
    }
    *out = a;
    return OFX_OK;
}

extern "C" int ofx_corner_flows(const ofx_lk_desc *levels, int n_levels, int window, int mode, float *d_uv, void *stream)
{
    CornerArgs a{};
    OFX_TRY(ofx_corner_args(levels, n_levels, window, mode, d_uv, nullptr, nullptr, nullptr, &a.hd, a.lv));
    if (mode == OFX_MODE_LK_FLOAT)
        hipLaunchKernelGGL((corner_kernel<OFX_MODE_LK_FLOAT, false>), dim3(1), dim3(64), 0, ofx_stream(stream), a);
    else if (mode == OFX_MODE_LK_FLOAT_FAST)
        hipLaunchKernelGGL((corner_kernel<OFX_MODE_LK_FLOAT, true>), dim3(1), dim3(64), 0, ofx_stream(stream), a);
    else
        hipLaunchKernelGGL((corner_kernel<OFX_MODE_COMPAT_CPU, false>), dim3(1), dim3(64), 0, ofx_stream(stream), a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
