// Pyramid downsample, global shift, flow composition and layout helpers for gfx950.
// All of these are pure streaming kernels (HBM-bound); they use 4-pixels-per-lane dword accesses where the layout
// allows it and one wave-row mapping so that every load/store instruction touches one contiguous span.
#include "pyr_march.h"
#include "stages_body.h"

using namespace ofx_dev;

namespace {

__global__ __launch_bounds__(256) void downsample_1ch_kernel(const DownArgs A)
{
    downsample_rows(A, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)blockIdx.y);
}

__global__ __launch_bounds__(kPyrThreads) void pyramid_fused_kernel(const PyrArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    pyramid_block(A, (int)blockIdx.x, (int)blockIdx.y + A.by0, (int)threadIdx.x, lds);
}

__global__ __launch_bounds__(256) void shift_1ch_kernel(const ShiftTable T)
{
    shift_block(T, (int)blockIdx.x, (int)threadIdx.x);
}

__global__ __launch_bounds__(256) void warp_u8_kernel(const WarpTable T)
{
    warp_block(T, (int)blockIdx.x, (int)threadIdx.x);
}

// 3-channel variant, one thread per destination pixel (API-compat path only: gpu::gauss_pyramid on colour images)
__global__ __launch_bounds__(256) void downsample_3ch_kernel(const uint8_t *src, uint8_t *dst, int dw, int dh)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const int sw = 2 * dw;
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int sy = 2 * y - 1 + p;
        if (sy < 0) continue;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int sx = 2 * x - 1 + q;
            if (sx < 0) continue;
            const int wgt = ((p == 1) ? 2 : 1) * ((q == 1) ? 2 : 1);
            const uint8_t *s = src + 3 * ((size_t)sy * sw + sx);
            acc[0] += wgt * s[0];
            acc[1] += wgt * s[1];
            acc[2] += wgt * s[2];
        }
    }
    uint8_t *d = dst + 3 * ((size_t)y * dw + x);
    d[0] = (uint8_t)(acc[0] >> 4);
    d[1] = (uint8_t)(acc[1] >> 4);
    d[2] = (uint8_t)(acc[2] >> 4);
}

// ---------------------------------------------------------------------------------------------------------------
// Global shift vector: OptFlowCPU.cpp:255-266.  `i * (1 >> offset)` is 0 for offset >= 1, so every pixel reads
// flow element 0 of each coarser level; float accumulation, coarsest level first.
struct FlowPtrs {
    const float *lv[OFX_MAX_LEVELS];
};

__global__ void shift_vector_kernel(const FlowPtrs P, int level, int max_level, float *uv)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float u = 0.0f, v = 0.0f;
    for (int k = max_level - 1; k > level; --k) {
        const float mult = (float)(1 << (k - level));
        u += mult * P.lv[k][0];
        v += mult * P.lv[k][1];
    }
    uv[0] = u;
    uv[1] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// main.cu:138-147 dense: u = sum_{k=levels-1..level} 2^(k-level) * flow_k(i>>s, j>>s); float u updated through a
// double product (u += (double)multiplier * f)
struct ComposeArgs {
    FlowPtrs P;
    float *dst;
    int w, h, levels, level;
};

__global__ __launch_bounds__(256) void compose_flow_kernel(const ComposeArgs A)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= A.w || i >= A.h) return;
    float u = 0.0f, v = 0.0f;
    for (int k = A.levels - 1; k >= A.level; --k) {
        const int sc = k - A.level;
        const size_t pos = (size_t)(i >> sc) * (size_t)(A.w >> sc) + (size_t)(j >> sc);
        const double m = (double)(1 << sc);
        const float2 f = reinterpret_cast<const float2 *>(A.P.lv[k])[pos];
        u = (float)((double)u + m * (double)f.x);
        v = (float)((double)v + m * (double)f.y);
    }
    reinterpret_cast<float2 *>(A.dst)[(size_t)i * A.w + j] = make_float2(u, v);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void extract_ch0_kernel(const uint8_t *src3, uint8_t *dst1, int w, int h, int pitch)
{
    const int x0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y;
    if (x0 >= pitch || y >= h) return;
    const uint8_t *s = src3 + 3 * ((size_t)y * w + x0);
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (x0 + k < w) out |= (uint32_t)s[3 * k] << (8 * k);
    *reinterpret_cast<uint32_t *>(dst1 + (size_t)y * pitch + x0) = out;
}

__global__ __launch_bounds__(256) void replicate_3ch_kernel(const uint8_t *src1, int pitch, uint8_t *dst3, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t v = src1[(size_t)y * pitch + x];
    uint8_t *d = dst3 + 3 * ((size_t)y * w + x);
    d[0] = d[1] = d[2] = v;
}

} // namespace

extern "C" int ofx_downsample_1ch(const uint8_t *d_src, int src_pitch, int src_row0, int src_rows, uint8_t *d_dst,
                                  const ofx_geom *dst, void *stream)
{
    OFX_TRY(ofx_check_geom(dst, "ofx_downsample_1ch"));
    OFX_REQUIRE(d_src && d_dst, "ofx_downsample_1ch: null pointer");
    OFX_REQUIRE(src_pitch >= 2 * dst->w && (src_pitch & 3) == 0, "ofx_downsample_1ch: src pitch %d too small for width %d",
                src_pitch, 2 * dst->w);
    OFX_REQUIRE(((uintptr_t)d_src & 3) == 0 && ((uintptr_t)d_dst & 3) == 0, "ofx_downsample_1ch: planes must be 4-byte aligned");
    OFX_REQUIRE(dst->out_y0 >= dst->row0 && dst->out_y1 <= dst->row0 + dst->rows,
                "ofx_downsample_1ch: output rows [%d,%d) not inside the destination buffer [%d,%d)", dst->out_y0, dst->out_y1,
                dst->row0, dst->row0 + dst->rows);
    if (dst->out_y1 <= dst->out_y0) return OFX_OK;
    // source rows 2*y-1 .. 2*y+1 (row -1 is the border)
    const int need_lo = 2 * dst->out_y0 - 1 > 0 ? 2 * dst->out_y0 - 1 : 0;
    const int need_hi = 2 * (dst->out_y1 - 1) + 2;
    OFX_REQUIRE(need_lo >= src_row0 && need_hi <= src_row0 + src_rows,
                "ofx_downsample_1ch: source rows [%d,%d) needed, buffer holds [%d,%d)", need_lo, need_hi, src_row0,
                src_row0 + src_rows);
    DownArgs a{d_src,    d_dst,       src_pitch,  src_row0,   src_row0 + src_rows, dst->w,
               dst->h,   dst->pitch,  dst->row0,  dst->out_y0, dst->out_y1};
    const int groups = ofx_div_up(dst->w, 4);
    dim3 grid(ofx_div_up(groups, 256), dst->out_y1 - dst->out_y0);
    hipLaunchKernelGGL(downsample_1ch_kernel, grid, dim3(256), 0, ofx_stream(stream), a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

// Fills the arguments of the fused pyramid stage; returns the dynamic LDS it needs and its grid (shared with the
// stream kernel's launcher in lk_level.hip).
int ofx_pyramid_args(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches, int levels,
                     uint8_t *d_level0_copy, int copy_pitch, const int *row0, const int *rows, PyrArgs *out, size_t *lds_bytes,
                     int *blocks_x, int *blocks_y)
{
    OFX_REQUIRE(d_level0 && d_levels && pitches && w > 0 && h > 0, "ofx_pyramid_1ch: bad arguments");
    OFX_REQUIRE(levels >= 2 && levels - 1 <= kPyrMaxProduced, "ofx_pyramid_1ch: %d levels unsupported (2..%d)", levels,
                kPyrMaxProduced + 1);
    OFX_REQUIRE(((uintptr_t)d_level0 & 3) == 0 && (pitch0 & 3) == 0, "ofx_pyramid_1ch: level 0 must be 4-byte aligned with a pitch multiple of 4");
    PyrArgs a{};
    a.src = d_level0;
    a.pitch[0] = pitch0;
    a.w[0] = w;
    a.h[0] = h;
    a.n = levels - 1;
    a.dst[0] = d_level0_copy;
    a.dst0_pitch = copy_pitch;
    for (int k = 1; k < levels; ++k) {
        OFX_REQUIRE(((w >> (k - 1)) & 1) == 0 && ((h >> (k - 1)) & 1) == 0, "ofx_pyramid_1ch: level %d has odd dimensions", k - 1);
        OFX_REQUIRE(d_levels[k] != nullptr && pitches[k] >= (w >> k), "ofx_pyramid_1ch: bad plane for level %d", k);
        a.dst[k] = d_levels[k];
        a.pitch[k] = pitches[k];
        a.w[k] = w >> k;
        a.h[k] = h >> k;
    }
    const int H0 = (1 << a.n) - 1;
    // Levels 1 .. n-1 live in LDS (level 0 is read from HBM directly, level n is only written out), in two areas used
    // alternately: even levels in area 0, odd levels in area 1, each with 16 bytes of slack in front (a thread's leftmost
    // source dword may start 4 bytes before its row).  Column of pixel x: x - X_k + delta[k], delta[n-1] = the halo of that
    // level rounded up to 4, delta[k] = 2*delta[k+1]; a row holds delta + tile + 12 bytes (groups of 4 that reach past the tile).
    size_t area[2] = {0, 0};
    for (int k = a.n - 1; k >= 1; --k) {
        const int Hk = H0 >> k, Tk = kPyrTile >> k;
        a.delta[k] = k == a.n - 1 ? ((Hk + 3) & ~3) : 2 * a.delta[k + 1];
        a.stride[k] = (Tk + a.delta[k] + 12 + 3) & ~3;
        const size_t bytes = 16 + (size_t)a.stride[k] * (size_t)(Tk + Hk + 1);
        if (bytes > area[k & 1]) area[k & 1] = bytes;
    }
    area[0] = (area[0] + 15) & ~(size_t)15;
    for (int k = 0; k <= a.n; ++k) a.lds_off[k] = 16 + ((k & 1) ? (int)area[0] : 0);
    *lds_bytes = area[0] + ((area[1] + 15) & ~(size_t)15);
    // destination row windows (row0 == NULL: whole levels).  The tiles launched are those that intersect any window,
    // expressed in level-0 rows: a level-k row y belongs to the tile row (y << k) / kPyrTile.
    int t0 = 0, t1 = ofx_div_up(h, kPyrTile);
    for (int k = 0; k < levels; ++k) {
        a.row0[k] = row0 ? row0[k] : 0;
        a.row1[k] = row0 ? row0[k] + rows[k] : (h >> k);
        OFX_REQUIRE(a.row0[k] >= 0 && a.row0[k] <= a.row1[k] && a.row1[k] <= (h >> k), "ofx_pyramid_1ch: bad row window at level %d", k);
    }
    if (row0) {
        int lo = h, hi = 0;
        for (int k = 0; k < levels; ++k) {
            if (a.row1[k] <= a.row0[k]) continue;
            lo = (a.row0[k] << k) < lo ? (a.row0[k] << k) : lo;
            hi = (a.row1[k] << k) > hi ? (a.row1[k] << k) : hi;
        }
        t0 = lo / kPyrTile;
        t1 = hi > lo ? ofx_div_up(hi, kPyrTile) : t0;
    }
    a.by0 = t0;
    *blocks_x = ofx_div_up(w, kPyrTile);
    *blocks_y = t1 - t0;
    *out = a;
    return OFX_OK;
}

// Arguments of the marching pyramid (pyr_march.h, the stream kernel's pyramid stage); *items = waves it needs.
int ofx_pyramid_march_args(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches, int levels,
                           uint8_t *d_level0_copy, int copy_pitch, const int *row0, const int *rows, int target_waves, PyrMarchArgs *out,
                           int *items)
{
    OFX_REQUIRE(d_level0 && d_levels && pitches && w > 0 && h > 0, "ofx_stream_launch: bad pyramid arguments");
    OFX_REQUIRE(levels >= 2 && levels - 1 <= kMarchMaxProduced, "ofx_stream_launch: %d levels unsupported (2..%d)", levels, kMarchMaxProduced + 1);
    OFX_REQUIRE(((uintptr_t)d_level0 & 3) == 0 && (pitch0 & 3) == 0 && pitch0 >= 8 && pitch0 >= w,
                "ofx_stream_launch: level 0 must be 4-byte aligned with a pitch that is a multiple of 4, >= 8 and >= the width");
    PyrMarchArgs a{};
    a.src = d_level0;
    a.src_pitch = pitch0;
    a.n = levels - 1;
    a.dst[0] = d_level0_copy;
    a.pitch[0] = copy_pitch;
    a.w[0] = w;
    a.h[0] = h;
    for (int k = 1; k < levels; ++k) {
        OFX_REQUIRE(((w >> (k - 1)) & 1) == 0 && ((h >> (k - 1)) & 1) == 0, "ofx_stream_launch: level %d has odd dimensions", k - 1);
        OFX_REQUIRE(d_levels[k] != nullptr && pitches[k] >= (w >> k) && (pitches[k] & 3) == 0, "ofx_stream_launch: bad plane for level %d", k);
        a.dst[k] = d_levels[k];
        a.pitch[k] = pitches[k];
        a.w[k] = w >> k;
        a.h[k] = h >> k;
    }
    OFX_REQUIRE(d_level0_copy == nullptr || (copy_pitch >= w && (copy_pitch & 3) == 0), "ofx_stream_launch: bad level-0 copy pitch");
    const int step = 1 << a.n;
    int lo = h, hi = 0;
    for (int k = 0; k < levels; ++k) {
        a.row0[k] = row0 ? row0[k] : 0;
        a.row1[k] = row0 ? row0[k] + rows[k] : (h >> k);
        OFX_REQUIRE(a.row0[k] >= 0 && a.row0[k] <= a.row1[k] && a.row1[k] <= (h >> k), "ofx_stream_launch: bad row window at level %d", k);
        if (a.row1[k] <= a.row0[k] || (k == 0 && d_level0_copy == nullptr)) continue;
        lo = (a.row0[k] << k) < lo ? (a.row0[k] << k) : lo;
        hi = (a.row1[k] << k) > hi ? (a.row1[k] << k) : hi;
    }
    if (hi <= lo) {
        *out = a;
        *items = 0;
        return OFX_OK;
    }
    a.y_lo = lo / step * step;
    a.y_hi = hi < h ? hi : h;
    // lanes of overlap on the left: 8*halo >= 2^n - 1, and the tile width a multiple of 2^n (pixels of the deepest level
    // sit in every 2^(n-3)-th lane)
    a.halo = a.n <= 3 ? 1 : 1 << (a.n - 3);
    a.tile_w = (64 - a.halo) * 8;
    a.tiles_x = ofx_div_up(w, a.tile_w);
    // Strips.  A marching wave runs at top priority next to the LK waves of its SIMD and takes its instructions out of their
    // issue slots, so the waves should spread over the machine rather than pile work on a few SIMDs: the caller names the
    // number of waves it wants (about 0.85 per SIMD for all the pyramids of a tick); each strip pays 2^n priming rows, so
    // a strip is at least twice that (and at most 128 rows when the caller names no target).
    const int span = a.y_hi - a.y_lo;
    const int want = target_waves > 0 ? target_waves : 16 * a.tiles_x;
    int sh = ofx_div_up(ofx_div_up(span * a.tiles_x, want), step) * step;
    sh = sh < 2 * step ? 2 * step : sh;
    if (target_waves <= 0) sh = sh > 128 ? 128 / step * step : sh; // (a named target is a budget of wave slots: never exceed it)
    a.strip_h = sh;
    a.strips = ofx_div_up(span, sh);
    *out = a;
    *items = a.tiles_x * a.strips;
    return OFX_OK;
}

extern "C" int ofx_pyramid_1ch(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches,
                               int levels, void *stream)
{
    PyrArgs a{};
    size_t lds_bytes = 0;
    int bx = 0, by = 0;
    OFX_TRY(ofx_pyramid_args(d_level0, pitch0, w, h, d_levels, pitches, levels, nullptr, 0, nullptr, nullptr, &a, &lds_bytes, &bx, &by));
    hipLaunchKernelGGL(pyramid_fused_kernel, dim3(bx, by), dim3(kPyrThreads), lds_bytes, ofx_stream(stream), a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

extern "C" int ofx_downsample_3ch(const uint8_t *d_src3, uint8_t *d_dst3, int dw, int dh, void *stream)
{
    OFX_REQUIRE(d_src3 && d_dst3 && dw > 0 && dh > 0, "ofx_downsample_3ch: bad arguments");
    dim3 grid(ofx_div_up(dw, 256), dh);
    hipLaunchKernelGGL(downsample_3ch_kernel, grid, dim3(256), 0, ofx_stream(stream), d_src3, d_dst3, dw, dh);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

extern "C" int ofx_shift_vector(const float *const *d_flow_levels, int level, int max_level, float *d_uv, void *stream)
{
    OFX_REQUIRE(d_flow_levels && d_uv, "ofx_shift_vector: null pointer");
    OFX_REQUIRE(max_level >= 1 && max_level <= OFX_MAX_LEVELS && level >= 0 && level < max_level,
                "ofx_shift_vector: bad level %d of %d", level, max_level);
    FlowPtrs p{};
    for (int k = level + 1; k < max_level; ++k) {
        OFX_REQUIRE(d_flow_levels[k] != nullptr, "ofx_shift_vector: flow level %d is null", k);
        p.lv[k] = d_flow_levels[k];
    }
    hipLaunchKernelGGL(shift_vector_kernel, dim3(1), dim3(64), 0, ofx_stream(stream), p, level, max_level, d_uv);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

int ofx_shift_table(const ofx_shift_desc *levels, int n, ShiftTable *out, int *blocks_out)
{
    OFX_REQUIRE(levels && n >= 1 && n <= OFX_MAX_LK_ITEMS, "ofx_shift_levels: bad descriptor count %d", n);
    ShiftTable t{};
    int blocks = 0, m = 0;
    for (int i = 0; i < n; ++i) {
        const ofx_geom *g = &levels[i].geom;
        OFX_TRY(ofx_check_geom(g, "ofx_shift_1ch"));
        OFX_REQUIRE(levels[i].d_src && levels[i].d_dst && levels[i].d_uv, "ofx_shift_1ch: null pointer");
        OFX_REQUIRE(levels[i].d_src != levels[i].d_dst, "ofx_shift_1ch: in-place shift is not supported");
        OFX_REQUIRE(g->out_y0 >= g->row0 && g->out_y1 <= g->row0 + g->rows, "ofx_shift_1ch: output rows outside the buffer");
        if (g->out_y1 <= g->out_y0) continue;
        const int bx = ofx_div_up(ofx_div_up(g->pitch, 16), 256);
        t.lv[m] = ShiftArgs{levels[i].d_src, levels[i].d_dst, levels[i].d_uv, g->w, g->h, g->pitch, g->row0, g->row0 + g->rows,
                            g->out_y0, g->out_y1, bx};
        t.first_block[m] = blocks;
        blocks += bx * ofx_div_up(g->out_y1 - g->out_y0, ofx_dev::kShiftRows);
        ++m;
    }
    t.n = m;
    t.first_block[m] = blocks;
    *out = t;
    *blocks_out = blocks;
    return OFX_OK;
}

extern "C" int ofx_shift_levels(const ofx_shift_desc *levels, int n, void *stream)
{
    ShiftTable t{};
    int blocks = 0;
    OFX_TRY(ofx_shift_table(levels, n, &t, &blocks));
    if (blocks == 0) return OFX_OK;
    hipLaunchKernelGGL(shift_1ch_kernel, dim3((unsigned)blocks), dim3(256), 0, ofx_stream(stream), t);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

extern "C" int ofx_shift_1ch(const uint8_t *d_src, uint8_t *d_dst, const ofx_geom *g, const float *d_uv, void *stream)
{
    OFX_REQUIRE(g != nullptr, "ofx_shift_1ch: geometry is null");
    ofx_shift_desc d{d_src, d_dst, *g, d_uv};
    return ofx_shift_levels(&d, 1, stream);
}

extern "C" int ofx_warp_levels(const ofx_warp_desc *levels, int n, void *stream)
{
    OFX_REQUIRE(levels && n >= 1 && n <= OFX_MAX_LK_ITEMS, "ofx_warp_levels: bad descriptor count %d", n);
    WarpTable t{};
    int blocks = 0, m = 0;
    for (int i = 0; i < n; ++i) {
        const ofx_geom *g = &levels[i].geom;
        OFX_TRY(ofx_check_geom(g, "ofx_warp_levels"));
        OFX_REQUIRE(levels[i].d_src && levels[i].d_dst && levels[i].d_flow, "ofx_warp_levels: null pointer");
        OFX_REQUIRE(levels[i].d_src != levels[i].d_dst, "ofx_warp_levels: in-place warp is not supported");
        OFX_REQUIRE(g->out_y0 >= g->row0 && g->out_y1 <= g->row0 + g->rows, "ofx_warp_levels: output rows outside the planes");
        OFX_REQUIRE(levels[i].flow_row0 <= g->out_y0, "ofx_warp_levels: flow_row0 > out_y0");
        if (g->out_y1 <= g->out_y0) continue;
        const int bx = ofx_div_up(g->pitch / 4, 256);
        t.lv[m] = WarpArgs{levels[i].d_src, levels[i].d_dst, levels[i].d_flow, levels[i].scale, g->w, g->h, g->pitch, g->row0,
                           g->out_y0, g->out_y1, levels[i].flow_row0, bx, g->row0 + g->rows, levels[i].d_status, levels[i].status_bit};
        t.first_block[m] = blocks;
        blocks += bx * (g->out_y1 - g->out_y0);
        ++m;
    }
    if (m == 0) return OFX_OK;
    t.n = m;
    t.first_block[m] = blocks;
    hipLaunchKernelGGL(warp_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, ofx_stream(stream), t);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

extern "C" int ofx_compose_flow(const float *const *d_flow_levels, int w, int h, int levels, int level, float *d_dst,
                                void *stream)
{
    OFX_REQUIRE(d_flow_levels && d_dst && w > 0 && h > 0, "ofx_compose_flow: bad arguments");
    OFX_REQUIRE(levels >= 1 && levels <= OFX_MAX_LEVELS && level >= 0 && level < levels, "ofx_compose_flow: bad level");
    ComposeArgs a{};
    for (int k = level; k < levels; ++k) {
        OFX_REQUIRE(d_flow_levels[k] != nullptr, "ofx_compose_flow: flow level %d is null", k);
        a.P.lv[k] = d_flow_levels[k];
    }
    a.dst = d_dst;
    a.w = w;
    a.h = h;
    a.levels = levels;
    a.level = level;
    dim3 grid(ofx_div_up(w, 256), h);
    hipLaunchKernelGGL(compose_flow_kernel, grid, dim3(256), 0, ofx_stream(stream), a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

extern "C" int ofx_extract_ch0(const uint8_t *d_src3, uint8_t *d_dst1, int w, int h, int dst_pitch, void *stream)
{
    OFX_REQUIRE(d_src3 && d_dst1 && w > 0 && h > 0, "ofx_extract_ch0: bad arguments");
    OFX_REQUIRE(dst_pitch >= w && (dst_pitch & 3) == 0 && ((uintptr_t)d_dst1 & 3) == 0, "ofx_extract_ch0: bad pitch/alignment");
    dim3 grid(ofx_div_up(dst_pitch / 4, 256), h);
    hipLaunchKernelGGL(extract_ch0_kernel, grid, dim3(256), 0, ofx_stream(stream), d_src3, d_dst1, w, h, dst_pitch);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

extern "C" int ofx_replicate_3ch(const uint8_t *d_src1, int src_pitch, uint8_t *d_dst3, int w, int h, void *stream)
{
    OFX_REQUIRE(d_src1 && d_dst3 && w > 0 && h > 0 && src_pitch >= w, "ofx_replicate_3ch: bad arguments");
    dim3 grid(ofx_div_up(w, 256), h);
    hipLaunchKernelGGL(replicate_3ch_kernel, grid, dim3(256), 0, ofx_stream(stream), d_src1, src_pitch, d_dst3, w, h);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
