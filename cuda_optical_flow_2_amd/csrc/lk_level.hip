// Fused dense Lucas-Kanade level kernel for gfx950 (MI355X).
//
// One kernel does what the reference does in ten launches plus two host loops per level
// (OptFlowGpu.cu:1930-1964 / OptFlowCPU.cpp:329-384): 3x3 derivative stencils, the five windowed sums of
// products and the 2x2 solve.  Algorithmic HBM traffic: 2 bytes read + 8 bytes written per pixel.
//
// Structure (DESIGN.md "lk_level"):
//   * one 64-lane wave per workgroup; a lane owns 4 adjacent columns, so a wave spans 256 image columns and
//     every global load is one aligned dword per lane (256 B per wave instruction), every flow store 32 B per lane.
//   * the wave marches DOWN a strip of rows.  Per step it loads one new row of each image, forms the three
//     derivative values of its 4 columns (left/right neighbour columns come from the adjacent lanes through
//     DPP wave shifts, no LDS), and updates 5 x 4 vertical running sums:  V += P(row entering) - P(row leaving).
//     The leaving row's derivatives are recomputed from the image (its rows are L2-resident) in the high halves of the
//     same packed-int16 instructions that compute the entering row's (lk_body.h).
//   * the horizontal half of the box sum is done in registers: in-lane prefix/suffix sums plus whole-lane totals
//     of the neighbouring lanes, again through DPP (a sliding difference per output for windows up to 9x9).
//   * the only use of LDS: each output row is exchanged through it so that both streaming store instructions cover
//     gap-free 128-byte lines (non-temporal stores of 16-byte pieces run at half the write bandwidth).
//   * all sums are exact int32, so results do not depend on strip/tile/shard boundaries.
//   * window radius R and mode are template parameters; the host dispatches.
//
// Border semantics follow the reference exactly: image taps outside the image contribute nothing
// (OptFlowCPU.cpp:98, OptFlowGpu.cu:1066-1075) and window taps outside the image are skipped
// (OptFlowCPU.cpp:182-191) -- i.e. the image and the derivative planes are zero-extended.
#include <stdlib.h>

#include "corner_body.h"
#include "lk_body.h"
#include "pyr_march.h"
#include "stages_body.h"

using namespace ofx_dev;

namespace {

#ifndef OFX_LK_MIN_WAVES
#define OFX_LK_MIN_WAVES(R) 3 // A/B on MI355X: capping at 128 VGPRs (4 waves) spills in the marching loop and is slower
#endif
template <int R, int MODE, bool SUMS, bool FAST>
__global__ __launch_bounds__(64, OFX_LK_MIN_WAVES(R)) void lk_level_kernel(const LkTable T)
{
    __shared__ __attribute__((aligned(16))) uint8_t xlds[kLkWaveLds];
    lk_wave<R, MODE, SUMS, true, FAST>(T, (int)blockIdx.x, (int)threadIdx.x, xlds);
}

// ---- the stream kernel: one launch = one pipeline tick ---------------------------------------------------------------
// A tick of a frame stream runs, as disjoint block ranges of ONE grid,
//     pyramid(newest frame(s))  |  corner flows(earlier pair(s))  |  fused LK(still earlier pair(s))
// Each stage consumes what earlier launches wrote, so there is no synchronisation inside the launch and none between
// streams; the small latency-bound stages run in the shadow of the VALU-bound LK stage.  Blocks are 256 threads; an LK
// block is four independent LK waves; the corner block runs one wave per pair.
constexpr int kPyrStages = 2 * OFX_STREAM_MAX_BATCH; // per frame of the tick: its pyramid and its top-left patch pyramid
constexpr int kCornerScratch = 128;                  // LDS of a corner block: the chain's floats, then the cached corners
struct StreamArgs {
    LkTable lk;
    PyrMarchArgs pyr[kPyrStages];
    CornerArgs corner[OFX_STREAM_MAX_BATCH];
    // blocks [0, OFX_STREAM_MAX_BATCH) = one corner wave each; [.., first[0]) LK (four waves per block);
    // [first[i], first[i+1]) pyramid stage i (four marching waves per block, pyr_march.h).
    // The LK blocks come first and are planned for a whole number of waves per SIMD (lk_wave_target): they all start at
    // once and run for the whole launch, while the short staging blocks stream through the remaining slots underneath.
    int first[kPyrStages + 1];
    int n_corner;
    unsigned long long *trace; // optional (ofx_debug_stream_trace): per block, start and end time (100 MHz wall clock)
    int trace_blocks;
};

unsigned long long *g_stream_trace = nullptr; // tools/stream_timeline.py
int g_stream_trace_blocks = 0;
int g_trace_header[kPyrStages + 1] = {0};

// lk_float fits 5 blocks per CU (<= 96 VGPRs) without scratch for every radius; compat_cpu needs ~120: 4 blocks (<= 128)
#ifndef OFX_STREAM_MIN_BLOCKS
#define OFX_STREAM_MIN_BLOCKS(R, MODE) ((MODE) == OFX_MODE_LK_FLOAT ? 5 : 4)
#endif
template <int R, int MODE, bool FAST>
__global__ __launch_bounds__(256, OFX_STREAM_MIN_BLOCKS(R, MODE)) void stream_kernel(const StreamArgs S)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const unsigned long long t_start = S.trace ? wall_clock64() : 0ull;
    // readfirstlane: the wave index is uniform, and everything derived from it (strip rows, row pointers, loop counters)
    // must live in SGPRs as it does in the stand-alone kernel
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (b < OFX_STREAM_MAX_BATCH) {
        // one corner chain per block (wave 0), so that the chains land on different CUs
        // the short latency-bound stages go first whenever they are ready to issue (the LK waves lower their own priority
        // from 3 to 0 as they advance, lk_body.h)
        __builtin_amdgcn_s_setprio(3);
        if (b < S.n_corner && wv == 0) corner_wave<MODE, FAST>(S.corner[b], tid & 63, reinterpret_cast<float *>(lds), lds + kCornerScratch);
    } else if (b < S.first[0]) {
        lk_wave<R, MODE, false, false, FAST>(S.lk, 4 * (b - OFX_STREAM_MAX_BATCH) + wv, tid & 63, lds + wv * kLkWaveLds);
    } else {
        int i = 0;
        while (i + 1 < kPyrStages && b >= S.first[i + 1]) ++i;
        __builtin_amdgcn_s_setprio(3);
        pyr_march_wave(S.pyr[i], 4 * (b - S.first[i]) + wv, tid & 63);
    }
    if (S.trace && b < S.trace_blocks && (tid & 63) == 0) { // one record per wave: 4 per block
        // where the wave ran: HW_ID (wave / SIMD / CU / SH / SE) in bits 32.., XCC_ID in bits 48.. of the start word's top
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (15 << 11));  // HW_REG_HW_ID, bits 0..15
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID, bits 0..3
        S.trace[2 * (4 * b + wv)] = t_start;
        S.trace[2 * (4 * b + wv) + 1] = (wall_clock64() & 0x0000ffffffffffffull) | ((unsigned long long)(hw & 0xffffu) << 48);
        S.trace[2 * (4 * b + wv)] = (t_start & 0x0000ffffffffffffull) | ((unsigned long long)(xcc & 0xfu) << 48);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct LkLevelIn {
    LkArgs a;     // everything but strip_h / tiles_x
    int rows_out;
};

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && atoi(e) > 0 ? atoi(e) : dflt;
}

// One strip height for all levels (so all waves run about equally long): the smallest that keeps the wave count within
// `capacity` (lk_wave_target), but at least `min_h` so the 2R priming rows of a strip stay a minor cost.
// The grid is sized to fit in ONE residency round: every wave runs for the whole kernel, so a second, partly filled
// round would nearly double the run time.
template <int R>
int plan_table(const LkLevelIn *lv, int n, int capacity, LkTable *out)
{
    using G = TileGeom<R>;
    const int min_h = env_int("OFX_LK_MIN_STRIP", 8);
    int max_rows = 1;
    for (int i = 0; i < n; ++i) max_rows = lv[i].rows_out > max_rows ? lv[i].rows_out : max_rows;
    int strip_h = min_h;
    for (; strip_h < max_rows; ++strip_h) {
        long waves = 0;
        for (int i = 0; i < n; ++i) waves += (long)ofx_div_up(lv[i].a.w, G::OUT_W) * ofx_div_up(lv[i].rows_out, strip_h);
        if (waves <= (long)capacity) break;
    }
    LkTable t{};
    t.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        t.lv[i] = lv[i].a;
        t.lv[i].tiles_x = ofx_div_up(lv[i].a.w, G::OUT_W);
        t.lv[i].strip_h = strip_h < lv[i].rows_out ? strip_h : lv[i].rows_out;
        t.first_block[i] = blocks;
        blocks += t.lv[i].tiles_x * ofx_div_up(lv[i].rows_out, t.lv[i].strip_h);
    }
    t.first_block[n] = blocks;
    *out = t;
    return blocks;
}

// Number of LK waves a launch is planned for.  Every LK wave runs for the whole launch, so what matters is how many of
// them share a SIMD: measured on MI355X (4K, 9x9) 3 per SIMD is the optimum once the march no longer waits on its own
// loads -- fewer leave issue slots empty, more shorten the strips (each strip pays 2R priming rows) -- and a count that
// is not a whole number per SIMD makes the fuller SIMDs set the time.  `reserve` slots per SIMD are left to the other
// stages of the stream kernel.
template <typename K>
int lk_wave_target(K kernel, int threads, size_t lds, int reserve, int dflt_per_simd)
{
    int dev = 0, cus = 256, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu <= 0) per_cu = 8 * 64 / threads;
    (void)hipGetLastError();
    const int occ = per_cu * (threads / 64) / 4; // waves per SIMD (4 SIMDs per CU)
    int per_simd = env_int("OFX_LK_WAVES_PER_SIMD", dflt_per_simd);
    if (per_simd > occ - reserve) per_simd = occ - reserve;
    if (per_simd < 1) per_simd = 1;
    // with wave slots to spare the plan may use the whole target (an uneven placement still fits in one round); a plan
    // that needs every slot keeps 5 % back, because a second, mostly empty round would double the run time
    const int fill = env_int("OFX_LK_FILL", per_simd < occ ? 100 : 95);
    return env_int("OFX_LK_TARGET_WAVES", (int)((long)cus * 4 * per_simd * fill / 100));
}

template <int R, int MODE, bool SUMS, bool FAST>
int launch_r(const LkLevelIn *lv, int n, hipStream_t st)
{
    static const int capacity = lk_wave_target(lk_level_kernel<R, MODE, SUMS, FAST>, 64, 0, 0, 3);
    LkTable t{};
    const int blocks = plan_table<R>(lv, n, capacity, &t);
    hipLaunchKernelGGL((lk_level_kernel<R, MODE, SUMS, FAST>), dim3((unsigned)blocks), dim3(64), 0, st, t);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

template <int R, int MODE, bool FAST>
int launch_stream_r(const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    // Next to the staging blocks the LK stage does best with 2 waves per SIMD when the tick carries one pair and 4 when it
    // carries more (measured, 4K: one pair 58.2 / 59.8 us per frame at 2 / 3; two pairs 59.7 / 57.2 / 56.5 at 2 / 3 / 4)
    static const int capacity1 = lk_wave_target(stream_kernel<R, MODE, FAST>, 256, 16 * 1024, 1, 2);
    static const int capacity2 = lk_wave_target(stream_kernel<R, MODE, FAST>, 256, 16 * 1024, 1, 4);
    int pairs = 0;
    for (int i = 0; i < n; ++i) pairs += (lv[i].a.w == lv[0].a.w && lv[i].a.h == lv[0].a.h) ? 1 : 0;
    const int capacity = pairs >= 2 ? capacity2 : capacity1;
    int lk_blocks = 0;
    if (n > 0) lk_blocks = ofx_div_up(plan_table<R>(lv, n, capacity, &S.lk), 4);
    S.first[0] = OFX_STREAM_MAX_BATCH + lk_blocks;
    for (int i = 0; i < kPyrStages; ++i) S.first[i + 1] = S.first[i] + stage_blocks[i];
    const int blocks = S.first[kPyrStages];
    S.trace = g_stream_trace;
    S.trace_blocks = g_stream_trace_blocks;
    if (g_stream_trace) // header: block ranges of this launch
        for (int i = 0; i <= kPyrStages; ++i) g_trace_header[i] = S.first[i];
    size_t corner_lds = 0; // a corner wave's scratch: the chain's floats and the cached corners of its levels
    for (int i = 0; i < S.n_corner; ++i) {
        const size_t need = (size_t)kCornerScratch + kCornerTileBytes + (size_t)S.corner[i].levels * kCornerCacheBytes;
        corner_lds = need > corner_lds ? need : corner_lds;
    }
    if (lds < corner_lds) lds = corner_lds;
    if (lds < 4 * (size_t)kLkWaveLds) lds = 4 * (size_t)kLkWaveLds; // an LK block: four waves, each with its exchange row
    hipLaunchKernelGGL((stream_kernel<R, MODE, FAST>), dim3((unsigned)blocks), dim3(256), lds, st, S);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

template <int MODE, bool FAST>
int launch_stream_mode(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_stream_r<1, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 2: return launch_stream_r<2, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 3: return launch_stream_r<3, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 4: return launch_stream_r<4, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 5: return launch_stream_r<5, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 6: return launch_stream_r<6, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 7: return launch_stream_r<7, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 8: return launch_stream_r<8, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 9: return launch_stream_r<9, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 10: return launch_stream_r<10, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    case 11: return launch_stream_r<11, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    default: break;
    }
    if constexpr (MODE == OFX_MODE_COMPAT_CPU) {
        if (radius == 12) return launch_stream_r<12, MODE, FAST>(lv, n, S, stage_blocks, lds, st);
    }
    ofx_set_error("ofx_stream_launch: window %d not supported in mode %d", 2 * radius + 1, MODE);
    return OFX_E_UNSUPPORTED;
}

template <int MODE, bool SUMS, bool FAST>
int launch_mode(int radius, const LkLevelIn *lv, int n, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_r<1, MODE, SUMS, FAST>(lv, n, st);
    case 2: return launch_r<2, MODE, SUMS, FAST>(lv, n, st);
    case 3: return launch_r<3, MODE, SUMS, FAST>(lv, n, st);
    case 4: return launch_r<4, MODE, SUMS, FAST>(lv, n, st);
    case 5: return launch_r<5, MODE, SUMS, FAST>(lv, n, st);
    case 6: return launch_r<6, MODE, SUMS, FAST>(lv, n, st);
    case 7: return launch_r<7, MODE, SUMS, FAST>(lv, n, st);
    case 8: return launch_r<8, MODE, SUMS, FAST>(lv, n, st);
    case 9: return launch_r<9, MODE, SUMS, FAST>(lv, n, st);
    case 10: return launch_r<10, MODE, SUMS, FAST>(lv, n, st);
    case 11: return launch_r<11, MODE, SUMS, FAST>(lv, n, st);
    default: break;
    }
    if constexpr (MODE == OFX_MODE_COMPAT_CPU) {
        if (radius == 12) return launch_r<12, MODE, SUMS, FAST>(lv, n, st);
    }
    ofx_set_error("ofx_lk_level: window %d not supported in mode %d", 2 * radius + 1, MODE);
    return OFX_E_UNSUPPORTED;
}

int lk_build_levels(const ofx_lk_desc *d, int n, int window, int mode, int32_t *d_sums, LkLevelIn *lv, int *count)
{
    OFX_REQUIRE(d != nullptr && n >= 1 && n <= OFX_MAX_LK_ITEMS, "ofx_lk_levels: bad descriptor count %d", n);
    OFX_REQUIRE(window >= 3 && (window & 1), "ofx_lk_level: window must be odd and >= 3 (got %d)", window);
    OFX_REQUIRE(mode == OFX_MODE_COMPAT_CPU || mode == OFX_MODE_LK_FLOAT || mode == OFX_MODE_LK_FLOAT_FAST, "ofx_lk_level: bad mode %d", mode);
    const int radius = window >> 1;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const ofx_geom *g = &d[i].geom;
        OFX_TRY(ofx_check_geom(g, "ofx_lk_level"));
        OFX_REQUIRE(d[i].d_prev && d[i].d_next && (d[i].d_flow || d_sums), "ofx_lk_level: null pointer");
        OFX_REQUIRE(((uintptr_t)d[i].d_prev & 3) == 0 && ((uintptr_t)d[i].d_next & 3) == 0, "ofx_lk_level: planes must be 4-byte aligned");
        OFX_REQUIRE(d[i].flow_row0 <= g->out_y0, "ofx_lk_level: flow_row0 %d > out_y0 %d", d[i].flow_row0, g->out_y0);
        OFX_TRY(ofx_check_halo(g, radius + 1, "ofx_lk_level"));
        const int rows_out = g->out_y1 - g->out_y0;
        if (rows_out <= 0) continue;
        LkArgs a{};
        a.prev = d[i].d_prev;
        a.next = d[i].d_next;
        a.uv = d[i].d_uv;
        a.accumulate = d[i].accumulate;
        a.min_det = d[0].min_det; // one value per launch
        a.flow = d[i].d_flow;
        a.sums = d_sums;
        // plane stride of the inspection output = rows from flow_row0 to out_y1
        a.sums_plane = (size_t)(g->out_y1 - d[i].flow_row0) * (size_t)g->w;
        a.w = g->w;
        a.h = g->h;
        a.pitch = g->pitch;
        a.row0 = g->row0;
        a.row_end = g->row0 + g->rows;
        a.out_y0 = g->out_y0;
        a.out_y1 = g->out_y1;
        a.flow_row0 = d[i].flow_row0;
        lv[m].a = a;
        lv[m].rows_out = rows_out;
        ++m;
    }
    *count = m;
    return OFX_OK;
}

int lk_dispatch(const ofx_lk_desc *d, int n, int window, int mode, int32_t *d_sums, void *stream)
{
    LkLevelIn lv[OFX_MAX_LK_ITEMS];
    int m = 0;
    OFX_TRY(lk_build_levels(d, n, window, mode, d_sums, lv, &m));
    if (m == 0) return OFX_OK;
    const int radius = window >> 1;
    hipStream_t st = ofx_stream(stream);
    if (d_sums) { // the sums do not depend on the solve
        return mode != OFX_MODE_COMPAT_CPU ? launch_mode<OFX_MODE_LK_FLOAT, true, false>(radius, lv, m, st)
                                           : launch_mode<OFX_MODE_COMPAT_CPU, true, false>(radius, lv, m, st);
    }
    if (mode == OFX_MODE_LK_FLOAT_FAST) return launch_mode<OFX_MODE_LK_FLOAT, false, true>(radius, lv, m, st);
    return mode == OFX_MODE_LK_FLOAT ? launch_mode<OFX_MODE_LK_FLOAT, false, false>(radius, lv, m, st)
                                     : launch_mode<OFX_MODE_COMPAT_CPU, false, false>(radius, lv, m, st);
}

} // namespace

extern "C" int ofx_stream_launch(const ofx_stream_stages *g, int window, int mode, void *stream)
{
    OFX_REQUIRE(g != nullptr, "ofx_stream_launch: null argument");
    OFX_REQUIRE(window >= 3 && (window & 1), "ofx_stream_launch: window must be odd and >= 3 (got %d)", window);
    OFX_REQUIRE(mode == OFX_MODE_COMPAT_CPU || mode == OFX_MODE_LK_FLOAT || mode == OFX_MODE_LK_FLOAT_FAST, "ofx_stream_launch: bad mode %d", mode);
    OFX_REQUIRE(g->n_pyr >= 0 && g->n_pyr <= OFX_STREAM_MAX_BATCH && g->n_corner >= 0 && g->n_corner <= OFX_STREAM_MAX_BATCH,
                "ofx_stream_launch: at most %d frames / pairs per tick", OFX_STREAM_MAX_BATCH);
    StreamArgs S{};
    size_t lds = 0;
    int stage_blocks[kPyrStages] = {0};
    int any = 0;
    // marching-pyramid waves wanted per frame: all the frames of the tick together ~0.85 per SIMD (see ofx_pyramid_march_args)
    static const int simds = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        return 4 * cus;
    }();
    const int pyr_target = g->n_pyr > 0 ? env_int("OFX_PYR_WAVES", simds * 85 / 100) / g->n_pyr : 0;
    for (int i = 0; i < g->n_pyr; ++i) {
        const ofx_pyramid_stage &P = g->pyr[i];
        if (P.levels < 2) continue;
        int items = 0;
        OFX_TRY(ofx_pyramid_march_args(P.d_frame, P.frame_pitch, P.w, P.h, P.d_levels, P.pitches, P.levels, P.d_levels[0], P.pitches[0],
                                       P.windowed ? P.row0 : nullptr, P.windowed ? P.rows : nullptr, pyr_target, &S.pyr[2 * i], &items));
        stage_blocks[2 * i] = ofx_div_up(items, 4);
        if (P.patch_levels >= 2) {
            OFX_REQUIRE(P.patch_w > 0 && P.patch_h > 0 && P.patch_w <= P.w && P.patch_h <= P.h,
                        "ofx_stream_launch: the patch must lie inside the frame");
            OFX_TRY(ofx_pyramid_march_args(P.d_frame, P.frame_pitch, P.patch_w, P.patch_h, P.d_patch_levels, P.patch_pitches, P.patch_levels,
                                           P.d_patch_levels[0], P.patch_pitches[0], nullptr, nullptr, 16, &S.pyr[2 * i + 1], &items));
            stage_blocks[2 * i + 1] = ofx_div_up(items, 4);
        }
        any += stage_blocks[2 * i] + stage_blocks[2 * i + 1];
    }
    for (int i = 0; i < g->n_corner; ++i) {
        const ofx_corner_stage &C = g->corner[i];
        OFX_REQUIRE(C.levels > 0, "ofx_stream_launch: empty corner stage");
        OFX_TRY(ofx_corner_args(C.level, C.levels, window, mode, C.d_uv, C.cols, C.d_status, &C.shard_rows[0][0], &S.corner[i]));
    }
    S.n_corner = g->n_corner;
    LkLevelIn lv[OFX_MAX_LK_ITEMS];
    int m = 0;
    if (g->n_lk > 0) OFX_TRY(lk_build_levels(g->lk, g->n_lk, window, mode, nullptr, lv, &m));
    if (any == 0 && m == 0 && g->n_corner == 0) return OFX_OK;
    hipStream_t st = ofx_stream(stream);
    if (mode == OFX_MODE_LK_FLOAT_FAST) return launch_stream_mode<OFX_MODE_LK_FLOAT, true>(window >> 1, lv, m, S, stage_blocks, lds, st);
    return mode == OFX_MODE_LK_FLOAT ? launch_stream_mode<OFX_MODE_LK_FLOAT, false>(window >> 1, lv, m, S, stage_blocks, lds, st)
                                     : launch_stream_mode<OFX_MODE_COMPAT_CPU, false>(window >> 1, lv, m, S, stage_blocks, lds, st);
}

// Debug / measurement hook (tools/stream_timeline.py): with a device buffer of 8 * capacity_blocks uint64 set, every
// wave of every later ofx_stream_launch records its start and end time there; first[] (2 * OFX_STREAM_MAX_BATCH + 1 ints) receives the block ranges of
// the last launch.  d_buf = NULL switches it off.
extern "C" int ofx_debug_stream_trace(unsigned long long *d_buf, int capacity_blocks, int *first)
{
    g_stream_trace = d_buf;
    g_stream_trace_blocks = d_buf ? capacity_blocks : 0;
    if (first)
        for (int i = 0; i <= kPyrStages; ++i) first[i] = g_trace_header[i];
    return OFX_OK;
}

extern "C" int ofx_lk_levels(const ofx_lk_desc *levels, int n, int window, int mode, void *stream)
{
    return lk_dispatch(levels, n, window, mode, nullptr, stream);
}

extern "C" int ofx_lk_level(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode,
                            float *d_flow, int flow_row0, void *stream)
{
    OFX_REQUIRE(d_flow && g, "ofx_lk_level: null argument");
    ofx_lk_desc d{d_prev, d_next, *g, d_flow, flow_row0, nullptr, 0, 0.0f};
    return lk_dispatch(&d, 1, window, mode, nullptr, stream);
}

extern "C" int ofx_lk_level_sums(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode,
                                 int32_t *d_sums5, int flow_row0, void *stream)
{
    OFX_REQUIRE(d_sums5 && g, "ofx_lk_level_sums: null argument");
    ofx_lk_desc d{d_prev, d_next, *g, nullptr, flow_row0, nullptr, 0, 0.0f};
    return lk_dispatch(&d, 1, window, mode, d_sums5, stream);
}
