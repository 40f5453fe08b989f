// Kernels, planner and launchers of the fused level kernel and of the stream kernel, as templates: included by lk_level.hip
// (entry points, argument checks) and by the lk_inst_*.hip translation units, each of which instantiates one (mode, solve)
// family -- the radii 1..12 of a family are ~90 s of hipcc on one core, six families in parallel are ~40 s.
#pragma once

#include <stdio.h>
#include <stdlib.h>

#include "corner_body.h"
#include "lk_body.h"
#include "pyr_march.h"
#include "stages_body.h"

using namespace ofx_dev;

namespace ofx_launch { // types that cross translation units
constexpr int kPyrStages = 2 * OFX_STREAM_MAX_BATCH; // per frame of the tick: its pyramid and its top-left patch pyramid
using ofx_dev::kCornerScratch;                       // LDS of a corner block: the chain's floats, then the cached corners
struct StreamArgs {
    LkTable lk;
    PyrMarchArgs pyr[kPyrStages];
    CornerHead corner[OFX_STREAM_MAX_BATCH];
    CornerLevel corner_lv[OFX_MAX_LK_ITEMS]; // the chains' levels, flat: chain i's level k at [corner[i].lv0 + k]
    // two-stage pipeline: the corner blocks build the patch pyramids their chains read (same geometry for every chain)
    PatchBuild patch;   // geometry of the patch planes (n > 0 when the corner blocks build and / or relocate them)
    int patch_build;    // the corner blocks build the patch pyramids of both frames first (two-stage pipeline)
    PatchBuildSlot patch_slot[OFX_STREAM_MAX_BATCH];
    // blocks [0, OFX_STREAM_MAX_BATCH) = one corner wave each; [.., first[0]) LK (four waves per block);
    // [first[i], first[i+1]) pyramid stage i (four marching waves per block, pyr_march.h).
    // The LK blocks come first and are planned for a whole number of waves per SIMD (lk_wave_target): they all start at
    // once and run for the whole launch, while the short staging blocks stream through the remaining slots underneath.
    int first[kPyrStages + 1];
    int n_corner;
    unsigned long long *trace; // optional (ofx_debug_stream_trace): per block, start and end time (100 MHz wall clock)
    int trace_blocks;
};
struct LkLevelIn {
    LkArgs a;     // everything but strip_h / tiles_x
    int rows_out;
};
} // namespace ofx_launch

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

// A refinement iteration of lk_iter on the buffer march (lk_body_buf.h): ITER = 1 adds to the flow, ITER = 2 also writes the warped
// image of the next iteration (lk_body_warp.h).
// ITER = 2 needs 126 VGPRs in interior tiles and 130 in the tiles at the image's left and right edge (128 for 9x9: four waves per
// SIMD, three for the other windows).  Capping it at 128 everywhere spills 2-8 registers and measured 1-3 % slower at 1080p, 4K
// and 8K (profiles/r03_ablation.txt), so the cap stays at three waves.
#ifndef OFX_ITER_MIN_WAVES
#define OFX_ITER_MIN_WAVES(ITER) 3
#endif
// DMA: the rows are fetched two steps ahead through LDS (lk_body_buf.h; chosen per launch as for the stream kernel)
// NC = 8: the march with eight columns per lane (lk_body_wide.h; no deep fetch)
#ifndef OFX_WIDE_ITER_MIN_WAVES
#define OFX_WIDE_ITER_MIN_WAVES 2
#endif
template <int R, int MODE, bool FAST, int ITER, bool DMA = false, int NC = 4>
// (with the LDS ring of the leaving rows the accumulating launches fit 128 VGPRs without scratch for R <= 7: four waves per SIMD)
__global__ __launch_bounds__(64, NC == 8 ? OFX_WIDE_ITER_MIN_WAVES : (lk_out_ring<R, ITER, DMA>() && R <= 7 ? 4 : OFX_ITER_MIN_WAVES(ITER))) void lk_iter_kernel(const LkTable T)
{
    __shared__ __attribute__((aligned(16))) uint8_t xlds[NC == 8 ? kLkWaveLdsW : (DMA ? kLkWaveLdsDma : kLkWaveLdsX + lk_ring_bytes<R, ITER, DMA>())];
    const int wave = (int)blockIdx.x, lane = (int)threadIdx.x;
    if constexpr (NC == 8) {
        static_assert(!DMA, "the wide march has no deep fetch");
        lk_wave_w<R, MODE, FAST, ITER, 8>(T, wave, lane, xlds);
        return;
    }
    if (wave >= T.first_block[T.n]) return;
    int level = 0, hi = T.n;
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (wave >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    const int tile = (wave - T.first_block[level]) % T.lv[level].tiles_x;
    const int cb0 = tile * TileGeom<R>::OUT_W - TileGeom<R>::LO_LANE * 4;
    if (cb0 >= 0 && cb0 + 256 <= T.lv[level].w) lk_wave_buf<R, MODE, FAST, true, DMA, ITER>(T, wave, lane, xlds);
    else lk_wave_buf<R, MODE, FAST, false, DMA, ITER>(T, wave, lane, xlds);
}

// ---- the stream kernel: one launch = one pipeline tick ---------------------------------------------------------------
// A tick of a frame stream runs, as disjoint block ranges of ONE grid,
//     pyramid(newest frame(s))  |  corner flows(earlier pair(s))  |  fused LK(still earlier pair(s))
// Each stage consumes what earlier launches wrote, so there is no synchronisation inside the launch and none between
// streams; the small latency-bound stages run in the shadow of the VALU-bound LK stage.  Blocks are 256 threads; an LK
// block is four independent LK waves; the corner block runs one wave per pair.
using ofx_launch::kCornerScratch;
using ofx_launch::kPyrStages;
using ofx_launch::StreamArgs;

} // namespace
namespace ofx_launch { // shared by the translation units that instantiate the kernels (defined in lk_level.hip)
extern unsigned long long *g_stream_trace; // tools/stream_timeline.py
extern int g_stream_trace_blocks;
extern int g_trace_header[2 * OFX_STREAM_MAX_BATCH + 1];
extern thread_local int g_stream_deep_fetch; // ofx_stream_stages.deep_fetch of the launch being dispatched (set by ofx_stream_launch)
} // namespace ofx_launch
namespace {
using ofx_launch::g_stream_trace;
using ofx_launch::g_stream_trace_blocks;
using ofx_launch::g_trace_header;
using ofx_launch::g_stream_deep_fetch;

// lk_float fits 5 blocks per CU (<= 96 VGPRs) without scratch for every radius; compat_cpu needs ~120: 4 blocks (<= 128)
#ifndef OFX_PYR_PRIO
#define OFX_PYR_PRIO 3 // priority of the marching-pyramid waves next to the LK waves (which go 3 -> 0 along their strips)
#endif
#ifndef OFX_STREAM_MIN_BLOCKS
#define OFX_STREAM_MIN_BLOCKS(R, MODE) ((MODE) == OFX_MODE_LK_FLOAT ? 5 : 4)
#endif
// DMA: the LK stage fetches its rows two steps ahead through LDS (lk_body_buf.h); chosen per launch by launch_stream_r
// WOUT: the LK stage is iteration 1 of pairs that have more (lk_iter): it also writes the warped images of their second iteration
// (lk_body_buf.h, ITER = WOUT = 3, or 5 on the row windows of a shard; ~128 VGPRs: three blocks per CU at least); 0: it does not
// NC = 8: the LK stage marches with eight columns per lane (lk_body_wide.h): ~170 VGPRs, OFX_WIDE_MIN_BLOCKS blocks per CU
#ifndef OFX_WIDE_MIN_BLOCKS
#define OFX_WIDE_MIN_BLOCKS 3
#endif
template <int R, int MODE, bool FAST, bool DMA, int WOUT = 0, int NC = 4>
__global__ __launch_bounds__(256, NC == 8 ? (WOUT ? 2 : OFX_WIDE_MIN_BLOCKS) : (WOUT ? 3 : OFX_STREAM_MIN_BLOCKS(R, MODE))) void stream_kernel(const StreamArgs S)
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
        if (b < S.n_corner) {
            if (S.patch_build) patch_build_block(S.patch, S.patch_slot[b], tid); // (all 256 threads; ends with a barrier)
            // the chain (wave 0) with its repair (all four waves: corner_block; without relocated planes -- S.corner[b].reloc == NULL --
            // the other waves only keep the barriers company).  ONE inlined copy of the chain per kernel: with two (corner_wave
            // next to corner_block) hipcc 7.2 fails with "illegal VGPR to SGPR copy" in the LK branch's pinned scalars.
            corner_block<MODE, FAST>(S.corner[b], S.corner_lv + S.corner[b].lv0, S.patch, tid, wv, reinterpret_cast<float *>(lds), lds + kCornerScratch,
                                     reinterpret_cast<int *>(lds + kCornerScratch - 32));
        }
    } else if (b < S.first[0]) {
        if constexpr (NC == 8) lk_wave_w<R, MODE, FAST, WOUT, 8>(S.lk, 4 * (b - OFX_STREAM_MAX_BATCH) + wv, tid & 63, lds + wv * kLkWaveLdsW);
        else lk_wave<R, MODE, false, false, FAST, DMA, WOUT>(S.lk, 4 * (b - OFX_STREAM_MAX_BATCH) + wv, tid & 63, lds + wv * (DMA ? kLkWaveLdsDma : kLkWaveLdsX));
    } else {
        int i = 0;
        while (i + 1 < kPyrStages && b >= S.first[i + 1]) ++i;
        __builtin_amdgcn_s_setprio(OFX_PYR_PRIO);
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
using ofx_launch::LkLevelIn;

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && atoi(e) > 0 ? atoi(e) : dflt;
}

// One strip height for all levels (so all waves run about equally long): the smallest that keeps the wave count within
// `capacity` (lk_wave_target), but at least `min_h` so the 2R priming rows of a strip stay a minor cost.
// The grid is sized to fit in ONE residency round: every wave runs for the whole kernel, so a second, partly filled
// round would nearly double the run time.
// Age skew (OFX_LK_SKEW="p0,p1,p2,p3", per cent): a SIMD serves its waves oldest first, and a wave's age rank on its SIMD is the
// quartile of its block index (every CU receives one block of each quartile in turn).  The strips of the items in quartile q are
// cut skew[q] per cent of the common height, so that the waves served first carry more rows.  Exact integer sums: the result
// does not depend on where the strips are cut.
inline const double *lk_skew()
{
    static double sk[4] = {1.0, 1.0, 1.0, 1.0};
    static const bool init = [] {
        const char *e = getenv("OFX_LK_SKEW");
        int v[4];
        if (e && sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4)
            for (int i = 0; i < 4; ++i) sk[i] = v[i] > 10 ? v[i] / 100.0 : 1.0;
        return true;
    }();
    (void)init;
    return sk;
}

template <int R, int NC = 4>
int plan_table(const LkLevelIn *lv, int n, int capacity, LkTable *out)
{
    using G = TileGeomW<R, NC>; // (NC = 4: TileGeom<R>)
    const int min_h = env_int("OFX_LK_MIN_STRIP", 8);
    const double *skew = lk_skew();
    const bool skewed = skew[0] != 1.0 || skew[1] != 1.0 || skew[2] != 1.0 || skew[3] != 1.0;
    int max_rows = 1;
    for (int i = 0; i < n; ++i) max_rows = lv[i].rows_out > max_rows ? lv[i].rows_out : max_rows;
    int quart[OFX_MAX_LK_ITEMS] = {0};
    auto height = [&](int i, int H) {
        int hi = skewed ? (int)(H * skew[quart[i]] + 0.5) : H;
        hi = hi < min_h ? min_h : hi;
        return hi < lv[i].rows_out ? hi : lv[i].rows_out;
    };
    auto item_waves = [&](int i, int H) { return (long)ofx_div_up(lv[i].a.w, G::OUT_W) * ofx_div_up(lv[i].rows_out, height(i, H)); };
    auto solve_h = [&]() {
        int H = min_h;
        for (; H < max_rows; ++H) {
            long waves = 0;
            for (int i = 0; i < n; ++i) waves += item_waves(i, H);
            if (waves <= (long)capacity) break;
        }
        return H;
    };
    int strip_h = solve_h();
    if (skewed) { // the quartile of an item = where the middle of its block range falls; two rounds settle it
        for (int round = 0; round < 2; ++round) {
            long total = 0, pos = 0;
            for (int i = 0; i < n; ++i) total += item_waves(i, strip_h);
            int q_new[OFX_MAX_LK_ITEMS];
            for (int i = 0; i < n; ++i) {
                const long wv = item_waves(i, strip_h);
                const int q = (int)((4 * (2 * pos + wv)) / (2 * (total > 0 ? total : 1)));
                q_new[i] = q < 0 ? 0 : (q > 3 ? 3 : q);
                pos += wv;
            }
            for (int i = 0; i < n; ++i) quart[i] = q_new[i];
            strip_h = solve_h();
        }
    }
    LkTable t{};
    t.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        t.lv[i] = lv[i].a;
        t.lv[i].tiles_x = ofx_div_up(lv[i].a.w, G::OUT_W);
        t.lv[i].strip_h = height(i, strip_h);
        t.first_block[i] = blocks;
        blocks += t.lv[i].tiles_x * ofx_div_up(lv[i].rows_out, t.lv[i].strip_h);
    }
    t.first_block[n] = blocks;
    *out = t;
    return blocks;
}

// Number of LK waves a launch is planned for.  Every LK wave runs for the whole launch, so what matters is how many of
// them share a SIMD: fewer leave issue slots empty, more shorten the strips (each strip pays its priming rows), and a count
// that is not a whole number per SIMD makes the fuller SIMDs set the time.  Measured on MI355X (one 4K pair, 9x9): with 2R
// priming steps per strip 3 per SIMD was the optimum; with the folded priming (R + 1 steps, lk_body.h) it is 4 -- 48.9 /
// 42.0 / 40.4 / 41.6 us at 2 / 3 / 4 / 5.  `reserve` slots per SIMD are left to the other
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
    static const int capacity = lk_wave_target(lk_level_kernel<R, MODE, SUMS, FAST>, 64, 0, 0, 4);
    LkTable t{};
    const int blocks = plan_table<R>(lv, n, capacity, &t);
    hipLaunchKernelGGL((lk_level_kernel<R, MODE, SUMS, FAST>), dim3((unsigned)blocks), dim3(64), 0, st, t);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

template <int R, int MODE, bool FAST, int ITER, bool DMA, int NC = 4>
int launch_iter_rd(const LkLevelIn *lv, int n, hipStream_t st)
{
    static const int capacity = lk_wave_target(lk_iter_kernel<R, MODE, FAST, ITER, DMA, NC>, 64, 0, 0, NC == 8 ? env_int("OFX_WIDE_WAVES_PER_SIMD", 2) : 4);
    LkTable t{};
    const int blocks = plan_table<R, NC>(lv, n, capacity, &t);
    hipLaunchKernelGGL((lk_iter_kernel<R, MODE, FAST, ITER, DMA, NC>), dim3((unsigned)blocks), dim3(64), 0, st, t);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

template <int R, int MODE, bool FAST, int ITER>
int launch_iter_r(const LkLevelIn *lv, int n, hipStream_t st)
{
#if OFX_LK_DMA_ROWS
    // the deep fetch, chosen as for the stream kernel (8K, 10 iterations: 15 380 vs 15 110 Mpix/s; 4K: no difference beyond the
    // +-1.5 % between runs -- profiles/r03_ablation.txt).  OFX_ITER_DMA=0 / 1 overrides.
    static const int forced = [] { const char *e = getenv("OFX_ITER_DMA"); return e ? atoi(e) : -1; }();
    long max_px = 0;
    for (int i = 0; i < n; ++i) max_px = (long)lv[i].a.w * lv[i].a.h > max_px ? (long)lv[i].a.w * lv[i].a.h : max_px;
    if (forced > 0 || (forced < 0 && max_px >= 16l * 1000 * 1000)) return launch_iter_rd<R, MODE, FAST, ITER, true>(lv, n, st);
#endif
    return launch_iter_rd<R, MODE, FAST, ITER, false>(lv, n, st);
}

// Deep fetch (DMA = true) pays where a step's row loads come from HBM -- measured on MI355X (profiles/r03_ablation.txt): 8K,
// two frames per launch: 279 vs 295 us (-5 %); 4K with its frames in the Infinity Cache: 247 vs 237 us (+4 %: the form costs
// ~60 more scalar instructions per step, and the loads are short there) -- so it is chosen by the size of the largest level:
// planes of 16 Mpx and more do not stay cached between their two uses.  OFX_LK_DMA=0 / 1 overrides.
template <int R, int MODE, bool FAST, bool DMA, int WOUT, int NC = 4>
int launch_stream_rd(const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    constexpr size_t wave_lds = NC == 8 ? kLkWaveLdsW : (DMA ? kLkWaveLdsDma : kLkWaveLdsX);
    // Next to the staging blocks the LK stage does best with 2 waves per SIMD when the tick carries one pair and 4 when it
    // carries more (measured, 4K: one pair 58.2 / 59.8 us per frame at 2 / 3; two pairs 59.7 / 57.2 / 56.5 at 2 / 3 / 4)
    // (eight columns per lane: a wave carries twice the pixels, and three blocks fit a CU: 1 / 2 waves per SIMD)
    static const int capacity1 = lk_wave_target(stream_kernel<R, MODE, FAST, DMA, WOUT, NC>, 256, 4 * wave_lds, 1, NC == 8 ? 1 : 2);
    static const int capacity2 = lk_wave_target(stream_kernel<R, MODE, FAST, DMA, WOUT, NC>, 256, 4 * wave_lds, 1, NC == 8 ? env_int("OFX_WIDE_WAVES_PER_SIMD", 2) : 4);
    int pairs = 0;
    for (int i = 0; i < n; ++i) pairs += (lv[i].a.w == lv[0].a.w && lv[i].a.h == lv[0].a.h) ? 1 : 0;
    const int capacity = pairs >= 2 ? capacity2 : capacity1;
    int lk_blocks = 0;
    if (n > 0) lk_blocks = ofx_div_up(plan_table<R, NC>(lv, n, capacity, &S.lk), 4);
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
    if (lds < 4 * wave_lds) lds = 4 * wave_lds; // an LK block: four waves, each with its exchange row (and its fetched rows)
    hipLaunchKernelGGL((stream_kernel<R, MODE, FAST, DMA, WOUT, NC>), dim3((unsigned)blocks), dim3(256), lds, st, S);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

// the same launches with eight columns per lane (one translation unit per family: lk_inst_*8.hip)
template <int MODE, bool FAST, int WOUT = 0>
int launch_stream_mode_w8(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_stream_rd<1, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 2: return launch_stream_rd<2, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 3: return launch_stream_rd<3, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 4: return launch_stream_rd<4, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 5: return launch_stream_rd<5, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 6: return launch_stream_rd<6, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 7: return launch_stream_rd<7, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 8: return launch_stream_rd<8, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 9: return launch_stream_rd<9, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 10: return launch_stream_rd<10, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    case 11: return launch_stream_rd<11, MODE, FAST, false, WOUT, 8>(lv, n, S, stage_blocks, lds, st);
    default: break;
    }
    ofx_set_error("ofx_stream_launch: window %d not supported with eight columns per lane", 2 * radius + 1);
    return OFX_E_UNSUPPORTED;
}

template <int MODE, bool FAST, int ITER>
int launch_iter_mode_w8(int radius, const LkLevelIn *lv, int n, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_iter_rd<1, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 2: return launch_iter_rd<2, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 3: return launch_iter_rd<3, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 4: return launch_iter_rd<4, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 5: return launch_iter_rd<5, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 6: return launch_iter_rd<6, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 7: return launch_iter_rd<7, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 8: return launch_iter_rd<8, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 9: return launch_iter_rd<9, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 10: return launch_iter_rd<10, MODE, FAST, ITER, false, 8>(lv, n, st);
    case 11: return launch_iter_rd<11, MODE, FAST, ITER, false, 8>(lv, n, st);
    default: break;
    }
    ofx_set_error("ofx_lk_levels: window %d not supported with eight columns per lane", 2 * radius + 1);
    return OFX_E_UNSUPPORTED;
}

template <int R, int MODE, bool FAST, int WOUT>
int launch_stream_r(const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
#if OFX_LK_BUFFER_PATH && OFX_LK_DMA_ROWS
    if constexpr (!WOUT) { // (a tick that also writes warped images measured slower with it: 8K 440 vs 416 us)
        static const int forced = [] { const char *e = getenv("OFX_LK_DMA"); return e ? atoi(e) : -1; }();
        long max_px = 0;
        for (int i = 0; i < n; ++i) max_px = (long)lv[i].a.w * lv[i].a.h > max_px ? (long)lv[i].a.w * lv[i].a.h : max_px;
        // (round 4, second session: WHERE the rows come from decides, not the level's size as such -- a ring of 4K or 1080p frames
        // longer than the Infinity Cache gains 5 % / 3.5 % with the deep fetch, a warm one loses 2 %: the caller can say which,
        // ofx_params.deep_fetch)
        const int want = forced >= 0 ? (forced > 0 ? 1 : -1) : g_stream_deep_fetch;
        if (want > 0 || (want == 0 && max_px >= 16l * 1000 * 1000)) return launch_stream_rd<R, MODE, FAST, true, WOUT>(lv, n, S, stage_blocks, lds, st);
    }
#endif
    return launch_stream_rd<R, MODE, FAST, false, WOUT>(lv, n, S, stage_blocks, lds, st);
}

template <int MODE, bool FAST, int WOUT = 0>
int launch_stream_mode(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_stream_r<1, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 2: return launch_stream_r<2, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 3: return launch_stream_r<3, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 4: return launch_stream_r<4, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 5: return launch_stream_r<5, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 6: return launch_stream_r<6, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 7: return launch_stream_r<7, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 8: return launch_stream_r<8, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 9: return launch_stream_r<9, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 10: return launch_stream_r<10, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    case 11: return launch_stream_r<11, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    default: break;
    }
    if constexpr (MODE == OFX_MODE_COMPAT_CPU) {
        if (radius == 12) return launch_stream_r<12, MODE, FAST, WOUT>(lv, n, S, stage_blocks, lds, st);
    }
    ofx_set_error("ofx_stream_launch: window %d not supported in mode %d", 2 * radius + 1, MODE);
    return OFX_E_UNSUPPORTED;
}

template <int MODE, bool FAST, int ITER>
int launch_iter_mode(int radius, const LkLevelIn *lv, int n, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_iter_r<1, MODE, FAST, ITER>(lv, n, st);
    case 2: return launch_iter_r<2, MODE, FAST, ITER>(lv, n, st);
    case 3: return launch_iter_r<3, MODE, FAST, ITER>(lv, n, st);
    case 4: return launch_iter_r<4, MODE, FAST, ITER>(lv, n, st);
    case 5: return launch_iter_r<5, MODE, FAST, ITER>(lv, n, st);
    case 6: return launch_iter_r<6, MODE, FAST, ITER>(lv, n, st);
    case 7: return launch_iter_r<7, MODE, FAST, ITER>(lv, n, st);
    case 8: return launch_iter_r<8, MODE, FAST, ITER>(lv, n, st);
    case 9: return launch_iter_r<9, MODE, FAST, ITER>(lv, n, st);
    case 10: return launch_iter_r<10, MODE, FAST, ITER>(lv, n, st);
    case 11: return launch_iter_r<11, MODE, FAST, ITER>(lv, n, st);
    default: break;
    }
    if constexpr (MODE == OFX_MODE_COMPAT_CPU) {
        if (radius == 12) return launch_iter_r<12, MODE, FAST, ITER>(lv, n, st);
    }
    ofx_set_error("ofx_lk_level: window %d not supported in mode %d", 2 * radius + 1, MODE);
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

} // namespace

// One external function per family (defined in lk_inst_*.hip): all levels of a fused launch / one stream tick.
namespace ofx_launch {
int levels_lk_float(int radius, const LkLevelIn *lv, int n, bool sums, hipStream_t st);
int levels_lk_float_fast(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int levels_compat_cpu(int radius, const LkLevelIn *lv, int n, bool sums, hipStream_t st);
// refinement iterations on the buffer march (lk_wave_buf's ITER): 1 flow += result; 2 the launch also writes the next iteration's
// warped images; 3 iteration 1 of pairs that have more: flow = result and the warped images of iteration 2
// (one translation unit per ITER: lk_inst_iter_*.hip)
int iter1_lk_float(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int iter2_lk_float(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int iter3_lk_float(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int iter1_lk_float_fast(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int iter2_lk_float_fast(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int iter3_lk_float_fast(int radius, const LkLevelIn *lv, int n, hipStream_t st);
int iter4_lk_float(int radius, const LkLevelIn *lv, int n, hipStream_t st);      // (2 on the row windows of a shard)
int iter4_lk_float_fast(int radius, const LkLevelIn *lv, int n, hipStream_t st);
// a tick whose LK stage also writes the warped images of its pairs' second iteration
int stream_lk_float_wout(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int stream_lk_float_fast_wout(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int stream_lk_float_wout_rw(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);      // (row windows)
int stream_lk_float_fast_wout_rw(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int stream_lk_float(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int stream_lk_float_fast(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int stream_compat_cpu(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
// eight columns per lane (lk_body_wide.h)
int stream_lk_float_w8(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int stream_lk_float_fast_w8(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st);
int levels_lk_float_w8(int radius, const LkLevelIn *lv, int n, hipStream_t st); // all levels of one pair (the pair-at-a-time path)
} // namespace ofx_launch
