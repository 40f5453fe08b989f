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
#include <string.h>

#include "lk_launch.h"


namespace ofx_launch {
unsigned long long *g_stream_trace = nullptr; // tools/stream_timeline.py
int g_stream_trace_blocks = 0;
int g_trace_header[2 * OFX_STREAM_MAX_BATCH + 1] = {0};
thread_local int g_stream_deep_fetch = 0;
} // namespace ofx_launch

namespace {

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
        OFX_REQUIRE((d[i].d_warp_out != nullptr) == (d[0].d_warp_out != nullptr) && (d[i].accumulate != 0) == (d[0].accumulate != 0),
                    "ofx_lk_levels: accumulate and d_warp_out must be set for all descriptors of a launch or for none");
        if (d[i].d_warp_out) { // the launch also writes the next iteration's warped image (lk_body_warp.h)
            OFX_REQUIRE(!d_sums && d[i].d_warp_src && d[i].d_flow, "ofx_lk_levels: d_warp_out needs d_warp_src and d_flow");
            OFX_REQUIRE(d[i].d_warp_out != d[i].d_next && d[i].d_warp_out != d[i].d_warp_src && d[i].d_warp_out != d[i].d_prev,
                        "ofx_lk_levels: d_warp_out must be a plane of its own");
            a.warp_src = d[i].d_warp_src;
            a.warp_out = d[i].d_warp_out;
            a.warp_scale = d[i].warp_scale;
            a.warp_status = d[i].d_warp_status;
            a.warp_status_bit = d[i].warp_status_bit;
            OFX_REQUIRE(a.warp_status == nullptr || (a.warp_status_bit >= 0 && a.warp_status_bit < 31), "ofx_lk_levels: bad warp_status_bit");
        }
        lv[m].a = a;
        lv[m].rows_out = rows_out;
        ++m;
    }
    *count = m;
    return OFX_OK;
}

// Columns per lane of the LK march of a launch: 8 (lk_body_wide.h) or 4.  The wide march has no deep fetch, so launches that would
// choose that (levels of 16 Mpx and more, launch_stream_r) keep four columns.  OFX_LK_COLS=4 / 8 overrides.
#ifndef OFX_LK_COLS_DEFAULT
#define OFX_LK_COLS_DEFAULT 4 // (measured, profiles/r04_ablation.txt: eight columns lose at 4K -- two LK waves per SIMD next to the pyramid stage)
#endif
int lk_cols(const LkLevelIn *lv, int m, int radius)
{
    static const int forced = [] { const char *e = getenv("OFX_LK_COLS"); return e ? atoi(e) : 0; }();
    static const int dma_forced = [] { const char *e = getenv("OFX_LK_DMA"); return e ? atoi(e) : -1; }();
    if (radius < 1 || radius > 11 || m <= 0) return 4;
    if (dma_forced < 0 && ofx_launch::g_stream_deep_fetch > 0) return 4;
    if (forced == 4 || forced == 8) return forced;
    long max_px = 0;
    for (int i = 0; i < m; ++i) max_px = (long)lv[i].a.w * lv[i].a.h > max_px ? (long)lv[i].a.w * lv[i].a.h : max_px;
    if (dma_forced > 0 || (dma_forced < 0 && max_px >= 16l * 1000 * 1000)) return 4;
    return OFX_LK_COLS_DEFAULT;
}

int lk_dispatch(const ofx_lk_desc *d, int n, int window, int mode, int32_t *d_sums, void *stream)
{
    LkLevelIn lv[OFX_MAX_LK_ITEMS];
    int m = 0;
    OFX_TRY(lk_build_levels(d, n, window, mode, d_sums, lv, &m));
    if (m == 0) return OFX_OK;
    const int radius = window >> 1;
    hipStream_t st = ofx_stream(stream);
    if ((lv[0].a.accumulate || lv[0].a.warp_out) && !d_sums) {
        // refinement iterations run on the buffer march (32-bit offsets: levels below 2 GB; larger ones keep the old form, which
        // cannot write the warped image)
        bool small = true;
        for (int i = 0; i < m; ++i)
            small = small && (size_t)(lv[i].a.row_end - lv[i].a.row0) * (size_t)lv[i].a.pitch < ((size_t)1 << 31) &&
                    (size_t)(lv[i].a.out_y1 - lv[i].a.flow_row0) * (size_t)lv[i].a.w * 8 < ((size_t)1 << 31);
        const bool wout = lv[0].a.warp_out != nullptr;
        static const bool old_form = [] { const char *e = getenv("OFX_ITER_OLD_MARCH"); return e && atoi(e) != 0; }();
        OFX_REQUIRE(!wout || (small && mode != OFX_MODE_COMPAT_CPU), "ofx_lk_levels: d_warp_out needs mode lk_float and levels below 2 GB");
        bool rowwin = false; // a shard's row window: the warp reports taps it cannot reach (ITER 4 / 5)
        for (int i = 0; i < m; ++i) rowwin = rowwin || lv[i].a.row0 != 0 || lv[i].a.row_end != lv[i].a.h;
        OFX_REQUIRE(!(wout && rowwin && !lv[0].a.accumulate), "ofx_lk_levels: iteration 1 with d_warp_out on a row window runs in the stream tick only");
        const int iter = wout ? (lv[0].a.accumulate ? (rowwin ? 4 : 2) : 3) : 1; // (lk_wave_buf's ITER)
        if (small && (wout || !old_form) && mode != OFX_MODE_COMPAT_CPU) // (compat_cpu accumulates in the old form below)
        {
            using F = int (*)(int, const LkLevelIn *, int, hipStream_t);
            static const F tab[2][4] = {{ofx_launch::iter1_lk_float, ofx_launch::iter2_lk_float, ofx_launch::iter3_lk_float, ofx_launch::iter4_lk_float},
                                        {ofx_launch::iter1_lk_float_fast, ofx_launch::iter2_lk_float_fast, ofx_launch::iter3_lk_float_fast,
                                         ofx_launch::iter4_lk_float_fast}};
            return tab[mode == OFX_MODE_LK_FLOAT_FAST][iter - 1](radius, lv, m, st);
        }
    }
    // (experiment, OFX_LK_PLAIN_COLS=8: the pair-at-a-time launch on the march with eight columns per lane)
    static const bool plain_wide = [] { const char *e = getenv("OFX_LK_PLAIN_COLS"); return e && atoi(e) == 8; }();
    if (plain_wide && !d_sums && mode == OFX_MODE_LK_FLOAT && !lv[0].a.accumulate && !lv[0].a.warp_out && radius >= 1 && radius <= 11) {
        bool small = true;
        for (int i = 0; i < m; ++i)
            small = small && (size_t)(lv[i].a.row_end - lv[i].a.row0) * (size_t)lv[i].a.pitch < ((size_t)1 << 31) &&
                    (size_t)(lv[i].a.out_y1 - lv[i].a.flow_row0) * (size_t)lv[i].a.w * 8 < ((size_t)1 << 31);
        if (small) return ofx_launch::levels_lk_float_w8(radius, lv, m, st);
    }
    if (d_sums) // the sums do not depend on the solve
        return mode != OFX_MODE_COMPAT_CPU ? ofx_launch::levels_lk_float(radius, lv, m, true, st) : ofx_launch::levels_compat_cpu(radius, lv, m, true, st);
    if (mode == OFX_MODE_LK_FLOAT_FAST) return ofx_launch::levels_lk_float_fast(radius, lv, m, st);
    return mode == OFX_MODE_LK_FLOAT ? ofx_launch::levels_lk_float(radius, lv, m, false, st) : ofx_launch::levels_compat_cpu(radius, lv, m, false, st);
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
    int n_clv = 0, n_build = 0;
    for (int i = 0; i < g->n_corner; ++i) {
        const ofx_corner_stage &C = g->corner[i];
        OFX_REQUIRE(C.levels > 0, "ofx_stream_launch: empty corner stage");
        OFX_REQUIRE(n_clv + C.levels <= OFX_MAX_LK_ITEMS, "ofx_stream_launch: the corner stages have more than %d (pair, level) items", OFX_MAX_LK_ITEMS);
        OFX_TRY(ofx_corner_args(C.level, C.levels, window, mode, C.d_uv, C.cols, C.d_status, &C.shard_rows[0][0], &S.corner[i], S.corner_lv + n_clv));
        S.corner[i].lv0 = n_clv;
        n_clv += C.levels;
        const bool reloc = C.levels >= 2 && C.d_patch_reloc[1] != nullptr;
        S.corner[i].pair_status = C.d_pair_status;
        if (C.build_patch || reloc) {
            // the chain's patch pyramids are built (and, when a shift leaves them, rebuilt elsewhere) by its own block: one
            // geometry for all the chains of a launch
            OFX_REQUIRE(C.levels >= 2 && C.patch_w > 0 && C.patch_h > 0, "ofx_stream_launch: corner stage %d: incomplete patch description", i);
            OFX_REQUIRE(!C.build_patch || (C.d_patch_src[0] && C.d_patch_src[1] && C.d_patch[0][1] && C.d_patch[1][1]),
                        "ofx_stream_launch: corner stage %d: incomplete patch description", i);
            uint8_t *const *planes = C.build_patch ? C.d_patch[0] : C.d_patch_reloc; // (the set whose layout defines the offsets)
            PatchBuild pb{};
            pb.n = C.levels - 1;
            pb.frame_stride = C.build_patch ? (int)(C.d_patch[1][1] - C.d_patch[0][1]) : 0;
            for (int k = 0; k < C.levels; ++k) {
                pb.pw[k] = C.patch_w >> k;
                pb.ph[k] = C.patch_h >> k;
                pb.pitch[k] = k ? C.patch_pitch[k] : 0;
                pb.off[k] = k ? (int)(planes[k] - planes[1]) : 0;
                OFX_REQUIRE(pb.pw[k] > 0 && pb.ph[k] > 0, "ofx_stream_launch: the patch is too small for %d levels", C.levels);
                if (k) {
                    OFX_REQUIRE(planes[k] != nullptr && (C.patch_pitch[k] & 3) == 0 && C.patch_pitch[k] >= ((pb.pw[k] + 3) & ~3) && ((uintptr_t)planes[k] & 3) == 0,
                                "ofx_stream_launch: corner stage %d: bad patch plane at level %d", i, k);
                    OFX_REQUIRE(!C.build_patch || C.d_patch[1][k] == C.d_patch[0][k] + pb.frame_stride,
                                "ofx_stream_launch: corner stage %d: bad patch plane at level %d", i, k);
                    OFX_REQUIRE(!reloc || (C.d_patch_reloc[k] != nullptr && C.d_patch_reloc[k] - C.d_patch_reloc[1] == pb.off[k] && ((uintptr_t)C.d_patch_reloc[k] & 3) == 0),
                                "ofx_stream_launch: corner stage %d: the relocated patch planes must be laid out like the patch planes (level %d)", i, k);
                    OFX_REQUIRE(((C.patch_w >> (k - 1)) & 1) == 0 && ((C.patch_h >> (k - 1)) & 1) == 0, "ofx_stream_launch: the patch must have even dimensions below its top level");
                }
            }
            if (C.build_patch)
                for (int f = 0; f < 2; ++f)
                    OFX_REQUIRE((C.patch_src_pitch[f] & 3) == 0 && C.patch_src_pitch[f] >= C.patch_w && ((uintptr_t)C.d_patch_src[f] & 3) == 0,
                                "ofx_stream_launch: corner stage %d: bad patch source", i);
            if (reloc) {
                // the relocated build reads the whole next frame through level 0's descriptor, and the chain must be able to
                // place the patch around any target: whole level-0 planes, a patch inside the frame, room for the window at
                // the coarsest level (its stencils span radius + 3 pixels; one more for the plane's first column / row)
                const ofx_geom &g0 = C.level[0].geom;
                OFX_REQUIRE(g0.row0 == 0 && g0.rows == g0.h && (C.cols[0] == 0 || C.cols[0] >= g0.w),
                            "ofx_stream_launch: corner stage %d: the repair needs level 0 to be the whole frames", i);
                OFX_REQUIRE(C.patch_w <= g0.w && C.patch_h <= g0.h, "ofx_stream_launch: corner stage %d: the patch must lie inside the frame", i);
                const int lc = C.levels - 1, need = (window >> 1) + 5;
                OFX_REQUIRE((pb.pw[lc] >= need || pb.pw[lc] >= (g0.w >> lc)) && (pb.ph[lc] >= need || pb.ph[lc] >= (g0.h >> lc)),
                            "ofx_stream_launch: corner stage %d: a %dx%d patch leaves %dx%d at the coarsest level, the repair needs %d", i, C.patch_w,
                            C.patch_h, pb.pw[lc], pb.ph[lc], need);
                S.corner[i].reloc = C.d_patch_reloc[1];
            }
            if (S.patch.n == 0) S.patch = pb;
            else OFX_REQUIRE(memcmp(&S.patch, &pb, sizeof pb) == 0, "ofx_stream_launch: the corner stages of a launch must share one patch geometry");
            if (C.build_patch) {
                S.patch_slot[i] = PatchBuildSlot{{C.d_patch_src[0], C.d_patch_src[1]}, {C.patch_src_pitch[0], C.patch_src_pitch[1]}, C.d_patch[0][1]};
                ++n_build;
            }
        }
    }
    OFX_REQUIRE(n_build == 0 || n_build == g->n_corner, "ofx_stream_launch: either every corner stage builds its patch or none does");
    S.patch_build = n_build > 0 ? 1 : 0;
    S.n_corner = g->n_corner;
    LkLevelIn lv[OFX_MAX_LK_ITEMS];
    int m = 0;
    if (g->n_lk > 0) {
        OFX_REQUIRE(!g->lk[0].accumulate, "ofx_stream_launch: a tick's LK stage is iteration 1 (no accumulate)");
        OFX_REQUIRE(g->lk[0].d_warp_out == nullptr || mode != OFX_MODE_COMPAT_CPU, "ofx_stream_launch: d_warp_out needs mode lk_float");
        OFX_TRY(lk_build_levels(g->lk, g->n_lk, window, mode, nullptr, lv, &m));
    }
    // the stream kernel's LK stage addresses planes and flow through buffer resources with 32-bit offsets (lk_body_buf.h)
    for (int i = 0; i < m; ++i) {
        const LkArgs &a = lv[i].a;
        OFX_REQUIRE((long long)(a.row_end - a.row0) * a.pitch < (1ll << 31) && (long long)(a.out_y1 - a.flow_row0) * a.w * 8 < (1ll << 31),
                    "ofx_stream_launch: level %dx%d is too large for one launch item (planes and flow rows must stay below 2 GB: "
                    "shard the level by rows)", a.w, a.h);
    }
    if (any == 0 && m == 0 && g->n_corner == 0) return OFX_OK;
    hipStream_t st = ofx_stream(stream);
    OFX_REQUIRE(g->deep_fetch >= -1 && g->deep_fetch <= 1, "ofx_stream_launch: deep_fetch must be -1, 0 or +1 (got %d)", g->deep_fetch);
    ofx_launch::g_stream_deep_fetch = g->deep_fetch;
    if (m > 0 && lv[0].a.warp_out) { // the LK stage also writes the warped images of its pairs' second iteration (levels below 2 GB: checked above)
        bool rw = false; // a shard's row windows
        for (int i = 0; i < m; ++i) rw = rw || lv[i].a.row0 != 0 || lv[i].a.row_end != lv[i].a.h;
        if (rw)
            return mode == OFX_MODE_LK_FLOAT_FAST ? ofx_launch::stream_lk_float_fast_wout_rw(window >> 1, lv, m, S, stage_blocks, lds, st)
                                                  : ofx_launch::stream_lk_float_wout_rw(window >> 1, lv, m, S, stage_blocks, lds, st);
        return mode == OFX_MODE_LK_FLOAT_FAST ? ofx_launch::stream_lk_float_fast_wout(window >> 1, lv, m, S, stage_blocks, lds, st)
                                              : ofx_launch::stream_lk_float_wout(window >> 1, lv, m, S, stage_blocks, lds, st);
    }
    if (mode != OFX_MODE_COMPAT_CPU && lk_cols(lv, m, window >> 1) == 8)
        return mode == OFX_MODE_LK_FLOAT_FAST ? ofx_launch::stream_lk_float_fast_w8(window >> 1, lv, m, S, stage_blocks, lds, st)
                                              : ofx_launch::stream_lk_float_w8(window >> 1, lv, m, S, stage_blocks, lds, st);
    if (mode == OFX_MODE_LK_FLOAT_FAST) return ofx_launch::stream_lk_float_fast(window >> 1, lv, m, S, stage_blocks, lds, st);
    return mode == OFX_MODE_LK_FLOAT ? ofx_launch::stream_lk_float(window >> 1, lv, m, S, stage_blocks, lds, st)
                                     : ofx_launch::stream_compat_cpu(window >> 1, lv, m, S, stage_blocks, lds, st);
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
