// Device-resident session: the frame loop of main.cu:192-272 with every buffer living in HBM.
//
// One hipMalloc arena holds, per pyramid level: the previous and the next frame's 1-channel planes, a scratch
// plane for the shifted next frame, the flow field, and the 2-float shift vector.  Nothing is allocated or freed
// while frames flow (the reference does 58 cudaMalloc/cudaFree calls per level).
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "compat_scratch.h"
#include "ofx_internal.h"

namespace {
constexpr size_t kAlign = 256;
// The stream pipeline with B frames per tick uses 3B + 2 image sets and 2B shift-vector slots (see stream_tick); the
// pair-at-a-time paths rotate 3 sets and alternate 2 slots.
constexpr int kMaxBatch = OFX_STREAM_MAX_BATCH;
constexpr int kSets = 3 * kMaxBatch + 2;
constexpr int kUvSlots = 2 * kMaxBatch;
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
} // namespace

struct ofx_session {
    ofx_params p{};
    int w[OFX_MAX_LEVELS]{}, h[OFX_MAX_LEVELS]{}, pitch[OFX_MAX_LEVELS]{};
    int own0[OFX_MAX_LEVELS]{}, own1[OFX_MAX_LEVELS]{}; // rows this rank computes
    int buf0[OFX_MAX_LEVELS]{}, buf1[OFX_MAX_LEVELS]{}; // rows the plane buffers hold
    int cmp0[OFX_MAX_LEVELS]{}, cmp1[OFX_MAX_LEVELS]{}; // rows this rank downsamples itself
    // rows the flow buffers hold: the own rows, or -- sharded sessions with refinement iterations -- the own rows plus
    // (radius + 1) * (iters - 1) either side: iteration j is computed on (radius + 1) * (iters - j) extra rows so that the
    // warp of iteration j + 1 finds the flow of every row its LK stencils touch without asking a neighbour (stream_tick)
    int fl0[OFX_MAX_LEVELS]{}, fl1[OFX_MAX_LEVELS]{};
    size_t flow_own_offset(int k) const { return (size_t)(own0[k] - fl0[k]) * (size_t)w[k] * 2; } // floats from a flow set to the own rows
    // storage: three image sets rotate through the roles prev -> (free) -> next, two shifted-scratch sets alternate, so
    // that the pipelined path can build frame i+1's pyramid / corner / shift while pair i's LK launch is running
    uint8_t *img[kSets][OFX_MAX_LEVELS]{};              // see kSets
    uint8_t *sh[2][OFX_MAX_LEVELS]{};
    // refinement iterations in the stream pipeline: per flow set (pair p -> set p mod B) the shifted and the warped next image
    uint8_t *itsh[kMaxBatch][3][OFX_MAX_LEVELS]{};
    bool fused_iters = false; // the accumulating launches also write the next iteration's warped image (lk_body_warp.h)
    int cur = 0, sht = 0;                               // img[cur] = previous frame, img[(cur+1)%3] = next frame
    uint8_t *plane[3][OFX_MAX_LEVELS]{};                // role view: 0 prev, 1 next, 2 shifted scratch
    hipStream_t aux = nullptr;                          // pipelined path: staging stream owned by the session
    hipEvent_t ev_ready = nullptr;                      // staging of the next pair finished (aux -> main)
    hipEvent_t ev_set_done[3] = {nullptr, nullptr, nullptr}; // last LK launch that read img[i] as `prev` finished
    bool set_busy[3] = {false, false, false};
    bool staged = false;
    int uv_slot = 0; // shift-vector slot of the pair in progress; alternates per pair so that the staging of the next
                     // pair (aux stream) never overwrites vectors the running LK launch still reads
    float *uv_cur() { return uv + (size_t)uv_slot * 2 * OFX_MAX_LEVELS; }
    long stream_n = -1;      // ticks of the stream pipeline so far (-1: not streaming)
    long stream_frames = -1; // total frames, known once draining starts (-1: still receiving)
    int pitch0_next() const { return pitch[0]; }
    // local_corner: the top-left patch of every frame as a pyramid of its own (same 5 sets as img)
    uint8_t *pimg[kSets][OFX_MAX_LEVELS]{};
    int pw[OFX_MAX_LEVELS]{}, ph[OFX_MAX_LEVELS]{}, ppitch[OFX_MAX_LEVELS]{};
    // stream_two_stage: the patch planes the corner block of slot i builds for its pair (frame 0: previous, 1: next)
    uint8_t *pscr[kMaxBatch][2][OFX_MAX_LEVELS]{};
    // the repair of a shift that leaves the patch (ofx_corner_stage.d_patch_reloc): per corner slot one more set of patch planes,
    // for the next frame's pyramid rebuilt around the shifted corner.  Allocated where the whole frames stay at hand
    // (borrow_frames) and the chain reads a patch (stream_two_stage, local_corner).
    uint8_t *preloc[kMaxBatch][OFX_MAX_LEVELS]{};
    bool repair = false;
    // pair-at-a-time sessions (neither local_corner nor stream_two_stage, whole frames): ofx_session_build_pyramid also walks the
    // pair's corner chain in one more block of its launch (pyr_corner.hip) on patch planes that block builds: pscr[0][0 / 1] hold the
    // patch pyramids of two image sets in turn (pset_img / pset_gen: which set's, and of which load), preloc[0] the repair's planes
    bool plain_fuse = false;
    bool corner_done = false; // the shift vectors of the pair (prev, next) are in uv_cur() already
    int pset_img[2] = {-1, -1};
    long pset_gen[2] = {0, 0}, img_gen[3] = {0, 0, 0};
    int debug_extent = 0; // test hook (OFX_DEBUG_CORNER_EXTENT): the chain may only read this many level-0 columns / rows of its patch planes
    int *corner_status = nullptr;
    int *pair_status = nullptr; // one word per shift-vector slot (pair p -> slot p mod 2B)
    const uint8_t *pframe[3] = {nullptr, nullptr, nullptr}; // borrow_frames, pair-at-a-time: the caller's frame behind img[i]'s level 0
    float *flow[OFX_MAX_LEVELS]{};       // where results are read from: flowset[0], or the newest pair's set in a two-frame stream
    float *flowset[kMaxBatch][OFX_MAX_LEVELS]{}; // stream pipeline: pair p's flow goes to set p mod stream_batch
    const uint8_t *held_frame[kMaxBatch]{};      // multi-frame stream tick: the frames waiting for the tick to fill
    int held_pitch[kMaxBatch]{};
    int n_held = 0;
    const uint8_t *bframe[kSets]{}; // borrow_frames: the caller's buffer behind image set i (level 0 is read from there)
    int bpitch[kSets]{};
    long reported = 0;                   // highest pair reported complete by the stream pipeline
    long corner_newest = 0;              // highest pair whose corner stage has been enqueued (ofx_session_pair_status)
    float *uv = nullptr;        // 2 floats per level
    uint8_t *staging = nullptr; // one tightly packed 3ch level-0 frame for host uploads
    void *arena = nullptr;
    size_t arena_bytes = 0;
    bool have_next = false, have_prev = false;
    // optional timing of the level-0 fused LK launch: event pairs recorded on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind; // OFX_TIME_* of event pair i
    size_t ev_used = 0;
    hipEvent_t ev_frame = nullptr; // staged path: the caller's frame is complete (caller's stream -> aux)
    int n_sets = 0;                // image sets allocated (3B + 2; the pair-at-a-time paths rotate the first three)
};

// Runs `launch` bracketed by a pair of timing events of kind `kind` when the session is armed (ofx_session_timing).
template <typename F>
static int timed_launch(ofx_session *s, int kind, void *stream, F &&launch)
{
    static const char *const names[OFX_TIME_KINDS] = {"ofx.lk_levels", "ofx.lk_levels_accumulate", "ofx.warp_levels", "ofx.stream_tick",
                                                      "ofx.shift_levels", "ofx.corner_flows", "ofx.pyramid", "ofx.lk_levels_accumulate_warp"};
    OfxRange range(names[kind]); // (roctx, OFX_ROCTX=1: the launch's enqueue on the host side of a --marker-trace timeline)
    const bool timed = s->timing && s->ev_used + 2 <= s->ev.size();
    if (timed) OFX_HIP(hipEventRecord(s->ev[s->ev_used], ofx_stream(stream)));
    OFX_TRY(launch());
    if (timed) {
        OFX_HIP(hipEventRecord(s->ev[s->ev_used + 1], ofx_stream(stream)));
        s->ev_kind[s->ev_used / 2] = kind;
        s->ev_used += 2;
    }
    return OFX_OK;
}

static void repoint(ofx_session *s)
{
    for (int k = 0; k < s->p.levels; ++k) {
        s->plane[0][k] = s->img[s->cur][k];
        s->plane[1][k] = s->img[(s->cur + 1) % 3][k];
        s->plane[2][k] = s->sh[s->sht][k];
    }
    // borrowed frames: level 0 is the caller's buffer (same pitch as the session's plane; only ever read)
    const size_t skip = (size_t)s->buf0[0] * (size_t)s->pitch[0];
    if (s->pframe[s->cur]) s->plane[0][0] = const_cast<uint8_t *>(s->pframe[s->cur]) + skip;
    if (s->pframe[(s->cur + 1) % 3]) s->plane[1][0] = const_cast<uint8_t *>(s->pframe[(s->cur + 1) % 3]) + skip;
}

static ofx_geom level_geom(const ofx_session *s, int k, int out0, int out1)
{
    ofx_geom g;
    g.w = s->w[k];
    g.h = s->h[k];
    g.pitch = s->pitch[k];
    g.row0 = s->buf0[k];
    g.rows = s->buf1[k] - s->buf0[k];
    g.out_y0 = out0;
    g.out_y1 = out1;
    return g;
}

extern "C" int ofx_session_create(const ofx_params *p_in, ofx_session **out)
{
    OFX_REQUIRE(p_in && out, "ofx_session_create: null argument");
    // The fused warp of a refinement iteration fetches a tap's dword AT the tap's byte (lk_body_warp.h): its source must be followed by
    // three readable bytes.  Every plane of a session is (by 64); the one warp source that would be a caller's buffer is level 0 of
    // a single-level session on borrowed frames (pair at a time: the stream pipeline needs two levels) -- such a session copies its frames.
    ofx_params eff = *p_in;
    if (eff.levels == 1 && eff.iters > 1 && eff.borrow_frames) eff.borrow_frames = 0, eff.stream_two_stage = 0;
    const ofx_params *p = &eff;
    OFX_REQUIRE(p->width > 0 && p->height > 0, "ofx_session_create: bad size %dx%d", p->width, p->height);
    OFX_REQUIRE(p->levels >= 1 && p->levels <= OFX_MAX_LEVELS, "ofx_session_create: levels %d out of range", p->levels);
    OFX_REQUIRE(p->window >= 3 && (p->window & 1), "ofx_session_create: window must be odd and >= 3");
    OFX_REQUIRE(p->mode == OFX_MODE_COMPAT_CPU || p->mode == OFX_MODE_LK_FLOAT || p->mode == OFX_MODE_LK_FLOAT_FAST,
                "ofx_session_create: bad mode %d", p->mode);
    OFX_REQUIRE(p->min_det >= 0.0f, "ofx_session_create: min_det must be >= 0 (0 = the reference's unguarded solve)");
    OFX_REQUIRE(p->iters >= 0 && p->iters <= 64, "ofx_session_create: iters %d out of range", p->iters);
    OFX_REQUIRE(p->stream_batch >= 0 && p->stream_batch <= OFX_STREAM_MAX_BATCH, "ofx_session_create: stream_batch %d (0 .. %d)", p->stream_batch,
                OFX_STREAM_MAX_BATCH);
    OFX_REQUIRE(p->stream_batch * p->levels <= OFX_MAX_LK_ITEMS, "ofx_session_create: stream_batch %d needs levels <= %d",
                p->stream_batch, OFX_MAX_LK_ITEMS / (p->stream_batch > 0 ? p->stream_batch : 1));
    OFX_REQUIRE(!p->stream_two_stage || p->borrow_frames, "ofx_session_create: stream_two_stage needs borrow_frames (the corner stage reads "
                                                            "both frames of a pair in the tick in which the second one arrives)");
    OFX_REQUIRE(p->iters <= 1 || p->mode != OFX_MODE_COMPAT_CPU, "ofx_session_create: refinement iterations need mode lk_float");
    OFX_REQUIRE(p->deep_fetch >= -1 && p->deep_fetch <= 1, "ofx_session_create: deep_fetch must be -1, 0 or +1 (got %d)", p->deep_fetch);
    OFX_REQUIRE(!p->frames_partial || (p->sharded && p->local_corner && !p->stream_two_stage),
                "ofx_session_create: frames_partial describes the frames of a sharded local_corner session (not stream_two_stage)");
    OFX_REQUIRE(p->iters <= 1 || !p->sharded || p->local_corner,
                "ofx_session_create: refinement iterations on a sharded session run through the stream pipeline, which needs local_corner");
    OFX_REQUIRE((p->width >> (p->levels - 1)) > 0 && (p->height >> (p->levels - 1)) > 0,
                "ofx_session_create: %d levels is too many for %dx%d", p->levels, p->width, p->height);
    for (int k = 0; k + 1 < p->levels; ++k)
        OFX_REQUIRE(((p->width >> k) & 1) == 0 && ((p->height >> k) & 1) == 0,
                    "ofx_session_create: level %d is %dx%d; every level that is downsampled must have even dimensions "
                    "(the reference assumes a source stride of exactly 2*w, OptFlowCPU.cpp:117)",
                    k, p->width >> k, p->height >> k);
    OFX_HIP(hipSetDevice(p->device));

    ofx_session *s = new (std::nothrow) ofx_session();
    OFX_REQUIRE(s != nullptr, "ofx_session_create: out of host memory");
    s->p = *p;
    // image sets: the stream pipeline cycles through 3B + 2 of them (B = frames per tick; a session created without a
    // stream_batch may still stream one frame per tick), the pair-at-a-time paths rotate the first three
    const int n_sets = (p->stream_two_stage ? 2 : 3) * (p->stream_batch >= 2 ? p->stream_batch : 1) + 2;
    s->n_sets = n_sets;
    size_t total = 0;
    // (streamed refinement iterations: two more scratch planes per pair of a tick)
    const int n_iter_sets = p->iters > 1 ? 3 * (p->stream_batch >= 2 ? p->stream_batch : 1) : 0; // per flow set: shifted, warped, warped'
    std::vector<size_t> off_plane[kSets + 2 + 3 * kMaxBatch], off_flow, off_flow2, flow_stride;
    for (int k = 0; k < p->levels; ++k) {
        s->w[k] = p->width >> k;
        s->h[k] = p->height >> k;
        s->pitch[k] = (int)align_up((size_t)s->w[k], 64);
        if (p->sharded) {
            s->own0[k] = p->own_y0[k];
            s->own1[k] = p->own_y1[k];
            s->buf0[k] = p->buf_y0[k];
            s->buf1[k] = p->buf_y1[k];
            const bool has_comp = p->comp_y1[k] > 0;
            s->cmp0[k] = has_comp ? p->comp_y0[k] : s->own0[k];
            s->cmp1[k] = has_comp ? p->comp_y1[k] : s->own1[k];
            const bool ok = 0 <= s->buf0[k] && s->buf0[k] <= s->own0[k] && s->own0[k] <= s->own1[k] &&
                            s->own1[k] <= s->buf1[k] && s->buf1[k] <= s->h[k] && s->buf0[k] < s->buf1[k] &&
                            s->buf0[k] <= s->cmp0[k] && s->cmp0[k] <= s->own0[k] && s->own1[k] <= s->cmp1[k] &&
                            s->cmp1[k] <= s->buf1[k];
            if (!ok) {
                ofx_set_error("ofx_session_create: level %d shard rows own [%d,%d) buf [%d,%d) invalid for height %d", k,
                              s->own0[k], s->own1[k], s->buf0[k], s->buf1[k], s->h[k]);
                delete s;
                return OFX_E_INVALID;
            }
        } else {
            s->own0[k] = s->buf0[k] = s->cmp0[k] = 0;
            s->own1[k] = s->buf1[k] = s->cmp1[k] = s->h[k];
        }
        s->fl0[k] = s->own0[k];
        s->fl1[k] = s->own1[k];
        if (p->sharded && p->iters > 1) {
            const int reach = (p->window >> 1) + 1, ext = reach * (p->iters - 1);
            s->fl0[k] = s->own0[k] - ext > 0 ? s->own0[k] - ext : 0;
            s->fl1[k] = s->own1[k] + ext < s->h[k] ? s->own1[k] + ext : s->h[k];
            const int lo = s->fl0[k] - reach > 0 ? s->fl0[k] - reach : 0, hi = s->fl1[k] + reach < s->h[k] ? s->fl1[k] + reach : s->h[k];
            if (lo < s->buf0[k] || hi > s->buf1[k]) {
                ofx_set_error("ofx_session_create: level %d: %d iterations need rows [%d,%d) in the buffers (own rows +- (radius + 1) * iters), "
                              "they hold [%d,%d): plan the shard with its iterations (ShardPlan(iters=...))", k, p->iters, lo, hi, s->buf0[k], s->buf1[k]);
                delete s;
                return OFX_E_INVALID;
            }
        }
        const size_t plane_bytes = align_up((size_t)s->pitch[k] * (size_t)(s->buf1[k] - s->buf0[k]) + 64, kAlign);
        for (int t = 0; t < n_sets + 2 + n_iter_sets; ++t) { // image sets + 2 shifted sets + the streamed iterations' scratch
            off_plane[t].push_back(total);
            total += plane_bytes;
        }
        off_flow.push_back(total);
        const size_t own_rows = (size_t)(s->fl1[k] - s->fl0[k]);
        const size_t flow_bytes = align_up((own_rows ? own_rows : 1) * (size_t)s->w[k] * 2 * sizeof(float), kAlign);
        total += flow_bytes;
        off_flow2.push_back(total); // further flow sets: a B-frame stream tick writes the flows of B pairs
        if (p->stream_batch >= 2) total += flow_bytes * (size_t)(p->stream_batch - 1);
        flow_stride.push_back(flow_bytes);
    }
    std::vector<size_t> off_patch[kSets];
    std::vector<size_t> off_pscr; // (per level; slot i, frame f at + (F i + f) * pscr_frame, F = frames per slot)
    size_t pscr_frame = 0;
    const int n_slots = p->stream_batch >= 2 ? p->stream_batch : 1;
    size_t patch_bytes[OFX_MAX_LEVELS] = {};
    // OFX_PLAIN_FUSED=1: the pair-at-a-time path's pyramid launch carries the pair's corner chain (two launches per pair instead of
    // three).  Off by default: measured SLOWER -- the chain's block takes 42-57 us (28-44 of them its patch build) against 13.5
    // (pyramid) + 12.1 us (the lone corner wave reading the finished pyramid) apart; profiles/r04_ablation.txt batch 10.
    const bool plain_fuse_wanted = !p->local_corner && !p->stream_two_stage && !p->sharded && p->levels >= 2 && p->levels - 1 <= 6 &&
                                   [] { const char *e = getenv("OFX_PLAIN_FUSED"); return e && atoi(e) != 0; }();
    bool plain_repair = false;
    if (p->local_corner || p->stream_two_stage || plain_fuse_wanted) {
        const int step = 1 << (p->levels - 1);
        int side = p->patch_size > 0 ? p->patch_size : step * ((p->window >> 1) + 2 + 8);
        if (p->patch_size <= 0 && side < 256) side = 256;
        if (plain_fuse_wanted && p->patch_size <= 0) { // (experiment: the side of the pair-at-a-time chain's patch)
            const char *e = getenv("OFX_PLAIN_PATCH");
            if (e && atoi(e) > 0) side = atoi(e);
        }
        side = (int)align_up((size_t)side, (size_t)step);
        const int pw0 = side < p->width ? side : p->width, ph0 = side < p->height ? side : p->height;
        for (int k = 0; k < p->levels; ++k) {
            s->pw[k] = pw0 >> k;
            s->ph[k] = ph0 >> k;
            s->ppitch[k] = (int)align_up((size_t)s->pw[k], 64);
            patch_bytes[k] = align_up((size_t)s->ppitch[k] * (size_t)s->ph[k] + 64, kAlign);
        }
        const int need = (p->window >> 1) + 2;
        const int lc = p->levels - 1;
        if (!plain_fuse_wanted && (s->pw[lc] < (need < s->w[lc] ? need : s->w[lc]) || s->ph[lc] < (need < s->h[lc] ? need : s->h[lc]))) {
            ofx_set_error("ofx_session_create: patch_size %d leaves %dx%d at the coarsest level, the corner needs %d", side, s->pw[lc],
                          s->ph[lc], need);
            delete s;
            return OFX_E_INVALID;
        }
        // With the frames borrowed the whole next frame is at hand when the chain runs: a shift that leaves the patch is repaired
        // (ofx_corner_stage.d_patch_reloc) -- provided the patch leaves room at the coarsest level to be placed around any
        // target (radius + 3 pixels of stencils + the plane's first column / row); a smaller patch (patch_size) keeps the
        // status bit.
        const int need_r = (p->window >> 1) + 5;
        const bool room = (s->pw[lc] >= need_r || s->pw[lc] >= s->w[lc]) && (s->ph[lc] >= need_r || s->ph[lc] >= s->h[lc]);
        s->repair = !plain_fuse_wanted && p->borrow_frames && !p->frames_partial && p->levels >= 3 && room;
        if (plain_fuse_wanted) {
            // the chain of such a session must be exact for every input (the stand-alone corner kernel reads whole planes): fused
            // only where a miss can be repaired, or cannot happen because the patch is the frame; even patch dimensions below the top
            const bool whole = pw0 == p->width && ph0 == p->height;
            const bool enough = s->pw[lc] >= (need < s->w[lc] ? need : s->w[lc]) && s->ph[lc] >= (need < s->h[lc] ? need : s->h[lc]);
            bool even = true;
            for (int k = 0; k + 1 < p->levels; ++k) even = even && (s->pw[k] & 1) == 0 && (s->ph[k] & 1) == 0;
            plain_repair = p->levels >= 3 && room && !whole;
            s->plain_fuse = enough && even && (whole || plain_repair);
        }
    }
    if (s->repair || plain_repair) {
        // Test hook: pretend the top-left patch planes are only this many level-0 pixels wide and high (never less than the
        // corner itself), so that ordinary frames drive the chain into the relocated planes; the results must not change.
        const char *e = getenv("OFX_DEBUG_CORNER_EXTENT");
        s->debug_extent = e ? atoi(e) : 0;
    }
    const int frames_per_slot = (p->stream_two_stage || s->plain_fuse ? 2 : 0) + (s->repair || (s->plain_fuse && plain_repair) ? 1 : 0);
    if (p->local_corner || p->stream_two_stage || s->plain_fuse) {
        for (int k = 0; k < p->levels; ++k) {
            off_pscr.push_back(pscr_frame);
            if (k >= 1) pscr_frame += patch_bytes[k];
            if (p->stream_two_stage || s->plain_fuse) continue; // no patch pyramids per image set: the corner blocks build what they read
            for (int t = 0; t < n_sets; ++t) {
                off_patch[t].push_back(total);
                total += patch_bytes[k];
            }
        }
        const size_t off_pscr_base = total;
        total += pscr_frame * (size_t)frames_per_slot * (size_t)n_slots;
        for (size_t &o : off_pscr) o += off_pscr_base;
    }
    const size_t off_status = total; // the sticky word, then one word per shift-vector slot
    total += align_up(sizeof(int) * (size_t)(1 + kUvSlots), kAlign);
    const size_t off_uv = total;
    total += align_up((size_t)OFX_MAX_LEVELS * 2 * sizeof(float) * kUvSlots, kAlign);
    const size_t off_staging = total; // one 3-channel frame for ofx_session_set_frame_host_3ch (unsharded sessions only)
    if (!p->sharded) total += align_up((size_t)p->width * (size_t)p->height * 3, kAlign);

    hipError_t e = hipMalloc(&s->arena, total);
    if (e != hipSuccess) {
        ofx_set_error("ofx_session_create: hipMalloc(%zu bytes): %s", total, hipGetErrorString(e));
        delete s;
        return OFX_E_HIP;
    }
    s->arena_bytes = total;
    e = hipMemset(s->arena, 0, total);
    if (e != hipSuccess) {
        ofx_set_error("ofx_session_create: hipMemset: %s", hipGetErrorString(e));
        (void)hipFree(s->arena);
        delete s;
        return OFX_E_HIP;
    }
    uint8_t *base = static_cast<uint8_t *>(s->arena);
    for (int k = 0; k < p->levels; ++k) {
        for (int t = 0; t < n_sets; ++t) s->img[t][k] = base + off_plane[t][k];
        for (int t = 0; t < 2; ++t) s->sh[t][k] = base + off_plane[n_sets + t][k];
        for (int t = 0; t < n_iter_sets; ++t) s->itsh[t / 3][t % 3][k] = base + off_plane[n_sets + 2 + t][k];
        s->flowset[0][k] = reinterpret_cast<float *>(base + off_flow[k]);
        for (int t = 1; t < kMaxBatch; ++t)
            s->flowset[t][k] = reinterpret_cast<float *>(base + (t < p->stream_batch ? off_flow2[k] + (size_t)(t - 1) * flow_stride[k] : off_flow[k]));
        s->flow[k] = s->flowset[0][k];
    }
    if (p->local_corner && !p->stream_two_stage)
        for (int k = 0; k < p->levels; ++k)
            for (int t = 0; t < n_sets; ++t) s->pimg[t][k] = base + off_patch[t][k];
    for (int i = 0; i < n_slots && frames_per_slot > 0; ++i)
        for (int k = 1; k < p->levels; ++k) {
            uint8_t *slot = base + off_pscr[k] + (size_t)(frames_per_slot * i) * pscr_frame;
            if (p->stream_two_stage || s->plain_fuse) {
                s->pscr[i][0][k] = slot;
                s->pscr[i][1][k] = slot + pscr_frame;
            }
            if (s->repair || (s->plain_fuse && plain_repair)) s->preloc[i][k] = slot + (size_t)(frames_per_slot - 1) * pscr_frame;
        }
    s->corner_status = reinterpret_cast<int *>(base + off_status);
    s->pair_status = s->corner_status + 1;
    s->uv = reinterpret_cast<float *>(base + off_uv);
    s->staging = p->sharded ? nullptr : base + off_staging;
    // OFX_ITER_FUSED=0 keeps one ofx_warp_levels launch per refinement iteration
    s->fused_iters = p->iters > 1 && [] { const char *e = getenv("OFX_ITER_FUSED"); return !e || atoi(e) != 0; }();
    // the launches that also write the warped image run on 32-bit buffer offsets (lk_body_buf.h): a session whose level 0 reaches
    // 2 GB of plane or of flow rows keeps the warp launch + the old accumulating march, as before round 3
    if ((size_t)(s->buf1[0] - s->buf0[0]) * (size_t)s->pitch[0] >= ((size_t)1 << 31) ||
        (size_t)(s->buf1[0] - s->buf0[0]) * (size_t)s->w[0] * 8 >= ((size_t)1 << 31))
        s->fused_iters = false;
    repoint(s);
    *out = s;
    return OFX_OK;
}

extern "C" int ofx_session_destroy(ofx_session *s)
{
    if (!s) return OFX_OK;
    hipError_t e = hipSuccess;
    (void)hipSetDevice(s->p.device);
    for (hipEvent_t ev : s->ev) (void)hipEventDestroy(ev);
    if (s->ev_frame) (void)hipEventDestroy(s->ev_frame);
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    for (hipEvent_t ev : s->ev_set_done)
        if (ev) (void)hipEventDestroy(ev);
    if (s->aux) (void)hipStreamDestroy(s->aux);
    if (s->arena) e = hipFree(s->arena);
    delete s;
    if (e != hipSuccess) {
        ofx_set_error("ofx_session_destroy: hipFree: %s", hipGetErrorString(e));
        return OFX_E_HIP;
    }
    return OFX_OK;
}

// the image set of the next frame has new contents: patch planes built from its earlier contents and vectors formed with it are stale
static void new_next(ofx_session *s)
{
    ++s->img_gen[(s->cur + 1) % 3];
    s->corner_done = false;
}

// copy rows [buf0,buf1) of a tightly packed w-bytes-per-row frame into the level-0 `next` plane
static int load_level0(ofx_session *s, const uint8_t *src, bool src_is_host, int src_pitch, hipStream_t st)
{
    const int rows = s->buf1[0] - s->buf0[0];
    if (s->pframe[(s->cur + 1) % 3]) { // (a borrowing session that is handed a host frame, or staged: back to its own plane)
        s->pframe[(s->cur + 1) % 3] = nullptr;
        repoint(s);
    }
    OFX_HIP(hipMemcpy2DAsync(s->plane[1][0], (size_t)s->pitch[0], src + (size_t)s->buf0[0] * (size_t)src_pitch, (size_t)src_pitch,
                             (size_t)s->w[0], (size_t)rows, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    s->have_next = true;
    new_next(s);
    return OFX_OK;
}

extern "C" int ofx_session_set_frame_host(ofx_session *s, const uint8_t *h_gray1, void *stream)
{
    OFX_REQUIRE(s && h_gray1, "ofx_session_set_frame_host: null argument");
    return load_level0(s, h_gray1, true, s->w[0], ofx_stream(stream));
}

extern "C" int ofx_session_set_frame_device(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream)
{
    OFX_REQUIRE(s && d_gray1, "ofx_session_set_frame_device: null argument");
    OFX_REQUIRE(pitch >= s->w[0], "ofx_session_set_frame_device: pitch %d < width %d", pitch, s->w[0]);
    if (s->p.borrow_frames) {
        // no copy: the pyramid, corner and LK launches read the caller's buffer in place -- until the run_flow of the pair in
        // which this frame is the PREVIOUS one has run (include/ofx.h, borrow_frames)
        OFX_REQUIRE(pitch == s->pitch[0] && ((uintptr_t)d_gray1 & 3) == 0,
                    "ofx_session_set_frame_device: a borrowed frame needs the session's level-0 pitch (%d bytes: the width rounded up to 64; "
                    "got %d) and a 4-byte aligned address", s->pitch[0], pitch);
        s->pframe[(s->cur + 1) % 3] = d_gray1;
        repoint(s);
        s->have_next = true;
        new_next(s);
        return OFX_OK;
    }
    return load_level0(s, d_gray1, false, pitch, ofx_stream(stream));
}

extern "C" int ofx_session_set_frame_host_3ch(ofx_session *s, const uint8_t *h_img3, void *stream)
{
    OFX_REQUIRE(s && h_img3, "ofx_session_set_frame_host_3ch: null argument");
    OFX_REQUIRE(!s->p.sharded, "ofx_session_set_frame_host_3ch: not available on a sharded session");
    hipStream_t st = ofx_stream(stream);
    if (s->pframe[(s->cur + 1) % 3]) {
        s->pframe[(s->cur + 1) % 3] = nullptr;
        repoint(s);
    }
    OFX_HIP(hipMemcpyAsync(s->staging, h_img3, (size_t)s->w[0] * (size_t)s->h[0] * 3, hipMemcpyHostToDevice, st));
    // the reference reads channel 0 only (OptFlowCPU.cpp:102, OptFlowGpu.cu:1079)
    OFX_TRY(ofx_extract_ch0(s->staging, s->plane[1][0], s->w[0], s->h[0], s->pitch[0], stream));
    s->have_next = true;
    new_next(s);
    return OFX_OK;
}

extern "C" int ofx_session_downsample_level(ofx_session *s, int k, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_downsample_level: null session");
    OFX_REQUIRE(k >= 1 && k < s->p.levels, "ofx_session_downsample_level: level %d out of range", k);
    OFX_REQUIRE(s->have_next, "ofx_session_downsample_level: no frame loaded");
    const ofx_geom g = level_geom(s, k, s->cmp0[k], s->cmp1[k]);
    return ofx_downsample_1ch(s->plane[1][k - 1], s->pitch[k - 1], s->buf0[k - 1], s->buf1[k - 1] - s->buf0[k - 1],
                              s->plane[1][k], &g, stream);
}

static int build_pyramid(ofx_session *s, void *stream)
{
    if (s->p.levels == 1) return OFX_OK;
    if (!s->p.sharded && s->p.levels - 1 <= 6) { // whole levels: one fused launch
        uint8_t *lv[OFX_MAX_LEVELS] = {};
        int pitches[OFX_MAX_LEVELS] = {};
        for (int k = 1; k < s->p.levels; ++k) {
            lv[k] = s->plane[1][k];
            pitches[k] = s->pitch[k];
        }
        if (s->plain_fuse && s->have_prev && (s->pitch0_next() & 3) == 0 && (((uintptr_t)s->plane[0][0] | (uintptr_t)s->plane[1][0]) & 3) == 0) {
            // the pair's corner chain rides in one more block of this launch (pyr_corner.hip); ofx_session_corner_flows then has
            // nothing left to do.  The previous frame's patch planes: the ones built when it was the next frame, if they still are.
            const int L = s->p.levels, ip = s->cur, in = (s->cur + 1) % 3;
            int have = -1;
            for (int t = 0; t < 2; ++t)
                if (s->pset_img[t] == ip && s->pset_gen[t] == s->img_gen[ip]) have = t;
            const int tp = have >= 0 ? have : 0, tn = 1 - tp;
            ofx_corner_stage C{};
            C.levels = L;
            C.d_uv = s->uv_cur();
            C.build_patch = 1;
            C.patch_w = s->pw[0];
            C.patch_h = s->ph[0];
            C.d_patch_src[0] = s->plane[0][0], C.d_patch_src[1] = s->plane[1][0];
            C.patch_src_pitch[0] = s->pitch[0], C.patch_src_pitch[1] = s->pitch0_next();
            auto extent = [&](int k, int full) { // (OFX_DEBUG_CORNER_EXTENT: as the stream pipeline's chain_extent)
                if (s->debug_extent <= 0 || k == 0 || s->preloc[0][1] == nullptr) return full;
                const int need = (s->p.window >> 1) + 2, lim = s->debug_extent >> k;
                const int e = lim > need ? lim : need;
                return e < full ? e : full;
            };
            for (int k = 0; k < L; ++k) {
                C.patch_pitch[k] = s->ppitch[k];
                C.d_patch[0][k] = s->pscr[0][tp][k];
                C.d_patch[1][k] = s->pscr[0][tn][k];
                C.d_patch_reloc[k] = s->preloc[0][k];
                const uint8_t *pp = k ? s->pscr[0][tp][k] : s->plane[0][0], *pn = k ? s->pscr[0][tn][k] : s->plane[1][0];
                ofx_geom pg{s->w[k], s->h[k], k ? s->ppitch[k] : s->pitch[0], 0, k ? extent(k, s->ph[k]) : s->h[0], 0, k ? extent(k, s->ph[k]) : s->h[0]};
                C.level[k] = ofx_lk_desc{pp, pn, pg, nullptr, 0, nullptr, 0, s->p.min_det};
                C.cols[k] = k ? extent(k, s->pw[k]) : 0;
            }
            OFX_TRY(timed_launch(s, OFX_TIME_PYRAMID, stream, [&] {
                return ofx_pyramid_corner_1ch(s->plane[1][0], s->pitch0_next(), s->w[0], s->h[0], lv, pitches, L, &C, have >= 0 ? 1 : 0, s->p.window, s->p.mode,
                                              stream);
            }));
            s->pset_img[tp] = ip, s->pset_gen[tp] = s->img_gen[ip];
            s->pset_img[tn] = in, s->pset_gen[tn] = s->img_gen[in];
            s->corner_done = true;
            return OFX_OK;
        }
        return timed_launch(s, OFX_TIME_PYRAMID, stream,
                            [&] { return ofx_pyramid_1ch(s->plane[1][0], s->pitch0_next(), s->w[0], s->h[0], lv, pitches, s->p.levels, stream); });
    }
    for (int k = 1; k < s->p.levels; ++k) OFX_TRY(ofx_session_downsample_level(s, k, stream));
    return OFX_OK;
}

extern "C" int ofx_session_build_pyramid(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_build_pyramid: null session");
    if (!s->have_next) {
        ofx_set_error("ofx_session_build_pyramid: no frame loaded");
        return OFX_E_STATE;
    }
    return build_pyramid(s, stream);
}

extern "C" int ofx_session_compute_uv(ofx_session *s, int level, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_compute_uv: null session");
    OFX_REQUIRE(level >= 0 && level < s->p.levels, "ofx_session_compute_uv: level %d out of range", level);
    if (level == s->p.levels - 1) return OFX_OK; // top level is not shifted (OptFlowCPU.cpp:321)
    const float *lv[OFX_MAX_LEVELS] = {};
    for (int k = 0; k < s->p.levels; ++k) lv[k] = s->flow[k];
    return ofx_shift_vector(lv, level, s->p.levels, s->uv_cur() + 2 * level, stream);
}

extern "C" int ofx_session_run_level(ofx_session *s, int level, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_level: null session");
    OFX_REQUIRE(level >= 0 && level < s->p.levels, "ofx_session_run_level: level %d out of range", level);
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_run_level: need a previous and a next frame (load, build, swap, load, build)");
        return OFX_E_STATE;
    }
    // (the flow buffers of such a session start above the own rows -- fl0 < own0 -- and only the stream pipeline addresses them so)
    OFX_REQUIRE(!(s->p.sharded && s->p.iters > 1), "ofx_session_run_level: refinement iterations on a sharded session run through the stream "
                                                    "pipeline (ofx_session_stream_*)");
    const uint8_t *next = s->plane[1][level];
    if (level != s->p.levels - 1) {
        // shift every row the LK stencil will read: own rows +- (radius + 1), clipped to the buffer
        const int halo = (s->p.window >> 1) + 1;
        int y0 = s->own0[level] - halo, y1 = s->own1[level] + halo;
        if (y0 < s->buf0[level]) y0 = s->buf0[level];
        if (y1 > s->buf1[level]) y1 = s->buf1[level];
        const ofx_geom gs = level_geom(s, level, y0, y1);
        OFX_TRY(ofx_shift_1ch(s->plane[1][level], s->plane[2][level], &gs, s->uv_cur() + 2 * level, stream));
        next = s->plane[2][level];
    }
    const ofx_geom g = level_geom(s, level, s->own0[level], s->own1[level]);
    auto launch = [&] { return ofx_lk_level(s->plane[0][level], next, &g, s->p.window, s->p.mode, s->flow[level], s->own0[level], stream); };
    if (level != 0) return launch();
    return timed_launch(s, OFX_TIME_LK, stream, launch);
}

extern "C" int ofx_session_timing(ofx_session *s, int max_launches)
{
    OFX_REQUIRE(s && max_launches >= 0, "ofx_session_timing: bad arguments");
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    s->ev.clear();
    s->ev_kind.assign((size_t)max_launches, 0);
    s->ev_used = 0;
    s->timing = max_launches > 0;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t e;
        OFX_HIP(hipEventCreate(&e));
        s->ev.push_back(e);
    }
    return OFX_OK;
}

// average / minimum over the recorded launches whose kind is in `kind_mask` (bit OFX_TIME_*); does not re-arm
static int timing_stats(ofx_session *s, unsigned kind_mask, double *avg_us, double *min_us, int *launches)
{
    double sum = 0, mn = 1e30;
    int n = 0;
    for (size_t i = 0; i < s->ev_used / 2; ++i) {
        if (!((kind_mask >> s->ev_kind[i]) & 1u)) continue;
        OFX_HIP(hipEventSynchronize(s->ev[2 * i + 1]));
        float ms = 0;
        OFX_HIP(hipEventElapsedTime(&ms, s->ev[2 * i], s->ev[2 * i + 1]));
        sum += ms * 1e3;
        if (ms * 1e3 < mn) mn = ms * 1e3;
        ++n;
    }
    *avg_us = n ? sum / n : 0.0;
    if (min_us) *min_us = n ? mn : 0.0;
    *launches = n;
    return OFX_OK;
}

extern "C" int ofx_session_timing_read(ofx_session *s, double *avg_us, double *min_us, int *launches)
{
    OFX_REQUIRE(s && avg_us && launches, "ofx_session_timing_read: bad arguments");
    // the dominant launches: the fused LK launches of the pair-at-a-time paths, the stream tick of the stream pipeline
    OFX_TRY(timing_stats(s, (1u << OFX_TIME_LK) | (1u << OFX_TIME_LK_ACC) | (1u << OFX_TIME_LK_ACC_WARP) | (1u << OFX_TIME_STREAM), avg_us, min_us, launches));
    s->ev_used = 0;
    return OFX_OK;
}

extern "C" int ofx_session_timing_read_kind(ofx_session *s, int kind, double *avg_us, double *min_us, int *launches)
{
    OFX_REQUIRE(s && avg_us && launches && kind >= 0 && kind < OFX_TIME_KINDS, "ofx_session_timing_read_kind: bad arguments");
    return timing_stats(s, 1u << kind, avg_us, min_us, launches);
}

// Shift vectors of every level at once (ofx_corner_flows); meaningful on the rank whose buffers start at row 0.
extern "C" int ofx_session_corner_flows(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_corner_flows: null session");
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_corner_flows: need a previous and a next frame");
        return OFX_E_STATE;
    }
    if (s->corner_done) return OFX_OK; // (ofx_session_build_pyramid's launch has formed them: pyr_corner.hip)
    ofx_lk_desc d[OFX_MAX_LEVELS];
    for (int k = 0; k < s->p.levels; ++k)
        d[k] = ofx_lk_desc{s->plane[0][k], s->plane[1][k], level_geom(s, k, s->own0[k], s->own1[k]), nullptr, s->own0[k], nullptr, 0, s->p.min_det};
    return timed_launch(s, OFX_TIME_CORNER, stream, [&] { return ofx_corner_flows(d, s->p.levels, s->p.window, s->p.mode, s->uv_cur(), stream); });
}

// Every level's fused LK in one launch, using the uv slots as they stand; below the top level the kernel reads `next`
// through the global shift (no separate shift pass, no shifted planes).
// one multi-level LK launch, bracketed by timing events when the session is armed (ofx_session_timing)
static int timed_lk_launch(ofx_session *s, const ofx_lk_desc *lk, int nl, void *stream)
{
    return timed_launch(s, lk[0].accumulate ? (lk[0].d_warp_out ? OFX_TIME_LK_ACC_WARP : OFX_TIME_LK_ACC) : OFX_TIME_LK, stream,
                        [&] { return ofx_lk_levels(lk, nl, s->p.window, s->p.mode, stream); });
}

static int lk_all_levels(ofx_session *s, const float *uv, void *stream)
{
    const int L = s->p.levels;
    ofx_lk_desc lk[OFX_MAX_LEVELS];
    if (s->p.iters <= 1) {
        int nl = 0;
        for (int k = L - 1; k >= 0; --k) // coarse levels first: their few waves start at once and finish early
            lk[nl++] = ofx_lk_desc{s->plane[0][k], s->plane[1][k], level_geom(s, k, s->own0[k], s->own1[k]), s->flow[k], s->own0[k],
                                   k == L - 1 ? nullptr : uv + 2 * k, 0, s->p.min_det};
        return timed_lk_launch(s, lk, nl, stream);
    }
    OFX_REQUIRE(!s->p.sharded, "refinement iterations on a sharded session run through the stream pipeline (ofx_session_stream_*)");
    // Extension (DESIGN.md "lk_iter"): iteration 1 is the reference level; every further iteration warps
    // the shifted next image by the flow so far (bilinear, rounded to u8) and adds the flow of (prev, warped).
    // sh[0] holds the globally shifted next image (the warp source), sh[1] the warped image.
    ofx_shift_desc sd[OFX_MAX_LEVELS];
    int ns = 0;
    for (int k = L - 2; k >= 0; --k)
        sd[ns++] = ofx_shift_desc{s->plane[1][k], s->sh[0][k], level_geom(s, k, 0, s->h[k]), uv + 2 * k};
    if (ns) OFX_TRY(timed_launch(s, OFX_TIME_SHIFT, stream, [&] { return ofx_shift_levels(sd, ns, stream); }));
    auto src = [&](int k) { return k == L - 1 ? s->plane[1][k] : s->sh[0][k]; };
    // The warped image alternates between two planes (sh[1] and the iteration scratch's third): fused (lk_body_warp.h), every launch
    // but the last writes the warped image the iteration after it reads, from the flow it has in registers -- no warp launch at all.
    uint8_t *const *wbuf[2] = {s->sh[1], s->itsh[0][2]};
    int nl = 0;
    for (int k = L - 1; k >= 0; --k) {
        lk[nl] = ofx_lk_desc{s->plane[0][k], src(k), level_geom(s, k, 0, s->h[k]), s->flow[k], 0, nullptr, 0, s->p.min_det};
        if (s->fused_iters) lk[nl].d_warp_src = src(k), lk[nl].d_warp_out = wbuf[0][k], lk[nl].warp_scale = OFX_ITER_SCALE;
        ++nl;
    }
    OFX_TRY(timed_lk_launch(s, lk, nl, stream));
    for (int it = 1; it < s->p.iters; ++it) {
        const bool fused = s->fused_iters, need_warp = !fused, wout = fused && it + 1 < s->p.iters;
        uint8_t *const *win = fused ? wbuf[(it - 1) & 1] : s->sh[1], *const *wnext = wbuf[it & 1];
        ofx_warp_desc wd[OFX_MAX_LEVELS];
        nl = 0;
        for (int k = L - 1; k >= 0; --k) {
            wd[nl] = ofx_warp_desc{src(k), win[k], level_geom(s, k, 0, s->h[k]), s->flow[k], 0, OFX_ITER_SCALE, nullptr, 0};
            lk[nl] = ofx_lk_desc{s->plane[0][k], win[k], level_geom(s, k, 0, s->h[k]), s->flow[k], 0, nullptr, 1, s->p.min_det};
            if (wout) lk[nl].d_warp_src = src(k), lk[nl].d_warp_out = wnext[k], lk[nl].warp_scale = OFX_ITER_SCALE;
            ++nl;
        }
        if (need_warp) OFX_TRY(timed_launch(s, OFX_TIME_WARP, stream, [&] { return ofx_warp_levels(wd, nl, stream); }));
        OFX_TRY(timed_lk_launch(s, lk, nl, stream));
    }
    return OFX_OK;
}

// per level: the image rows the LK stencils of this shard's own rows touch (before the shift) and the rows its buffers hold
static void shard_reach(const ofx_session *s, int (*rows)[4])
{
    const int reach = s->p.window / 2 + 1; // the LK stencil of the rows it computes reaches radius + 1 rows beyond them
    for (int k = 0; k < s->p.levels; ++k) {
        const int n0 = s->fl0[k] - reach, n1 = s->fl1[k] + reach;
        rows[k][0] = n0 < 0 ? 0 : n0;
        rows[k][1] = n1 > s->h[k] ? s->h[k] : n1;
        rows[k][2] = s->buf0[k]; // (rows outside comp but inside buf are the caller's to fill: the halo exchange)
        rows[k][3] = s->buf1[k];
    }
}

extern "C" int ofx_session_run_levels(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_levels: null session");
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_run_levels: need a previous and a next frame");
        return OFX_E_STATE;
    }
    if (s->p.sharded && !s->p.local_corner && s->p.levels > 1) {
        // the shift vectors were computed elsewhere (rank 0's corner kernel, then the broadcast): check them against this
        // shard's halo on the device, without a host round trip (ofx_session_corner_status reads the word)
        int rows[OFX_MAX_LEVELS][4];
        shard_reach(s, rows);
        OFX_TRY(ofx_shard_margin_check(s->uv_cur(), s->p.levels, s->h, &rows[0][0], s->corner_status, stream));
    }
    return lk_all_levels(s, s->uv_cur(), stream);
}

extern "C" int ofx_session_run_flow(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_flow: null session");
    OFX_REQUIRE(!s->p.sharded || s->buf0[0] == 0, "ofx_session_run_flow: on a sharded session only the rank holding row 0 can "
                                                   "form the shift vectors; use corner_flows + broadcast + run_levels");
    OFX_TRY(ofx_session_corner_flows(s, stream));
    return ofx_session_run_levels(s, stream);
}

// The reference's literal sequence (one level after the other, main.cu:256-262); kept for comparison and tests.
extern "C" int ofx_session_run_flow_sequential(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_flow_sequential: null session");
    for (int k = s->p.levels - 1; k >= 0; --k) {
        OFX_TRY(ofx_session_compute_uv(s, k, stream));
        OFX_TRY(ofx_session_run_level(s, k, stream));
    }
    return OFX_OK;
}

extern "C" int ofx_session_swap(ofx_session *s)
{
    OFX_REQUIRE(s, "ofx_session_swap: null session");
    s->cur = (s->cur + 1) % 3;
    s->uv_slot ^= 1;
    repoint(s);
    s->have_prev = s->have_next;
    s->have_next = false;
    s->staged = false;
    s->corner_done = false;
    return OFX_OK;
}

// ---- pipelined path ---------------------------------------------------------------------------------------------------
// A pair is split in two halves that run on different streams:
//   staging (aux stream):  load the new frame, build its pyramid, corner flows, shift of every level
//   solve   (main stream): the multi-level LK launch
// The LK launch of pair i is VALU-bound and fills the chip; the staging kernels of pair i+1 are small and latency-bound,
// so running them underneath it hides them almost entirely.  Hazards are covered by two events: `ev_ready` (staging ->
// LK of the same pair) and `ev_set_done[x]` (LK that read image set x as `prev` -> the staging that overwrites set x two
// pairs later; the same event also orders the reuse of the shifted-scratch set).
static int ensure_pipeline(ofx_session *s)
{
    if (s->ev_ready) return OFX_OK;
    OFX_HIP(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    OFX_HIP(hipEventCreateWithFlags(&s->ev_frame, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) OFX_HIP(hipEventCreateWithFlags(&s->ev_set_done[i], hipEventDisableTiming));
    // highest priority: the staging kernels are tiny and must slip in between the LK waves of the previous pair
    int prio_lo = 0, prio_hi = 0;
    OFX_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    OFX_HIP(hipStreamCreateWithPriority(&s->aux, hipStreamNonBlocking, prio_hi));
    return OFX_OK;
}

extern "C" int ofx_session_aux_stream(ofx_session *s, void **stream)
{
    OFX_REQUIRE(s && stream, "ofx_session_aux_stream: null argument");
    OFX_TRY(ensure_pipeline(s));
    *stream = s->aux;
    return OFX_OK;
}

// staging, part 1: frame -> next image set, pyramid.  `d_gray1` is read on the staging stream: the caller orders that stream
// behind the frame's producer (ofx_session_submit_device does it with an event on its `stream` argument).
extern "C" int ofx_session_stage_frame(ofx_session *s, const uint8_t *d_gray1, int pitch, void *aux_stream)
{
    OFX_REQUIRE(s && d_gray1, "ofx_session_stage_frame: null argument");
    OFX_REQUIRE(pitch >= s->w[0], "ofx_session_stage_frame: pitch %d < width %d", pitch, s->w[0]);
    if (!s->have_prev) {
        ofx_set_error("ofx_session_stage_frame: no previous frame (load, build, swap first)");
        return OFX_E_STATE;
    }
    OFX_TRY(ensure_pipeline(s));
    hipStream_t aux = aux_stream ? ofx_stream(aux_stream) : s->aux;
    const int nxt = (s->cur + 1) % 3;
    if (s->set_busy[nxt]) OFX_HIP(hipStreamWaitEvent(aux, s->ev_set_done[nxt], 0));
    OFX_TRY(load_level0(s, d_gray1, false, pitch, aux));
    return build_pyramid(s, aux);
}

// staging, part 2 (after the corner flows / the broadcast of the shift vectors): signal the solve half.  (The shift
// itself is fused into the LK launch; the name is kept from when it was a separate pass.)
extern "C" int ofx_session_stage_shift(ofx_session *s, void *aux_stream)
{
    OFX_REQUIRE(s, "ofx_session_stage_shift: null session");
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_stage_shift: stage a frame first");
        return OFX_E_STATE;
    }
    OFX_TRY(ensure_pipeline(s));
    hipStream_t aux = aux_stream ? ofx_stream(aux_stream) : s->aux;
    OFX_HIP(hipEventRecord(s->ev_ready, aux));
    s->staged = true;
    return OFX_OK;
}

// solve half: one multi-level LK launch on the caller's stream, then the staged frame becomes the previous frame.
extern "C" int ofx_session_solve_staged(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_solve_staged: null session");
    if (!s->staged) {
        ofx_set_error("ofx_session_solve_staged: nothing staged (stage_frame, corner_flows, stage_shift first)");
        return OFX_E_STATE;
    }
    hipStream_t st = ofx_stream(stream);
    OFX_HIP(hipStreamWaitEvent(st, s->ev_ready, 0));
    OFX_TRY(lk_all_levels(s, s->uv_cur(), stream));
    OFX_HIP(hipEventRecord(s->ev_set_done[s->cur], st));
    s->set_busy[s->cur] = true;
    return ofx_session_swap(s);
}

// Whole pair, pipelined: staging on the session's aux stream, solve on `stream`; on return the new frame is the
// previous frame.  Equivalent to set_frame_device + build_pyramid + run_flow + swap, bit for bit.
extern "C" int ofx_session_submit_device(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_submit_device: null session");
    OFX_REQUIRE(!s->p.sharded || s->buf0[0] == 0, "ofx_session_submit_device: on a sharded session drive the halves yourself "
                                                   "(stage_frame, corner_flows on the rank holding row 0, broadcast, stage_shift, "
                                                   "solve_staged)");
    // the frame may still be in production on the caller's stream (an async upload, a decoder or grayscale kernel): the
    // staging stream reads it, so it is ordered behind everything enqueued on `stream` so far
    OFX_TRY(ensure_pipeline(s));
    OFX_HIP(hipEventRecord(s->ev_frame, ofx_stream(stream)));
    OFX_HIP(hipStreamWaitEvent(s->aux, s->ev_frame, 0));
    OFX_TRY(ofx_session_stage_frame(s, d_gray1, pitch, nullptr));
    OFX_TRY(ofx_session_corner_flows(s, s->aux));
    OFX_TRY(ofx_session_stage_shift(s, nullptr));
    return ofx_session_solve_staged(s, stream);
}

extern "C" int ofx_session_plane(ofx_session *s, int which, int level, uint8_t **d_ptr, ofx_geom *geom)
{
    OFX_REQUIRE(s && which >= 0 && which < 3 && level >= 0 && level < s->p.levels, "ofx_session_plane: bad arguments");
    if (d_ptr) *d_ptr = s->plane[which][level];
    if (geom) *geom = level_geom(s, level, s->own0[level], s->own1[level]);
    return OFX_OK;
}

extern "C" int ofx_session_flow(ofx_session *s, int level, float **d_ptr, int *row0, int *rows)
{
    OFX_REQUIRE(s && level >= 0 && level < s->p.levels, "ofx_session_flow: bad arguments");
    if (d_ptr) *d_ptr = s->flow[level] + s->flow_own_offset(level);
    if (row0) *row0 = s->own0[level];
    if (rows) *rows = s->own1[level] - s->own0[level];
    return OFX_OK;
}

extern "C" int ofx_session_shift_uv(ofx_session *s, int level, float **d_uv)
{
    OFX_REQUIRE(s && d_uv && level >= 0 && level < s->p.levels, "ofx_session_shift_uv: bad arguments");
    *d_uv = s->uv_cur() + 2 * level; // slot of the pair in progress (alternates per pair)
    return OFX_OK;
}

extern "C" int ofx_session_get_flow_host(ofx_session *s, int level, float *h_dst, void *stream)
{
    OFX_REQUIRE(s && h_dst && level >= 0 && level < s->p.levels, "ofx_session_get_flow_host: bad arguments");
    const size_t bytes = (size_t)(s->own1[level] - s->own0[level]) * (size_t)s->w[level] * 2 * sizeof(float);
    hipStream_t st = ofx_stream(stream);
    OFX_HIP(hipMemcpyAsync(h_dst, s->flow[level] + s->flow_own_offset(level), bytes, hipMemcpyDeviceToHost, st));
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

extern "C" int ofx_session_corner_status(ofx_session *s, int *h_status, void *stream)
{
    OFX_REQUIRE(s && h_status, "ofx_session_corner_status: null argument");
    hipStream_t st = ofx_stream(stream);
    OFX_HIP(hipMemcpyAsync(h_status, s->corner_status, sizeof(int), hipMemcpyDeviceToHost, st));
    OFX_HIP(hipMemsetAsync(s->corner_status, 0, sizeof(int), st));
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

extern "C" int ofx_session_pair_status(ofx_session *s, int pair, int *h_status, void *stream)
{
    OFX_REQUIRE(s && h_status, "ofx_session_pair_status: null argument");
    const int slots = 2 * (s->p.stream_batch >= 2 ? s->p.stream_batch : 1);
    OFX_REQUIRE(pair >= 1, "ofx_session_pair_status: pairs are counted from 1 (frame 0 -> frame 1)");
    OFX_REQUIRE(pair <= s->corner_newest && pair > s->corner_newest - slots,
                "ofx_session_pair_status: pair %d is not among the newest %d pairs whose corner stage has run (newest: %ld)", pair, slots, s->corner_newest);
    hipStream_t st = ofx_stream(stream);
    OFX_HIP(hipMemcpyAsync(h_status, s->pair_status + (pair % slots), sizeof(int), hipMemcpyDeviceToHost, st));
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

// ---- stream pipeline: one launch per tick of B frames ----------------------------------------------------------------
// Frame f (0-based) belongs to tick f / B (B = stream_batch: 1, 2 or 4).  Pair p is (frame p-1 -> frame p).  The tick
// whose first frame is f0 runs, side by side in one grid,
//     pyramid(frames f0 .. f0+B-1) | corner(pairs f0-B .. f0-1) | LK(pairs f0-2B .. f0-B-1, shift fused)
// (with ofx_params.stream_two_stage: pyramid(f0 .. f0+B-1) | corner(pairs f0 .. f0+B-1, on patch pyramids the corner blocks
// build themselves) | LK(pairs f0-B .. f0-1): 2B + 2 image sets, a pair done one tick earlier)
// so every stage consumes what earlier ticks wrote and the ticks are ordered by the stream.  Frame f lives in image set
// f mod (3B+2) and pair p's shift vectors in slot p mod 2B: a set is last read by LK(pair f+1), at the latest in the tick
// that starts with frame f+2B+1, and rewritten by the tick that holds frame f+3B+2; a slot is read by LK(pair p) one tick
// after the corner stage wrote it and rewritten two ticks after.  The flows of pair p go to flow set p mod B.  After a
// tick every pair <= f0-B-1 is done.
static int stream_batch_of(const ofx_session *s) { return s->p.stream_batch >= 2 ? s->p.stream_batch : 1; }

static int stream_tick(ofx_session *s, const uint8_t *const *frames, const int *pitches, int n_frames, void *stream, int *completed_pair)
{
    const int B = stream_batch_of(s);
    // D = ticks between a frame's arrival and the LK stage of the pair it completes: 2 (pyramid | corner | LK), or 1 with
    // stream_two_stage (the corner stage runs in the frame's own tick, on patch pyramids it builds itself)
    const int D = s->p.stream_two_stage ? 1 : 2;
    const int sets = (D + 1) * B + 2, slots = 2 * B;
    const long f0 = s->stream_n; // index of the first frame of this tick
    const int L = s->p.levels;
    auto uvslot = [&](long pair) { return s->uv + (size_t)(pair % slots) * 2 * OFX_MAX_LEVELS; };
    auto set_of = [&](long frame) { return (int)(frame % sets); };
    const long last_frame = s->stream_frames >= 0 ? s->stream_frames - 1 : f0 + n_frames - 1;
    // level k of a frame as the LK / corner stages see it: the session's plane, or (borrow_frames, level 0) the caller's buffer
    auto plane_of = [&](long frame, int k) -> const uint8_t * {
        const int set = set_of(frame);
        if (k == 0 && s->p.borrow_frames) return s->bframe[set] + (size_t)s->buf0[0] * (size_t)s->bpitch[set];
        return s->img[set][k];
    };
    auto patch_of = [&](long frame, int k) -> const uint8_t * {
        const int set = set_of(frame);
        return (k == 0 && s->p.borrow_frames) ? s->bframe[set] : s->pimg[set][k];
    };
    auto pitch_of = [&](long frame, int k, bool patch) {
        if (k == 0 && s->p.borrow_frames) return s->bpitch[set_of(frame)];
        return patch ? s->ppitch[k] : s->pitch[k];
    };
    // columns / rows of level k's patch planes the corner chain may read (all of them, unless the test hook narrows them)
    auto chain_extent = [&](int k, int full) {
        if (s->debug_extent <= 0 || k == 0) return full;
        const int need = (s->p.window >> 1) + 2, lim = s->debug_extent >> k;
        const int e = lim > need ? lim : need;
        return e < full ? e : full;
    };
    // the stages struct is several KB: keep it off the stack of callers with small stacks
    static thread_local ofx_stream_stages g;
    memset(&g, 0, sizeof g);
    for (int i = 0; i < n_frames; ++i) { // pyramid(frame f0 + i)
        OFX_REQUIRE(pitches[i] >= s->w[0] && (pitches[i] & 3) == 0 && ((uintptr_t)frames[i] & 3) == 0,
                    "ofx_session_stream_submit: frame must be 4-byte aligned with a pitch multiple of 4 and >= width");
        if (s->p.borrow_frames && f0 + i >= 1)
            OFX_REQUIRE(pitches[i] == s->bpitch[set_of(f0 + i - 1)] || (i > 0 && pitches[i] == pitches[i - 1]),
                        "ofx_session_stream_submit: borrowed frames must all have the same pitch");
        ofx_pyramid_stage &P = g.pyr[g.n_pyr++];
        const int set = set_of(f0 + i);
        P.d_frame = frames[i];
        P.frame_pitch = pitches[i];
        P.w = s->w[0];
        P.h = s->h[0];
        P.levels = L;
        P.windowed = s->p.sharded ? 1 : 0;
        for (int k = 0; k < L; ++k) {
            P.d_levels[k] = s->img[set][k];
            P.pitches[k] = s->pitch[k];
            P.row0[k] = s->buf0[k];
            P.rows[k] = s->buf1[k] - s->buf0[k];
        }
        if (s->p.local_corner && D == 2) { // the same frame's top-left patch, as a pyramid of its own
            P.patch_w = s->pw[0];
            P.patch_h = s->ph[0];
            P.patch_levels = L;
            for (int k = 0; k < L; ++k) {
                P.d_patch_levels[k] = s->pimg[set][k];
                P.patch_pitches[k] = s->ppitch[k];
            }
        }
        if (s->p.borrow_frames) { // no copies of level 0: the later stages read the caller's buffer
            s->bframe[set] = frames[i];
            s->bpitch[set] = pitches[i];
            P.d_levels[0] = nullptr;
            P.d_patch_levels[0] = nullptr;
        }
    }
    for (long pc = f0 - (D - 1) * B; pc <= f0 - (D - 1) * B + B - 1; ++pc) { // corner(pair pc)
        if (pc < 1 || pc > last_frame) continue;
        const int slot_i = g.n_corner;
        if (pc > s->corner_newest) s->corner_newest = pc;
        ofx_corner_stage &C = g.corner[g.n_corner++];
        C.levels = L;
        C.d_uv = uvslot(pc);
        if (D == 1) {
            // two stages: the pair's second frame arrived with this tick; the block builds the patch pyramids of both frames
            // (levels >= 1) into its slot's planes and walks the chain on them (level 0: the frames themselves, borrowed)
            C.build_patch = 1;
            C.patch_w = s->pw[0];
            C.patch_h = s->ph[0];
            for (int f = 0; f < 2; ++f) {
                C.d_patch_src[f] = s->bframe[set_of(pc - 1 + f)];
                C.patch_src_pitch[f] = s->bpitch[set_of(pc - 1 + f)];
            }
            for (int k = 0; k < L; ++k) {
                C.patch_pitch[k] = s->ppitch[k];
                C.d_patch[0][k] = s->pscr[slot_i][0][k];
                C.d_patch[1][k] = s->pscr[slot_i][1][k];
                C.d_patch_reloc[k] = s->repair ? s->preloc[slot_i][k] : nullptr;
                const uint8_t *pp = k ? s->pscr[slot_i][0][k] : C.d_patch_src[0], *pn = k ? s->pscr[slot_i][1][k] : C.d_patch_src[1];
                // level 0 is the frames themselves, whole (every rank of a sharded stream is handed whole frames)
                ofx_geom pg{s->w[k], s->h[k], k ? s->ppitch[k] : C.patch_src_pitch[1], 0, k ? chain_extent(k, s->ph[k]) : s->h[0], 0,
                            k ? chain_extent(k, s->ph[k]) : s->h[0]};
                C.level[k] = ofx_lk_desc{pp, pn, pg, nullptr, 0, nullptr, 0, s->p.min_det};
                C.cols[k] = k ? chain_extent(k, s->pw[k]) : 0;
            }
            C.d_status = s->corner_status;
            C.d_pair_status = s->pair_status + (pc % slots);
            if (s->p.sharded) shard_reach(s, C.shard_rows);
            continue;
        }
        // (three stages: both pyramids are complete since the previous tick)
        for (int k = 0; k < L; ++k) {
            // (both frames of a pair come through the same API with the same pitch; a borrowed level 0 uses the caller's)
            if (s->p.local_corner) {
                // (a borrowed level 0 is the whole frame: the chain may read all of it, and the repair rebuilds from it)
                const bool whole0 = k == 0 && s->p.borrow_frames && !s->p.frames_partial; // (partial frames: the patch's extent only)
                const int rows_k = whole0 ? s->h[0] : chain_extent(k, s->ph[k]);
                ofx_geom pg{s->w[k], s->h[k], pitch_of(pc, k, true), 0, rows_k, 0, rows_k};
                C.level[k] = ofx_lk_desc{patch_of(pc - 1, k), patch_of(pc, k), pg, nullptr, 0, nullptr, 0, s->p.min_det};
                C.cols[k] = whole0 ? 0 : chain_extent(k, s->pw[k]);
            } else {
                ofx_geom cg = level_geom(s, k, 0, s->h[k]);
                cg.pitch = pitch_of(pc, k, false);
                C.level[k] = ofx_lk_desc{plane_of(pc - 1, k), plane_of(pc, k), cg, nullptr, 0, nullptr, 0, s->p.min_det};
            }
        }
        C.d_pair_status = s->pair_status + (pc % slots);
        if (s->p.local_corner) {
            C.d_status = s->corner_status;
            if (s->p.sharded) shard_reach(s, C.shard_rows);
            if (s->repair) {
                C.patch_w = s->pw[0];
                C.patch_h = s->ph[0];
                for (int k = 0; k < L; ++k) {
                    C.patch_pitch[k] = s->ppitch[k];
                    C.d_patch_reloc[k] = s->preloc[slot_i][k];
                }
            }
        }
    }
    long newest = -1;
    static thread_local ofx_shift_desc sd0[OFX_MAX_LK_ITEMS];
    int ns0 = 0;
    for (long pl = f0 - D * B; pl <= f0 - D * B + B - 1; ++pl) { // LK(pair pl), reading next through the shift vectors the previous tick wrote
        if (pl < 1 || pl > last_frame) continue;
        const int b = (int)(pl % B);
        float *const *fl = s->flowset[b];
        if (s->fused_iters)
            OFX_REQUIRE(!s->p.borrow_frames || (pitch_of(pl, 0, false) == s->pitch[0] && pitch_of(pl - 1, 0, false) == s->pitch[0]),
                        "ofx_session_stream_submit: with refinement iterations borrowed frames need a row pitch of %d bytes (the "
                        "width rounded up to 64), got %d", s->pitch[0], pitch_of(pl, 0, false));
        for (int k = L - 1; k >= 0; --k) {
            ofx_geom lg = level_geom(s, k, s->fl0[k], s->fl1[k]); // (the own rows, unless iterations follow on a shard)
            lg.pitch = pitch_of(pl, k, false);
            ofx_lk_desc &d = g.lk[g.n_lk++];
            d = ofx_lk_desc{plane_of(pl - 1, k), plane_of(pl, k), lg, fl[k], s->fl0[k], k == L - 1 ? nullptr : uvslot(pl) + 2 * k, 0, s->p.min_det};
            if (s->fused_iters) {
                // refinement iterations follow (lk_body_warp.h): the LK stage is iteration 1 of the pair and also writes the warped
                // image of iteration 2, so the globally shifted next image (the warp's source) is made BEFORE the tick -- its
                // vectors are a tick old -- and the LK stage reads it as it is instead of shifting on the fly
                const uint8_t *src = d.d_next;
                if (k != L - 1) {
                    sd0[ns0++] = ofx_shift_desc{d.d_next, s->itsh[b][0][k], level_geom(s, k, s->buf0[k], s->buf1[k]), uvslot(pl) + 2 * k};
                    src = s->itsh[b][0][k];
                }
                d.d_next = src, d.d_uv = nullptr;
                d.d_warp_src = src, d.d_warp_out = s->itsh[b][1][k], d.warp_scale = OFX_ITER_SCALE;
                if (s->p.sharded) d.d_warp_status = s->corner_status, d.warp_status_bit = 16 + k; // (a tap row beyond the halo rows)
            }
        }
        newest = pl;
    }
    if (ns0) OFX_TRY(timed_launch(s, OFX_TIME_SHIFT, stream, [&] { return ofx_shift_levels(sd0, ns0, stream); }));
    *completed_pair = -1;
    if (newest > s->reported) {
        *completed_pair = (int)newest;
        s->reported = newest;
        for (int k = 0; k < L; ++k) s->flow[k] = s->flowset[newest % B][k];
    }
    bool time_it = g.n_lk > 0;
    g.deep_fetch = s->p.deep_fetch; // (ofx_params.deep_fetch: where the caller's frames come from)
#ifdef OFX_EXPERIMENTS
    // stage ablation for timing experiments (tools/stream_timeline.py): the flows reported complete are then NOT computed, so
    // the knob only exists in builds made with -DOFX_EXPERIMENTS (OFX_BUILD_DEFS)
    static const int skip = [] { const char *e = getenv("OFX_STREAM_SKIP"); return e ? atoi(e) : 0; }();
    if (skip & 1) g.n_pyr = 0;
    if (skip & 2) g.n_corner = 0;
    if (skip & 8) g.n_lk = 0;
    time_it = time_it || (skip & 8);
#endif
    if (time_it)
        OFX_TRY(timed_launch(s, OFX_TIME_STREAM, stream, [&] { return ofx_stream_launch(&g, s->p.window, s->p.mode, stream); }));
    else
        OFX_TRY(ofx_stream_launch(&g, s->p.window, s->p.mode, stream));
    // Extension (lk_iter): the tick's LK stage was iteration 1 of its pairs.  Every further iteration is
    // one warp launch and one accumulating LK launch over ALL levels of ALL those pairs (B x levels items: the strips are B
    // times as tall as in the pair-at-a-time path), after one launch that materialises the globally shifted next images the
    // warp reads.  Same arithmetic, same bits as ofx_session_run_flow with iters > 1.
    if (s->p.iters > 1 && newest >= 1) {
        static thread_local ofx_shift_desc sd[OFX_MAX_LK_ITEMS];
        static thread_local ofx_warp_desc wd[OFX_MAX_LK_ITEMS];
        static thread_local ofx_lk_desc ld[OFX_MAX_LK_ITEMS];
        const int reach = s->p.window / 2 + 1;
        auto clip = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
        for (int it = 1; it < s->p.iters; ++it) { // it = iterations done so far; this pass computes iteration it + 1
            // fused (lk_body_warp.h): every launch but the last also writes the warped images of the pass after it, into the other of
            // the flow set's two warped planes (the tick's LK stage wrote those of this loop's first pass): no warp launch
            const bool fused = s->fused_iters, need_warp = !fused, wout = fused && it + 1 < s->p.iters;
            const int wi = fused ? 1 + ((it - 1) & 1) : 1, wo = 1 + (it & 1);
            int ns = 0, nw = 0;
            for (long pl = f0 - D * B; pl <= f0 - D * B + B - 1; ++pl) {
                if (pl < 1 || pl > last_frame) continue;
                const int b = (int)(pl % B);
                // (a borrowed level 0 is read in place here too; the launches below address a level's planes -- the caller's
                // frame, the shifted and the warped image -- with ONE pitch, so borrowed frames must have the session's)
                OFX_REQUIRE(!s->p.borrow_frames || (pitch_of(pl, 0, false) == s->pitch[0] && pitch_of(pl - 1, 0, false) == s->pitch[0]),
                            "ofx_session_stream_submit: with refinement iterations borrowed frames need a row pitch of %d bytes (the "
                            "width rounded up to 64), got %d", s->pitch[0], pitch_of(pl, 0, false));
                for (int k = L - 1; k >= 0; --k) {
                    // rows of this iteration on a shard: the own rows + (radius + 1) * (iters - 1 - it) either side, so that the
                    // next warp finds the flow of every row its LK touches (whole levels: everything)
                    const int ext = s->p.sharded ? reach * (s->p.iters - 1 - it) : 0;
                    const int a = clip(s->own0[k] - ext, 0, s->h[k]), e = clip(s->own1[k] + ext, 0, s->h[k]);
                    const int wa = clip(a - reach, s->buf0[k], s->buf1[k]), we = clip(e + reach, s->buf0[k], s->buf1[k]);
                    const uint8_t *next_k = plane_of(pl, k);
                    const uint8_t *src = next_k;
                    if (k != L - 1) {
                        // the globally shifted next image, every row the buffers hold (once per pair, before iteration 2)
                        if (it == 1 && !fused) sd[ns++] = ofx_shift_desc{next_k, s->itsh[b][0][k], level_geom(s, k, s->buf0[k], s->buf1[k]), uvslot(pl) + 2 * k};
                        src = s->itsh[b][0][k];
                    }
                    wd[nw] = ofx_warp_desc{src, s->itsh[b][wi][k], level_geom(s, k, wa, we), s->flowset[b][k], s->fl0[k], OFX_ITER_SCALE,
                                           s->p.sharded ? s->corner_status : nullptr, 16 + k};
                    ld[nw] = ofx_lk_desc{plane_of(pl - 1, k), s->itsh[b][wi][k], level_geom(s, k, a, e), s->flowset[b][k], s->fl0[k], nullptr, 1, s->p.min_det};
                    if (wout) {
                        ld[nw].d_warp_src = src, ld[nw].d_warp_out = s->itsh[b][wo][k], ld[nw].warp_scale = OFX_ITER_SCALE;
                        if (s->p.sharded) ld[nw].d_warp_status = s->corner_status, ld[nw].warp_status_bit = 16 + k;
                    }
                    ++nw;
                }
            }
            if (ns) OFX_TRY(timed_launch(s, OFX_TIME_SHIFT, stream, [&] { return ofx_shift_levels(sd, ns, stream); }));
            if (need_warp) OFX_TRY(timed_launch(s, OFX_TIME_WARP, stream, [&] { return ofx_warp_levels(wd, nw, stream); }));
            OFX_TRY(timed_launch(s, wout ? OFX_TIME_LK_ACC_WARP : OFX_TIME_LK_ACC, stream, [&] { return ofx_lk_levels(ld, nw, s->p.window, s->p.mode, stream); }));
        }
    }
    s->stream_n = f0 + B;
    return OFX_OK;
}

extern "C" int ofx_session_stream_begin(ofx_session *s)
{
    OFX_REQUIRE(s, "ofx_session_stream_begin: null session");
    OFX_REQUIRE(!s->p.sharded || s->p.local_corner,
                "ofx_session_stream_begin: on a sharded session the stream pipeline needs local_corner (the corner flows "
                "computed from each frame's top-left patch); otherwise drive the staged API");
    OFX_REQUIRE(s->p.levels >= 2 && s->p.levels - 1 <= 6, "ofx_session_stream_begin: %d levels unsupported (2..7)", s->p.levels);
    // staging work of the pair-at-a-time pipelined path may still be in flight on the session's own stream; the stream
    // pipeline is about to reuse the same image sets from the caller's stream
    if (s->aux) OFX_HIP(hipStreamSynchronize(s->aux));
    for (bool &b : s->set_busy) b = false;
    s->stream_n = 0;
    s->stream_frames = -1;
    s->n_held = 0;
    s->reported = 0;
    s->corner_newest = 0;
    s->have_prev = s->have_next = s->staged = false;
    s->corner_done = false;
    s->pset_img[0] = s->pset_img[1] = -1;
    for (int k = 0; k < s->p.levels; ++k) s->flow[k] = s->flowset[0][k];
    return OFX_OK;
}

// Submit the next frame of the stream.  *completed_pair (may be NULL) receives the highest pair (frame p-1 -> frame p,
// frames counted from 0) whose flow is complete after this call in `stream` order, or -1 when the call completed none.
extern "C" int ofx_session_stream_submit(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream, int *completed_pair)
{
    OFX_REQUIRE(s && d_gray1, "ofx_session_stream_submit: null argument");
    if (s->stream_n < 0) {
        ofx_set_error("ofx_session_stream_submit: call ofx_session_stream_begin first");
        return OFX_E_STATE;
    }
    OFX_REQUIRE(s->stream_frames < 0, "ofx_session_stream_submit: the stream is being drained");
    int dummy = -1;
    if (!completed_pair) completed_pair = &dummy;
    const int B = stream_batch_of(s);
    if (s->n_held + 1 < B) { // the tick is not full yet: remember the frame
        s->held_frame[s->n_held] = d_gray1;
        s->held_pitch[s->n_held] = pitch;
        ++s->n_held;
        *completed_pair = -1;
        return OFX_OK;
    }
    const uint8_t *fr[kMaxBatch];
    int pt[kMaxBatch];
    for (int i = 0; i < s->n_held; ++i) fr[i] = s->held_frame[i], pt[i] = s->held_pitch[i];
    fr[s->n_held] = d_gray1;
    pt[s->n_held] = pitch;
    const int n = s->n_held + 1;
    s->n_held = 0;
    return stream_tick(s, fr, pt, n, stream, completed_pair);
}

extern "C" int ofx_session_stream_submit_frames(ofx_session *s, const uint8_t *const *d_gray1, const int *pitches, int pitch0, int n,
                                                void *stream, int *completed_pair)
{
    OFX_REQUIRE(s && d_gray1 && n >= 1, "ofx_session_stream_submit_frames: bad arguments");
    int newest = -1;
    for (int i = 0; i < n; ++i) {
        int done = -1;
        OFX_TRY(ofx_session_stream_submit(s, d_gray1[i], pitches ? pitches[i] : pitch0, stream, &done));
        newest = done > newest ? done : newest;
    }
    if (completed_pair) *completed_pair = newest;
    return OFX_OK;
}

// Run one more tick without a new frame (frames still waiting for their tick to fill go out with it); call until it
// reports -2 in *completed_pair (pipeline empty).  Two ticks drain a full pipeline.
extern "C" int ofx_session_stream_drain(ofx_session *s, void *stream, int *completed_pair)
{
    OFX_REQUIRE(s && completed_pair, "ofx_session_stream_drain: null argument");
    if (s->stream_n < 0) {
        ofx_set_error("ofx_session_stream_drain: not streaming");
        return OFX_E_STATE;
    }
    const uint8_t *fr[kMaxBatch];
    int pt[kMaxBatch];
    const int n = s->n_held;
    for (int i = 0; i < n; ++i) fr[i] = s->held_frame[i], pt[i] = s->held_pitch[i];
    s->n_held = 0;
    if (s->stream_frames < 0) s->stream_frames = s->stream_n + n; // number of frames the stream received
    if (n == 0 && s->reported >= s->stream_frames - 1) { // every pair (the last one is stream_frames - 1) has been reported
        *completed_pair = -2;
        s->stream_n = -1;
        s->stream_frames = -1;
        return OFX_OK;
    }
    return stream_tick(s, fr, pt, n, stream, completed_pair);
}

extern "C" int ofx_session_flow_of(ofx_session *s, int pair, int level, float **d_ptr, int *row0, int *rows)
{
    OFX_REQUIRE(s && level >= 0 && level < s->p.levels, "ofx_session_flow_of: bad arguments");
    const int B = stream_batch_of(s);
    OFX_REQUIRE(pair >= 1 && pair <= s->reported && pair > s->reported - B,
                "ofx_session_flow_of: pair %d is not among the newest %d completed pairs (newest: %ld)", pair, B, s->reported);
    if (d_ptr) *d_ptr = s->flowset[pair % B][level] + s->flow_own_offset(level);
    if (row0) *row0 = s->own0[level];
    if (rows) *rows = s->own1[level] - s->own0[level];
    return OFX_OK;
}

// gpu::calc_opt_flow (OptFlowGpu.cuh:33, OptFlowGpu.cu:1909-1979) with host pointers: upload both images and the
// two floats of every coarser flow level that the shift reads, run one level on the device, download its flow.
extern "C" int ofx_calc_opt_flow_host(const uint8_t *h_prev3, const uint8_t *h_next3, int w, int h, float **h_flow_pyr, int level,
                                      int max_level, int window, int mode)
{
    OFX_REQUIRE(h_prev3 && h_next3 && h_flow_pyr && w > 0 && h > 0, "ofx_calc_opt_flow_host: bad arguments");
    OFX_REQUIRE(max_level >= 1 && max_level <= OFX_MAX_LEVELS && level >= 0 && level < max_level,
                "ofx_calc_opt_flow_host: bad level %d of %d", level, max_level);
    OFX_REQUIRE(h_flow_pyr[level] != nullptr, "ofx_calc_opt_flow_host: flow level %d is null", level);
    const size_t n = (size_t)w * (size_t)h;
    const int pitch = (int)align_up((size_t)w, 64);
    const size_t plane = (size_t)pitch * (size_t)h + 64;
    // buffers come from the calling thread's cached arena (compat_scratch.h): no allocation per call in a frame loop
    ofx_compat::Scratch sc;
    uint8_t *d_p1 = sc.alloc<uint8_t>(plane), *d_n1 = sc.alloc<uint8_t>(plane), *d_s1 = sc.alloc<uint8_t>(plane);
    float *d_flow = sc.alloc<float>(2 * n);
    float *d_coarse = sc.alloc<float>(2 * OFX_MAX_LEVELS + 2); // 2 floats per level, then the shift vector
    if (!sc.ok()) return sc.rc();
    float *d_uv = d_coarse + 2 * OFX_MAX_LEVELS;
    // the reference reads channel 0 only (OptFlowGpu.cu:1079): it is picked out while the images are staged, a third of the bytes
    OFX_TRY(ofx_compat::stage_h2d_ch0(d_p1, pitch, h_prev3, w, h));
    OFX_TRY(ofx_compat::stage_h2d_ch0(d_n1, pitch, h_next3, w, h));
    ofx_geom g{w, h, pitch, 0, h, 0, h};
    const uint8_t *d_next = d_n1;
    if (level != max_level - 1) {
        float coarse[2 * OFX_MAX_LEVELS] = {};
        const float *lv[OFX_MAX_LEVELS] = {};
        for (int k = level + 1; k < max_level; ++k) {
            OFX_REQUIRE(h_flow_pyr[k] != nullptr, "ofx_calc_opt_flow_host: flow level %d is null", k);
            coarse[2 * k] = h_flow_pyr[k][0];
            coarse[2 * k + 1] = h_flow_pyr[k][1];
            lv[k] = d_coarse + 2 * k;
        }
        OFX_HIP(hipMemcpy(d_coarse, coarse, sizeof coarse, hipMemcpyHostToDevice));
        OFX_TRY(ofx_shift_vector(lv, level, max_level, d_uv, nullptr));
        // (ofx_shift_1ch writes every pixel: shifted byte, own byte or the zero of the reference's fresh scratch pages)
        OFX_TRY(ofx_shift_1ch(d_n1, d_s1, &g, d_uv, nullptr));
        d_next = d_s1;
    }
    OFX_TRY(ofx_lk_level(d_p1, d_next, &g, window, mode, d_flow, 0, nullptr));
    sc.download(h_flow_pyr[level], d_flow, 2 * n);
    return sc.rc();
}


// main.cu:138-147 with host pointers: the dense field at `level` that visualizeFlowField samples for its arrows --
// sum over k >= level of 2^(k-level) * flow_k(y >> (k-level), x >> (k-level)) -- composed on the device.
extern "C" int ofx_compose_flow_host(float *const *h_flow_pyr, int w, int h, int levels, int level, float *h_dst)
{
    OFX_REQUIRE(h_flow_pyr && h_dst && w > 0 && h > 0, "ofx_compose_flow_host: bad arguments");
    OFX_REQUIRE(levels >= 1 && levels <= OFX_MAX_LEVELS && level >= 0 && level < levels, "ofx_compose_flow_host: bad level");
    // level k is uploaded as (w >> s) x (h >> s), s = k - level, and the kernel reads its row i >> s for i < h: with h not a
    // multiple of 2^s that is one row past the upload (the reference has the same overrun, on host memory, main.cu:138-147)
    OFX_REQUIRE(w % (1 << (levels - 1 - level)) == 0 && h % (1 << (levels - 1 - level)) == 0,
                "ofx_compose_flow_host: %dx%d is not a multiple of %d (the coarsest level's scale)", w, h, 1 << (levels - 1 - level));
    ofx_compat::Scratch sc;
    const float *d_lv[OFX_MAX_LEVELS] = {};
    for (int k = level; k < levels; ++k) {
        OFX_REQUIRE(h_flow_pyr[k] != nullptr, "ofx_compose_flow_host: flow level %d is null", k);
        // level k of a pyramid whose level `level` is w x h
        d_lv[k] = sc.upload(h_flow_pyr[k], 2 * (size_t)(w >> (k - level)) * (size_t)(h >> (k - level)));
    }
    float *d_dst = sc.alloc<float>(2 * (size_t)w * (size_t)h);
    if (sc.ok()) sc.run(ofx_compose_flow(d_lv, w, h, levels, level, d_dst, nullptr));
    sc.download(h_dst, d_dst, 2 * (size_t)w * (size_t)h);
    return sc.rc();
}
