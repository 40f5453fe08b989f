// The LK march on buffer resources: the scalar diet of round 3.  Included by lk_body.h (which holds the shared pieces: packed
// rows, derivatives, accumulation, box sums, the LDS exchange) -- not a stand-alone header.
//
// profiles/r03_ablation.txt: adding 40 scalar instructions to a row step lengthens the stream launch by 8.5 us -- a SIMD pays
// ~1.35 cycles for every scalar instruction of its waves, and lk_wave_impl spends ~220 of them per step (a fifth of the launch),
// three quarters on row addresses (a 64-bit multiply-add per load), on wave-uniform tests with their branches ("is this row
// inside the image / the strip / the buffer") and on exec masks around the stores.  This form of the march hands that work to
// the memory pipeline's own range check:
//   * the planes and the flow are addressed through BUFFER RESOURCES (base + extent in four SGPRs): a load is
//     buffer_load_dword v, lane offset, rsrc, row offset -- the row offset is one 32-bit scalar, no 64-bit arithmetic;
//   * a row that does not exist gets the offset kOob (0x80000000): the load then lies outside the resource's extent and
//     returns 0, exactly what the zero border wants -- no branch; the rows of `next` come from a 64-entry table of offsets
//     (the shifted row, or kOob), so the reference's row map costs one v_readlane;
//   * a lane that must not store gets the lane offset kOob: the store is dropped by the same check -- no exec mask.
// tools/ubench/buffer_ops.hip checks these behaviours on the device (unaligned dword, marker row, dropped stores).
// The arithmetic, the LDS exchange, the packed registers and the order of loads and stores are those of lk_wave_impl; results
// are bit-identical (the same tests run on both; OFX_LK_BUFFER_PATH=0 builds the old form everywhere).  The old form stays
// for the inspection variant (SUMS) and for accumulating launches (MAY_ACC); the host keeps levels of 2 GB and more out of
// launches that use this one (lk_launch.h).
#pragma once

namespace ofx_dev {

#ifndef OFX_LK_BUFFER_PATH
#define OFX_LK_BUFFER_PATH 1
#endif
#ifndef OFX_LK_HBOX_LOCKSTEP
#define OFX_LK_HBOX_LOCKSTEP 1
#endif
constexpr int kOob = (int)0x80000000;
#if defined(OFX_LK_STORE_AUX) // (experiments: other cache-policy bits of the flow stores -- bit 0 sc0, bit 1 nt, bit 4 sc1)
#elif defined(OFX_X_TINYSTORE)
#define OFX_LK_STORE_AUX 0 // (cached)
#else
#define OFX_LK_STORE_AUX (OFX_LK_NT_STORES ? 2 : 0) // nt: streaming stores (see "the flow stores" in lk_body.h)
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00027000);
}

// DMA (OFX_LK_DMA_ROWS): the rows of a step are fetched TWO steps ahead, straight into LDS (buffer_load ... lds: no VGPR holds
// them while they are in flight), six per step -- prev / shifted next / own next, for the entering and the leaving window -- into
// one of two sets of six 256-byte rows behind the wave's exchange row.  The point is the wait: gfx9 counts loads and stores in
// ONE counter, so the march's "wait for the rows fetched a step ahead" was also a wait for the two flow stores of the step
// before -- 28 us of a 242 us launch (profiles/r03_ablation.txt: the launch without its stores).  Loads return in order among
// themselves, so with the next step's six loads already issued behind them, s_waitcnt vmcnt(6) means "this step's rows have
// arrived" whatever the stores are doing: if one of this step's rows were still out, all six younger loads would be too, and
// the count could not be six.  Stores may now take up to two steps to complete before a wave waits for them.
#ifndef OFX_LK_DMA_ROWS
#define OFX_LK_DMA_ROWS 1
#endif
// Deferred flow store (experiment; -DOFX_LK_DEFER_STORE=1): a step's output row stays in a SECOND exchange row
// of LDS and is stored one step later, in the middle of the next step -- after that step's row loads have been issued, so that no
// load queues behind the two 1 KB stores of the row before it.  Measured: profiles/r04_ablation.txt.
// 0: never; 1: every launch of the buffer march; 2 (shipped): the accumulating launches of lk_iter only (ITER 1, 2, 4), whose step
// holds the most memory operations (row loads, the old flow, eight warp taps, three stores): between +2.3 % and 0 at 4K / 5
// iterations (two batches, different boxes); the reference-defined tick measured 0 % (ring in the Infinity Cache) to -3 % (frames
// from HBM) with it and keeps the plain order.
#ifndef OFX_LK_DEFER_STORE
#define OFX_LK_DEFER_STORE 2
#endif
// The leaving window's rows out of an LDS ring (round 4).  In the accumulating launches 16 B/px of flow stream through the XCD's 4 MB
// L2 between a row's first read (entering the vertical window) and its second (leaving it, 2R + 1 steps later): the second read
// misses and goes to the fabric -- 391 MB of the launch's 1 600 MB of fabric reads (profiles/r04_ablation.txt batch 6).  A lane
// keeps the finished dwords (prev, next) of the last 2R + 1 entering rows in 8 bytes per row of LDS of its own and takes the leaving
// row from there: no load, no fabric traffic, no finish.  (R + 1) KB per wave; accumulating launches without deep fetch,
// R <= 8.  OFX_LK_OUT_RING=0: both rows from memory, as the tick does (whose L2 holds them: it fetches 1.10 x its algorithmic reads).
// MEASURED (profiles/r04_ablation.txt batch 6): fabric reads of the launch 1 601 -> 1 386 MB (traffic 1.35 x -> 1.23 x algorithmic), launch
// 462 vs 459 us, 1080p and the pair-at-a-time path 1.6 % slower (the packed selectors it needs to stay within 128 VGPRs cost ~18 vector
// instructions per row step): the launch is not bound by its fabric traffic either.  Bit-exact; OFF by default.
#ifndef OFX_LK_OUT_RING
#define OFX_LK_OUT_RING 0
#endif
template <int R, int ITER, bool DMA>
constexpr bool lk_out_ring()
{
    return OFX_LK_OUT_RING && (ITER == 1 || ITER == 2 || ITER == 4) && !DMA && R <= 8;
}
template <int R, int ITER, bool DMA>
constexpr int lk_ring_bytes()
{
    return lk_out_ring<R, ITER, DMA>() ? (R + 1) * 1024 : 0; // (two rows per 16-byte lane slot: the address is 16 * lane + a scalar)
}
#ifndef OFX_LK_ACC_LOAD_AUX
#define OFX_LK_ACC_LOAD_AUX 0
#endif
// 1: the accumulating launches read the old flow in the exchanged (gap-free) layout and put it back in place through LDS.  Built to
// test whether the launch's fabric reads (FETCH_SIZE: 1.65 x its algorithmic reads) come from the two half-used 16-byte loads per
// lane: they do not (FETCH_SIZE 811 vs 781 GiB-units, launch 452-455 vs 439-457 us: profiles/r04_ablation.txt batch 5).  Off.
#ifndef OFX_LK_ACC_XLOAD
#define OFX_LK_ACC_XLOAD 0
#endif
constexpr int kLkXRows = OFX_LK_DEFER_STORE ? 2 : 1;            // exchange rows per wave
constexpr int kLkWaveLdsX = kLkXRows * kLkWaveLds;              // what a wave of the buffer march owns without the deep fetch
constexpr int kLkDmaRowBytes = 256, kLkDmaRows = 6, kLkDmaSetBytes = kLkDmaRows * kLkDmaRowBytes;
constexpr int kLkWaveLdsDma = kLkWaveLdsX + 2 * kLkDmaSetBytes; // the exchange row(s), then two sets of fetched rows

// ITER (refinement iterations of lk_iter): 0 = flow = result (the reference's level); 1 = flow += result,
// the row's old flow fetched through the flow's own resource with the step's rows; 2 = the same, and the march also writes the
// warped image the NEXT iteration reads (lk_body_warp.h): the row's new flow is in registers after the add, the warp's first stage
// runs there and issues its tap loads, the second stage and the row's store follow one step later (whole levels only: lk_level.hip);
// 3 = iteration 1 of a pair that has more: flow = result, and the warped image of iteration 2;
// 4 / 5 = 2 / 3 on the row window of a shard: the planes hold rows [row0, row_end) only, a tap row outside them is reported.
template <int R, int MODE, bool FAST, bool INTERIOR, bool DMA, int ITER = 0>
__device__ __forceinline__ void lk_wave_buf(const LkTable &T, int wave, int lane, uint8_t *xlds)
{
    using G = TileGeom<R>;
    constexpr int NS = 2 * R + 1;
    constexpr bool ACC = ITER == 1 || ITER == 2 || ITER == 4, WOUT = ITER >= 2, ROWWIN = ITER >= 4;
    constexpr bool DEFER = OFX_LK_DEFER_STORE == 1 || (OFX_LK_DEFER_STORE == 2 && ACC);
    constexpr bool RING = lk_out_ring<R, ITER, DMA>(); // the leaving rows come out of the wave's LDS ring (xlds + kLkWaveLdsX)

    if (wave >= T.first_block[T.n]) return;
    int level = 0, hi = T.n;
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (wave >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    LkArgs A = T.lv[level];
    pin_scalar(A.w);
    pin_scalar(A.h);
    pin_scalar(A.pitch);
    pin_scalar(A.row0);
    pin_scalar(A.row_end);
    pin_scalar(A.flow_row0);
    pin_scalar(A.min_det);
    const SolveOpts sopt{A.min_det};
    const int block = wave - T.first_block[level];
    const int tile = block % A.tiles_x;
    const int strip = block / A.tiles_x;
    const int cb = tile * G::OUT_W - G::LO_LANE * 4 + 4 * lane; // first of this lane's 4 image columns
    const int ys = A.out_y0 + strip * A.strip_h;
    const int ye = min(ys + A.strip_h, A.out_y1);

    // the planes (both have the level's geometry) and the flow as buffer resources
    const int plane_bytes = (A.row_end - A.row0) * A.pitch;
    const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(A.prev, plane_bytes), rs_next = make_rsrc(A.next, plane_bytes);
    const __amdgpu_buffer_rsrc_t rs_flow = make_rsrc(A.flow, (A.out_y1 - A.flow_row0) * A.w * 8);

    // column validity: bytes outside [0,w) read as zero, derivatives there are zero
    const bool ld_ok = INTERIOR || (cb >= 0 && cb < A.w);
    uint32_t bmask = INTERIOR ? 0xffffffffu : 0u;
    if constexpr (!INTERIOR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = (cb + j) >= 0 && (cb + j) < A.w;
            bmask |= in ? (0xffu << (8 * j)) : 0u;
        }
    }
    uint32_t col_off = ld_ok ? (uint32_t)cb : 0u;
    // the exchanged layout of an output row (see "the flow stores"): lane l stores chunks l and l + 64 of the row
    const int x0 = tile * G::OUT_W;
    const int nv = min(x0 + G::OUT_W, A.w) - x0;
    uint32_t l16 = 16u * (uint32_t)lane;
    const int lim = 8 * nv - 16, c16 = 16 * lane;
    const bool st_lo4 = c16 <= lim, st_lo2 = c16 == lim + 8, st_hi4 = c16 <= lim - 1024, st_hi2 = c16 == lim + 8 - 1024;
    // lane offsets of the two stores; a lane that has nothing to store points outside the resource (the store is dropped)
    uint32_t vo_lo = st_lo4 ? l16 : (uint32_t)kOob, vo_hi = st_hi4 ? l16 + 1024u : (uint32_t)kOob;
    // (a full tile: every lane's first chunk exists -- interior tiles use l16 itself, the register the exchange keeps anyway)
    const bool ragged = __any(st_lo2 || st_hi2) != 0; // a level of odd width ends inside a chunk: that lane stores one pixel
    const lds_ptr xl_w = (lds_ptr)xlds + 32 * lane;
    const lds_ptr xl_base = (lds_ptr)xlds + 32 * G::LO_LANE;
    // ACC: where this lane's pixels lie in a flow row (the byte offset of the first; pixels outside the image get kOob
    // from their column mask when the row is fetched: they read 0 and are never stored)
    [[maybe_unused]] uint32_t nat_off = (uint32_t)cb * 8u;
    // WOUT: the warp source and the warped image as resources; which of this lane's pixels are output pixels of the tile
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_wsrc = rs_prev, rs_wout = rs_prev;
    [[maybe_unused]] uint32_t wvo = (uint32_t)kOob, wmiss = 0u;
    [[maybe_unused]] int wnpx = 0;
    [[maybe_unused]] WarpRowState WM;
    if constexpr (WOUT) {
        pin_scalar(A.warp_scale);
        rs_wsrc = make_rsrc(A.warp_src, plane_bytes + (OFX_WARP_LEAN ? 3 : 0)); // (lk_body_warp.h: a tap dword may start in the plane's last three bytes)
        rs_wout = make_rsrc(A.warp_out, plane_bytes);
        const bool out_lane = lane >= G::LO_LANE && lane <= G::HI_LANE && cb < A.w;
        wvo = out_lane ? (uint32_t)cb : (uint32_t)kOob; // (a level whose width is no multiple of 4 ends inside the dword: the rest is row padding)
        wnpx = out_lane ? min(4, A.w - cb) : 0;
        warp_row_clear(WM);
    }

    const int y_lim = min(min(ye + R + 1, A.h), A.row_end);
    const int y_min = max(0, A.row0);
    const int y_first = ys - R; // first derivative row this strip needs
    // byte offset of image row y in the planes, kOob where the row is the zero border or not needed: one unsigned range test
    const int span_in = max(y_lim - y_min, 0);
    const int y_min_out = max(y_min, y_first - 1), span_out = max(y_lim - y_min_out, 0); // (rows of the leaving window before y_first - 1 are never used)
    auto row_off = [&](int y) -> int { return (uint32_t)(y - y_min) < (uint32_t)span_in ? (y - A.row0) * A.pitch : kOob; };
    auto row_off_out = [&](int y) -> int { return (uint32_t)(y - y_min_out) < (uint32_t)span_out ? (y - A.row0) * A.pitch : kOob; };

    // ---- fused shift (see lk_wave_impl): per lane the base column of the shifted dword and the byte selectors
    const float su = A.uv ? A.uv[0] : 0.0f, sv = A.uv ? A.uv[1] : 0.0f;
    uint32_t nb_off = 0u, sel = 0u, own_sel = 0u;
    bool all_in;
    {
        int n[4], nb = 0x7fffffff;
        bool in[4], lane_all_in = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = cb + j;
            const float tx = (float)x + su;
            in[j] = x >= 0 && x < A.w && tx > -1.0f && tx < (float)A.w;
            n[j] = in[j] ? (int)tx : 0;
            if (in[j]) nb = min(nb, n[j]);
        }
        nb = max(0, min(nb == 0x7fffffff ? 0 : nb, A.pitch - 4)); // the dword stays inside the row pitch
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = cb + j;
            const uint32_t sj = !(x >= 0 && x < A.w) ? 0x0cu : (in[j] ? (uint32_t)(n[j] - nb) : (uint32_t)(4 + j));
            sel |= sj << (8 * j);
            own_sel |= (!(x >= 0 && x < A.w) ? 0x0cu : (uint32_t)(4 + j)) << (8 * j);
            if (x >= 0 && x < A.w && !in[j]) lane_all_in = false;
        }
        nb_off = (uint32_t)nb;
        all_in = __all(lane_all_in) != 0;
    }
    const int y_none = (A.h + 2) / 3, y_part = (A.h % 3) ? A.h / 3 : -1;
    // The row map y -> (int)((float)y + v) as a table of BYTE OFFSETS for 64 consecutive rows (lane i: row map_base + i): the
    // offset of the target row in the planes, or kOob when row y is not needed (outside [y_min, y_lim)) or its target is outside
    // the image or the buffer.  A step reads its two entries with v_readlane_b32.
    int map_base = 0, row_tab = kOob;
    auto refresh_map = [&](int y0) {
        map_base = y0;
        const int y = y0 + lane;
        const float ty = (float)y + sv;
        const bool yin = ty > -1.0f && ty < (float)A.h;
        const int ny = yin ? (int)ty : 0;
        const bool ok = y >= y_min && y < y_lim && yin && ny >= A.row0 && ny < A.row_end;
        row_tab = ok ? (ny - A.row0) * A.pitch : kOob;
    };
    struct NextRaw {
        uint32_t own, sh; // the row's own dword / the dword at the shifted position (0 where not needed)
        int miss;         // wave-uniform: all ones when the shifted row does not exist
    };
    // po: row_off / row_off_out of the same row y (the planes share their geometry: the own row of next sits where prev's does)
    auto fetch_next = [&](int y, int po) -> NextRaw {
        NextRaw r;
        const int e = __builtin_amdgcn_readlane(row_tab, y - map_base);
        r.sh = __builtin_amdgcn_raw_buffer_load_b32(rs_next, nb_off, e, 0);
        r.miss = e >> 31;
        asm("" : "=v"(r.own)); // (never selected while every byte comes from the shifted dword: left undefined, not zeroed)
        // an interior tile whose target row exists takes every byte from the shifted dword; otherwise (a column or the row
        // leaves the image) the pixels concerned keep their own byte while 3 * (y * w + x) < w * h, else 0 (OptFlowCPU.cpp:247)
        if (__builtin_expect(!all_in || r.miss != 0, 0)) {
            r.own = __builtin_amdgcn_raw_buffer_load_b32(rs_next, col_off, y < y_none ? po : kOob, 0);
            if (y == y_part) {
                uint32_t km = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (3ll * (cb + j) < (long long)A.w * (A.h % 3)) km |= 0xffu << (8 * j);
                r.own &= km;
            }
        }
        return r;
    };
    // both selectors give 0 for columns outside the image, and own == sh == 0 for rows outside it
    auto finish_next = [&](const NextRaw &r) -> uint32_t {
        // selector = miss ? own_sel : sel, as ONE v_bfi_b32 on the scalar mask (written as bit operations hipcc turns it into a
        // compare, a 64-bit select and a v_cndmask)
        uint32_t sx;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sx) : "s"(r.miss), "v"(own_sel), "v"(sel));
        return __builtin_amdgcn_perm(r.own, r.sh, sx);
    };
    auto fetch_prev = [&](int po) -> uint32_t { return __builtin_amdgcn_raw_buffer_load_b32(rs_prev, col_off, po, 0); };
    auto finish_row = [&](uint32_t raw) -> uint32_t {
        if constexpr (INTERIOR) return raw;
        else return raw & bmask;
    };
    auto load_pair = [&](int y, bool out, uint32_t &p, uint32_t &n) {
        const int po = out ? row_off_out(y) : row_off(y);
        p = finish_row(fetch_prev(po));
        n = finish_next(fetch_next(y, po));
    };

    // ---- DMA form: issue the six rows of a step into LDS set `set`; take them out again a step later
    const uint32_t dma_base = (uint32_t)(uintptr_t)((lds_ptr)xlds + kLkWaveLdsX); // LDS byte address of set 0, row 0 (wave-uniform)
    uint32_t dma_lane = dma_base + 4u * (uint32_t)lane;                           // this lane's dword in set 0, row 0
    auto dma_load = [&](const __amdgpu_buffer_rsrc_t &rs, uint32_t voff, int soff, uint32_t lds_addr) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(uintptr_t)lds_addr, 4, voff, soff, 0, 0);
    };
    // own row of `next` for image row y (po: its offset): needed when a column target or the row target leaves the image
    // Four rows per step when every column target of the wave lies inside the image (all_in: the shifted dwords supply every
    // byte while the shifted ROW exists), six otherwise (the own rows of next as well).  all_in waves meet a row whose target
    // leaves the image only at the image's top or bottom: they fetch its own row when they take it (take_one; an ordinary load,
    // a stall of one memory round trip on those few rows).
    auto issue_rows = [&](int set, int y_in, int y_out) {
        const int po_in = row_off(y_in), po_out = row_off_out(y_out);
        const int e_in = __builtin_amdgcn_readlane(row_tab, y_in - map_base);
        // (a row of the leaving window before y_first - 1 is never used and may lie below the table)
        const int e_out = y_out >= y_first - 1 ? __builtin_amdgcn_readlane(row_tab, y_out - map_base) : kOob;
        const uint32_t a = dma_base + (uint32_t)(set * kLkDmaSetBytes);
        dma_load(rs_prev, col_off, po_in, a);
        dma_load(rs_prev, col_off, po_out, a + 1 * kLkDmaRowBytes);
        dma_load(rs_next, nb_off, e_in, a + 2 * kLkDmaRowBytes);
        dma_load(rs_next, nb_off, e_out, a + 3 * kLkDmaRowBytes);
        if (__builtin_expect(!all_in, 0)) {
            dma_load(rs_next, col_off, y_in < y_none ? po_in : kOob, a + 4 * kLkDmaRowBytes);
            dma_load(rs_next, col_off, y_out < y_none ? po_out : kOob, a + 5 * kLkDmaRowBytes);
        }
    };
    // the row of next for image row y from what arrived (sh: shifted dword; own: own dword, fetched only for !all_in waves)
    auto take_one = [&](int y, bool out, uint32_t sh, uint32_t own) -> uint32_t {
        const int e = (!out || y >= y_first - 1) ? __builtin_amdgcn_readlane(row_tab, y - map_base) : kOob;
        NextRaw r;
        r.sh = sh;
        r.miss = e >> 31;
        r.own = own;
        if (__builtin_expect(r.miss != 0 && all_in, 0)) // the row's target left the image (or the row is not needed: then own is 0 too)
            r.own = __builtin_amdgcn_raw_buffer_load_b32(rs_next, col_off, y < y_none ? (out ? row_off_out(y) : row_off(y)) : kOob, 0);
        if (__builtin_expect(y == y_part && (!all_in || r.miss != 0), 0)) {
            uint32_t km = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (3ll * (cb + j) < (long long)A.w * (A.h % 3)) km |= 0xffu << (8 * j);
            r.own &= km;
        }
        return finish_next(r);
    };

    // (the folded priming and the slot rotation are those of lk_wave_impl)
    constexpr int H = OFX_LK_FOLD_PRIMING ? R - 1 : 0;
    constexpr int PR = 2 * R - H;
    const int y_lo0 = y_first + H;
    const int nsteps = (ye - ys) + PR;
    RowPk<MODE> wp[3];
    const s2 two = pk_two();
    refresh_map(y_first - 1);
    // RING: slot ((r - (y_lo0 - 1)) mod NS) holds image row r; this lane's 8 bytes of it are (prev, next), finished
    // (rows 2j and 2j + 1 share a lane's 16 bytes of KB j, so that the address is the lane's 16 * lane -- a register the exchange keeps
    // anyway -- plus a scalar: a pointer of its own would be the 129th VGPR of a kernel that must fit 128)
    auto ring_at = [&](int slot) -> lds_ptr { return (lds_ptr)xlds + (kLkWaveLdsX + (slot >> 1) * 1024 + (slot & 1) * 8) + lane_off_var(l16); };
    {
        uint32_t pi, ni, po = 0u, no = 0u;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            load_pair(y_lo0 - 1 + t, false, pi, ni);
            if constexpr (H > 0) load_pair(y_first - 1 + t, true, po, no);
            unpack_pk(pi, ni, po, no, wp[t]);
            if constexpr (RING) *(__attribute__((address_space(3))) u32x2 *)(ring_at(t % NS)) = u32x2{pi, ni};
        }
    }
    [[maybe_unused]] int rslot = 3 % NS; // the slot of the row the next step adds (and of the one it takes out of the ring)
    if constexpr (DMA) { // the rows step 0 takes at its end (the b rows of step 1): y_lo0 + 2 and the high stream's
        const int ro0 = (H > 1 && 1 < H) ? y_first + 2 : y_lo0 - NS + 2;
        issue_rows(0, y_lo0 + 2, ro0);
    }
    int vxx[4] = {0, 0, 0, 0}, vyy[4] = {0, 0, 0, 0}, vxy[4] = {0, 0, 0, 0}, vxt[4] = {0, 0, 0, 0}, vyt[4] = {0, 0, 0, 0};
    // byte offset, in the flow, of the tile's first output pixel in the row the next emitting step writes
    // (made scalar by hand, and recomputed per step rather than carried: as a running sum hipcc keeps it in a VGPR and wraps
    // every store in a waterfall loop)
    int fso0 = __builtin_amdgcn_readfirstlane(((ys - A.flow_row0) * A.w + x0) * 8);
    int fstep = A.w * 8;
    pin_scalar(fso0);
    pin_scalar(fstep);

    // the two streaming stores of output row (step s_row) from its chunks xlo / xhi of the exchanged layout
    auto store_row = [&](int s_row, const f32x4 xlo, const f32x4 xhi) {
#if defined(OFX_X_TINYSTORE) // timing experiment: every row lands in the first MB of the flow (L2 hits, no HBM write stream)
        const int fso = __builtin_amdgcn_readfirstlane((fso0 + (s_row - PR) * fstep) & 0xff000);
#else
        const int fso = __builtin_amdgcn_readfirstlane(fso0 + (s_row - PR) * fstep); // this row's offset in the flow
#endif
        // two gap-free streaming stores of 1 KB; the lanes past the tile's end are dropped by the resource's range check
#if defined(OFX_X_NOSTORE) // timing experiment: the row is exchanged but never stored
        asm volatile("" : : "v"(xlo), "v"(xhi), "s"(fso));
#else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xlo), rs_flow, INTERIOR ? lane_off_var(l16) : vo_lo, fso, OFX_LK_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xhi), rs_flow, vo_hi, fso, OFX_LK_STORE_AUX);
#endif
        if (__builtin_expect(ragged, 0)) { // the one lane whose chunk holds a single pixel
            const u32x4 ql = __builtin_bit_cast(u32x4, xlo), qh = __builtin_bit_cast(u32x4, xhi);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{ql.x, ql.y}, rs_flow, st_lo2 ? l16 : (uint32_t)kOob, fso, OFX_LK_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{qh.x, qh.y}, rs_flow, st_hi2 ? l16 + 1024u : (uint32_t)kOob, fso, OFX_LK_STORE_AUX);
        }
    };
    // OFX_LK_DEFER_STORE: the row of step s_row out of its exchange row (s_row & 1), and its stores
    [[maybe_unused]] auto deferred_store = [&](int s_row) {
        const lds_ptr xl_r = xl_base + (s_row & 1) * kLkWaveLds + lane_off_var(l16);
        const f32x4 dlo = *(__attribute__((address_space(3))) f32x4 *)(xl_r);
        const f32x4 dhi = *(__attribute__((address_space(3))) f32x4 *)(xl_r + 1024);
        store_row(s_row, dlo, dhi);
    };

    auto body = [&](auto K, int s) {
        constexpr int k = decltype(K)::value; // s mod 3
        const int yy = y_lo0 + s;             // derivative row entering the window (low halves)
        const int yo = yy - NS;               // derivative row leaving it (high halves, once the folded priming is over)
        const bool folded = H > 0 && s < H;   // high halves: the entering row y_first + s
        const int yh = folded ? y_first + s : yo;

        const int ro = (H > 1 && s + 1 < H) ? y_first + s + 2 : yo + 2; // b row of the high stream's next step
        uint32_t pf_ip = 0u, pf_op = 0u;
        NextRaw pf_in = {0u, 0u, -1}, pf_on = {0u, 0u, -1};
        if constexpr (DMA) {
            // issue the rows of the step AFTER the next one (this step's were issued a step ago and are taken below); the table
            // must still hold this step's rows then: yo + 2 is the lowest of them
            const int ro_next = (H > 1 && s + 2 < H) ? y_first + s + 3 : yo + 3;
            if (yy + 3 - map_base >= 64) refresh_map(min(yo + 2, ro));
            issue_rows((s + 1) & 1, yy + 3, ro_next);
        } else {
            // the loads of the rows the next step adds, finished at the end of this step, before its stores (one vmcnt for both kinds)
            if (yy + 2 - map_base >= 64) refresh_map(yo + 2); // (yo + 2 is the lowest row still to be looked up)
            const int po_in = row_off(yy + 2), po_out = row_off_out(ro);
            pf_ip = fetch_prev(po_in);
            pf_in = fetch_next(yy + 2, po_in);
#ifndef OFX_X_NO_OUTROWS // (diagnostic builds, profiles/r04_ablation.txt batch 6: which loads the launch's fabric reads belong to)
            if (RING && s >= NS - 3) {
                // the leaving row entered NS steps ago (or with the priming rows): its finished dwords wait in the ring
                const u32x2 q = *(const __attribute__((address_space(3))) u32x2 *)(ring_at(rslot));
                pf_op = q.x;
                pf_on.sh = q.y;
            } else {
                pf_op = fetch_prev(po_out);
                // (a row of the leaving window before y_first - 1 is never used and may lie below the table: its pixels are zeros)
                if (ro >= y_first - 1) pf_on = fetch_next(ro, po_out);
            }
#endif
        }
        const bool emit = s >= PR;
        // ACC: the flow this row adds to, as it lies (this lane's 4 pixels), fetched with the step's rows
        [[maybe_unused]] f32x4 old_a, old_b;
        if constexpr (ACC) {
            asm("" : "=v"(old_a), "=v"(old_b));
            if (emit) {
                const int fnat = __builtin_amdgcn_readfirstlane(fso0 + (s - PR) * fstep - x0 * 8); // offset of the row's pixel 0
                // (OFX_LK_ACC_LOAD_AUX: cache-policy bits of the old flow's loads.  2 = nt -- "read once, do not displace the image rows the
                // trailing window re-reads" -- measured 20 % SLOWER at 4K / 5 iterations: 554 vs 457 us, profiles/r04_ablation.txt batch 4)
#ifdef OFX_X_NO_OLDFLOW
                old_a = old_b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (false)
#endif
                if constexpr (INTERIOR && OFX_LK_ACC_XLOAD) {
                    // the old flow in the EXCHANGED layout, as the stores write it: each of the two loads covers 1 KB without gaps
                    // (in place, a lane's two 16-byte halves make every instruction touch all sixteen lines of the row and use half of
                    // each).  Put back in place through the exchange row right before the add.
                    const int fx = __builtin_amdgcn_readfirstlane(fso0 + (s - PR) * fstep);
                    old_a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_flow, lane_off_var(l16), fx, OFX_LK_ACC_LOAD_AUX));
                    old_b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_flow, vo_hi, fx, OFX_LK_ACC_LOAD_AUX));
                } else if constexpr (INTERIOR) {
                    old_a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_flow, nat_off, fnat, OFX_LK_ACC_LOAD_AUX));
                    old_b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_flow, nat_off + 16u, fnat, OFX_LK_ACC_LOAD_AUX));
                } else { // (per pixel: a level of odd width ends inside a 16-byte piece)
                    auto off = [&](int j) { return ((bmask >> (8 * j)) & 1u) ? nat_off + 8u * (uint32_t)j : (uint32_t)kOob; };
                    const u32x2 p0 = __builtin_amdgcn_raw_buffer_load_b64(rs_flow, off(0), fnat, OFX_LK_ACC_LOAD_AUX), p1 = __builtin_amdgcn_raw_buffer_load_b64(rs_flow, off(1), fnat, OFX_LK_ACC_LOAD_AUX);
                    const u32x2 p2 = __builtin_amdgcn_raw_buffer_load_b64(rs_flow, off(2), fnat, OFX_LK_ACC_LOAD_AUX), p3 = __builtin_amdgcn_raw_buffer_load_b64(rs_flow, off(3), fnat, OFX_LK_ACC_LOAD_AUX);
                    old_a = __builtin_bit_cast(f32x4, u32x4{p0.x, p0.y, p1.x, p1.y});
                    old_b = __builtin_bit_cast(f32x4, u32x4{p2.x, p2.y, p3.x, p3.y});
                }
            }
        }

        const uint32_t him = folded ? 0x00010000u : (yo >= y_first ? 0xffff0000u : 0u);
        uint32_t rowm = ((uint32_t)yy < (uint32_t)A.h ? 0x00000001u : 0u) | ((uint32_t)yh < (uint32_t)A.h ? him : 0u);
        // (one scalar multiplier pair for all columns.  readfirstlane, not pin_scalar: hipcc's uniformity analysis does not see
        // that this value -- or the store offset below -- is wave-uniform, and an "s" constraint on it fails to compile)
        if constexpr (INTERIOR) rowm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowm);
        // (edge tiles: a column's mask comes out of the byte mask -- one sign-extending bit-field extract -- instead of living in four
        // registers of its own: the accumulating launches sit on the 128-VGPR line; interior tiles: cm is the constant -1)
        uint32_t mm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mm[j] = INTERIOR ? rowm : ((uint32_t)__builtin_amdgcn_sbfe((int)bmask, 8 * j, 1) & rowm);
        s2 ix[4], iy[4], it[4];
        derivs_pk(wp[k], wp[(k + 1) % 3], wp[(k + 2) % 3], two, ix, iy, it);
        accumulate_pk(ix, iy, it, mm, vxx, vyy, vxy, vxt, vyt);
        if constexpr (DEFER) {
            if (s > PR) deferred_store(s - 1); // the row the step before exchanged: this step's row loads are already on their way
        }
        auto take_rows = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMA) {
                // this step's rows have arrived once at most the younger loads (the next step's four or six) are outstanding
                // (WOUT: an emitting step has also issued the eight tap loads of its row's warp since)
                if (WOUT && emit) {
                    if (__builtin_expect(all_in, 1)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                } else {
                    if (__builtin_expect(all_in, 1)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                }
                const lds_ptr rp = (lds_ptr)(uintptr_t)(dma_lane + (uint32_t)((s & 1) * kLkDmaSetBytes));
                auto row = [&](int i) { return *(const __attribute__((address_space(3))) uint32_t *)(rp + i * kLkDmaRowBytes); };
                const uint32_t a_p = row(0), a_po = row(1), a_n = row(2), a_no = row(3);
                uint32_t o_n, o_no;
                asm("" : "=v"(o_n), "=v"(o_no)); // (never selected while every byte comes from the shifted dwords)
                if (__builtin_expect(!all_in, 0)) o_n = row(4), o_no = row(5);
                unpack_pk(finish_row(a_p), take_one(yy + 2, false, a_n, o_n), finish_row(a_po), take_one(ro, true, a_no, o_no), wp[k]);
            } else {
                if constexpr (RING) {
                    const uint32_t p_in = finish_row(pf_ip), n_in = finish_next(pf_in);
                    uint32_t p_out, n_out;
                    if (s >= NS - 3) p_out = pf_op, n_out = pf_on.sh;
                    else p_out = finish_row(pf_op), n_out = finish_next(pf_on);
                    unpack_pk(p_in, n_in, p_out, n_out, wp[k]);
                    *(__attribute__((address_space(3))) u32x2 *)(ring_at(rslot)) = u32x2{p_in, n_in};
                    rslot = rslot + 1 == NS ? 0 : rslot + 1;
                } else {
                    unpack_pk(finish_row(pf_ip), finish_next(pf_in), finish_row(pf_op), finish_next(pf_on), wp[k]);
                }
            }
            pin_row(wp[k]);
        };

        f32x4 xlo, xhi;
        asm("" : "=v"(xlo), "=v"(xhi));
        if (emit) {
            float uv[8];
#if OFX_LK_HBOX_LOCKSTEP
            // the five box sums stage by stage (hbox4x5): no wait states between the dependent DPP operations of one quantity
            int hb[5][4];
            {
                int va[5][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) va[0][j] = vxx[j], va[1][j] = vyy[j], va[2][j] = vxy[j], va[3][j] = vxt[j], va[4][j] = vyt[j];
                hbox4x5<R>(va, hb);
            }
            const int(&hxx)[4] = hb[0], (&hyy)[4] = hb[1], (&hxy)[4] = hb[2], (&hxt)[4] = hb[3], (&hyt)[4] = hb[4];
#else
            int hxx[4], hyy[4], hxy[4], hxt[4], hyt[4];
            hbox4<R>(vxx, hxx);
            hbox4<R>(vyy, hyy);
            hbox4<R>(vxy, hxy);
            hbox4<R>(vxt, hxt);
            hbox4<R>(vyt, hyt);
#endif
#ifdef OFX_X_NOSOLVE // timing experiment (results wrong by construction)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uv[2 * j] = __int_as_float(hxx[j] ^ hxy[j] ^ hxt[j]);
                uv[2 * j + 1] = __int_as_float(hyy[j] ^ hyt[j]);
            }
#else
            solve_lane<MODE, FAST>(hxx, hyy, hxy, hxt, hyt, sopt, uv);
#endif
            if constexpr (ACC && INTERIOR && OFX_LK_ACC_XLOAD) {
                // chunks l and l + 64 of the row -> this lane's own 32 bytes (the inverse of the exchange in front of the stores; the
                // halo lanes, whose pixels are never stored, get whatever the row held).  A wave's LDS operations execute in order.
                const lds_ptr xr = DEFER ? (lds_ptr)xlds + (s & 1) * kLkWaveLds : (lds_ptr)xlds;
                const lds_ptr xq = xr + 32 * G::LO_LANE + lane_off_var(l16);
                *(__attribute__((address_space(3))) f32x4 *)(xq) = old_a;
                *(__attribute__((address_space(3))) f32x4 *)(xq + 1024) = old_b;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                old_a = *(__attribute__((address_space(3))) f32x4 *)(xr + 32 * lane);
                old_b = *(__attribute__((address_space(3))) f32x4 *)(xr + 32 * lane + 16);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (ACC) { // (the old flow is zero in the columns outside the image, which are never stored)
                uv[0] = old_a.x + uv[0], uv[1] = old_a.y + uv[1], uv[2] = old_a.z + uv[2], uv[3] = old_a.w + uv[3];
                uv[4] = old_b.x + uv[4], uv[5] = old_b.y + uv[5], uv[6] = old_b.z + uv[6], uv[7] = old_b.w + uv[7];
            }
            if constexpr (WOUT) {
                // the warped row of the step before: second stage and store (nothing is pending in the first emitting step: its
                // store goes nowhere); then this row's first stage, from the flow just formed -- its tap loads have a step to arrive
                const int yw = yy - R; // this step's output row
                const uint32_t wn = warp_row_finish(WM);
                const int wso = __builtin_amdgcn_readfirstlane(s > PR ? (yw - 1 - A.row0) * A.pitch : kOob);
#ifdef OFX_X_NO_WSTORE
                asm volatile("" : : "v"(wn), "s"(wso));
#else
                __builtin_amdgcn_raw_buffer_store_b32(wn, rs_wout, wvo, wso, 0);
#endif
                const float fu[4] = {uv[0], uv[2], uv[4], uv[6]}, fv[4] = {uv[1], uv[3], uv[5], uv[7]};
                warp_row_prepare<ROWWIN>(rs_wsrc, A.warp_scale, A.w, A.h, A.pitch, A.row0, A.row_end, cb, yw, wnpx, fu, fv, WM, wmiss);
            }
            const lds_ptr xl_ws = DEFER ? xl_w + (s & 1) * kLkWaveLds : xl_w;
            *(__attribute__((address_space(3))) f32x4 *)(xl_ws) = f32x4{uv[0], uv[1], uv[2], uv[3]};
            *(__attribute__((address_space(3))) f32x4 *)(xl_ws + 16) = f32x4{uv[4], uv[5], uv[6], uv[7]};
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (!DEFER) {
                const lds_ptr xl_r = xl_base + lane_off_var(l16);
                xlo = *(__attribute__((address_space(3))) f32x4 *)(xl_r);
                xhi = *(__attribute__((address_space(3))) f32x4 *)(xl_r + 1024);
            }
        }
        take_rows();
        if constexpr (!DEFER) {
            if (emit) store_row(s, xlo, xhi);
        }
    };

#if OFX_LK_PROGRESS_PRIORITY
    const int q1 = nsteps / 4, q2 = nsteps / 2, q3 = nsteps - nsteps / 4;
    __builtin_amdgcn_s_setprio(3);
#define OFX_LK_PRIO_STEP()                               \
    do {                                                 \
        if (s >= q3) __builtin_amdgcn_s_setprio(0);      \
        else if (s >= q2) __builtin_amdgcn_s_setprio(1); \
        else if (s >= q1) __builtin_amdgcn_s_setprio(2); \
    } while (0)
#else
#define OFX_LK_PRIO_STEP() ((void)0)
#endif
    int s = 0;
    while (true) {
        body(std::integral_constant<int, 0>{}, s);
        if (++s >= nsteps) break;
        body(std::integral_constant<int, 1>{}, s);
        if (++s >= nsteps) break;
        body(std::integral_constant<int, 2>{}, s);
        if (++s >= nsteps) break;
        OFX_LK_PRIO_STEP();
    }
#undef OFX_LK_PRIO_STEP
    if constexpr (DEFER) {
        if (nsteps > PR) deferred_store(nsteps - 1); // the last row
    }
    // the rows issued by the last step are never taken: they must have landed before the wave gives its LDS back
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (WOUT) { // the warped row of the last step
        const uint32_t wn = warp_row_finish(WM);
        __builtin_amdgcn_raw_buffer_store_b32(wn, rs_wout, wvo, (ye - 1 - A.row0) * A.pitch, 0);
        if constexpr (ROWWIN) {
            if (__any(wmiss != 0u) && A.warp_status != nullptr && lane == 0) atomicOr(A.warp_status, 1 << A.warp_status_bit);
        }
    }
}

} // namespace ofx_dev
