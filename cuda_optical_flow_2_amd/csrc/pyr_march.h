// The pyramid of a frame as a MARCH: one wave walks down a column tile of level 0 and produces the rows of every level
// as it goes, entirely in registers -- no LDS, no barriers, no tile halo recomputed level by level.  Used by the stream
// kernel, where the pyramid's instructions come out of the LK stage's issue slots: the tiled kernel (stages_body.h,
// still used for stand-alone launches, where latency matters more than instruction count) spends ~1.4 k wave-instructions
// per 64x64 tile, ~10x the arithmetic, on addressing, guards, LDS traffic and a 52 % halo; the march spends ~90 per
// 512 x 2 level-0 pixels.
//
// Layout.  Lane l holds 8 adjacent level-0 bytes (two dwords: a wave row is 512 contiguous bytes), hence 4 level-1, 2
// level-2 and 1 level-3 pixel; from level 4 on a pixel lives in every 2nd, 4th, 8th lane.  The 3x3 stencil at stride 2
// reaches one source pixel to the LEFT of 2x and one row ABOVE 2y, so
//   * horizontally a tile carries `halo` lanes of overlap on its left (8*halo >= 2^n - 1 level-0 pixels), whose results
//     are computed but not stored; the left neighbour of a lane's first pixel arrives by one DPP / permute per row;
//   * vertically a strip starts 2^n level-0 rows early (the rows above feed the carried rows of every level) and every
//     level keeps just two source rows: `carry` (row 2y-1) and `mid` (row 2y); row 2y+1 completes an output row.
// Arithmetic: gauss3 = [1 2 1]^T [1 2 1] / 16 on exact integers, vertical sums first (<= 1020), then horizontal (<= 4080),
// then >> 4 -- the integers of down4 / cpu::downscale_gaussian (OptFlowCPU.cpp:112-148).  Pixels outside the image are
// zero at every level (skipped taps, OptFlowCPU.cpp:133).
#pragma once

#include "lk_body.h" // lane_shift_right / lane_shift_left, pin helpers

// streaming (nt) reads of the source frame when the session keeps its own copy: measured no gain at 4K, -20 % at 1080p
#ifndef OFX_PYR_NT_SOURCE
#define OFX_PYR_NT_SOURCE 0
#endif

namespace ofx_dev {

constexpr int kMarchMaxProduced = 6;

struct PyrMarchArgs {
    const uint8_t *src; // whole level-0 frame
    int src_pitch;
    uint8_t *dst[kMarchMaxProduced + 1]; // dst[0] = optional level-0 copy, dst[k] = level k plane
    int pitch[kMarchMaxProduced + 1];
    int w[kMarchMaxProduced + 1], h[kMarchMaxProduced + 1];
    int row0[kMarchMaxProduced + 1], row1[kMarchMaxProduced + 1]; // destination row windows (whole levels: 0 .. h)
    int n;                                                         // produced levels (1 .. 6)
    int tiles_x, strips, strip_h, tile_w, halo; // tile_w = (64 - halo) * 8 level-0 columns, strip_h level-0 rows (multiple of 2^n)
    int y_lo;                                   // first level-0 row covered (multiple of 2^n)
    int y_hi;                                   // end of the level-0 rows covered
};

__device__ __forceinline__ uint32_t march_shift_up(uint32_t v, int s, int lane) // value of lane - s, 0 if there is none
{
    const uint32_t r = (uint32_t)__shfl_up((int)v, s);
    return lane >= s ? r : 0u;
}
__device__ __forceinline__ uint32_t march_shift_down(uint32_t v, int s, int lane) // value of lane + s, 0 if there is none
{
    const uint32_t r = (uint32_t)__shfl_down((int)v, s);
    return lane + s < 64 ? r : 0u;
}

// one wave = one (tile, strip) item
__device__ __forceinline__ void pyr_march_wave(const PyrMarchArgs &A, int item, int lane)
{
    if (item >= A.tiles_x * A.strips) return;
    const int tx = item % A.tiles_x, st = item / A.tiles_x;
    const int n = A.n;
    const int Ys = A.y_lo + st * A.strip_h;                       // level-0 rows [Ys, Ye) belong to this strip
    const int Ye = min(Ys + A.strip_h, A.y_hi);
    const int r_start = max(0, Ys - (1 << n));                    // the march starts here (a multiple of 2^n)
    const int col0 = tx * A.tile_w - 8 * A.halo + 8 * lane;       // level-0 column of this lane's first byte (may be < 0)
    const bool own = lane >= A.halo;                              // this lane's pixels belong to the tile (are stored)

    // per-level pixel position and validity masks of this lane (pixels outside the image are zero)
    auto bytes_mask = [](int x, int cnt, int w) { // mask of the bytes j < cnt with 0 <= x + j < w
        uint32_t m = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < cnt && x + j >= 0 && x + j < w) m |= 0xffu << (8 * j);
        return m;
    };
    const uint32_t m0lo = bytes_mask(col0, 4, A.w[0]), m0hi = bytes_mask(col0 + 4, 4, A.w[0]);
    const int x1 = col0 >> 1, x2 = col0 >> 2, x3 = col0 >> 3; // arithmetic shifts: col0 is a multiple of 8
    const uint32_t m1 = bytes_mask(x1, 4, A.w[1]);
    const uint32_t m2 = n >= 2 ? bytes_mask(x2, 2, A.w[2]) : 0u;
    const uint32_t m3 = n >= 3 ? bytes_mask(x3, 1, A.w[3]) : 0u;
    uint32_t mk[kMarchMaxProduced + 1] = {0, 0, 0, 0, 0, 0, 0}; // levels >= 4: one pixel in every 2^(k-3)-th lane
    int xk[kMarchMaxProduced + 1] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 4; k <= kMarchMaxProduced; ++k) {
        xk[k] = col0 >> k;
        const bool holds = (lane & ((1 << (k - 3)) - 1)) == 0; // tile base and halo are multiples of 2^n >= 2^k
        mk[k] = (k <= n && holds && xk[k] >= 0 && xk[k] < A.w[k]) ? 0xffu : 0u;
    }
    // Branch-free loads: the addresses are clamped into the row (pitch >= 8, a multiple of 4); a clamped dword lies entirely
    // outside the image and is masked away.
    const uint32_t lo_off = (uint32_t)min(max(col0, 0), A.src_pitch - 4), hi_off = (uint32_t)min(max(col0, 0) + 4, A.src_pitch - 4);

    // (raw: the masks are applied where a fetched row is taken into use, so that nothing waits on a load just issued)
    auto fetch_row = [&](int r, uint32_t &lo, uint32_t &hi) {
        lo = hi = 0u;
        if (r >= 0 && r < A.h[0]) { // uniform
            const uint8_t *row = A.src + (size_t)r * (size_t)A.src_pitch;
            pin_scalar(row); // scalar row base + 32-bit lane offset: no 64-bit VALU address arithmetic
#if OFX_PYR_NT_SOURCE
            if (A.dst[0] != nullptr) { // the session keeps its own copy of level 0: nobody reads the source frame again
                lo = gload_u32_nt(row, lo_off);
                hi = gload_u32_nt(row, hi_off);
            } else
#endif
            {
                lo = gload_u32(row, lo_off);
                hi = gload_u32(row, hi_off);
            }
        }
    };
    auto load_row = [&](int r, uint32_t &lo, uint32_t &hi) {
        fetch_row(r, lo, hi);
        lo &= m0lo;
        hi &= m0hi;
    };
    auto emit_ok = [&](int k, int yk) { // does level-k row yk belong to this strip and to the destination window?
        const int y0 = yk << k;
        return y0 >= Ys && y0 < Ye && yk >= A.row0[k] && yk < A.row1[k];
    };

    uint32_t c_lo = 0u, c_hi = 0u;           // level-0 row 2*y1 - 1
    if (r_start > 0) load_row(r_start - 1, c_lo, c_hi);
    const uint32_t M = 0x00ff00ffu;
    // per level k >= 2: the source rows 2y-1 (carry) and 2y (mid) of the output row in the making.  Scalars, not arrays:
    // hipcc copied whole arrays around every conditional update (~70 v_mov per step).
    uint32_t carry2 = 0, carry3 = 0, carry4 = 0, carry5 = 0, carry6 = 0, mid2 = 0, mid3 = 0, mid4 = 0, mid5 = 0, mid6 = 0;
    auto level2 = [&](uint32_t c, uint32_t m, uint32_t o) { // source: 4 pixels per lane -> 2 pixels
        const uint32_t E = (c & M) + 2u * (m & M) + (o & M), O = ((c >> 8) & M) + 2u * ((m >> 8) & M) + ((o >> 8) & M);
        const uint32_t left = (uint32_t)lane_shift_right((int)O) >> 16;
        const uint32_t a = 2u * E + O + ((O << 16) | left);
        return (((a >> 4) & 0xffu) | (((a >> 20) & 0xffu) << 8)) & m2;
    };
    auto level3 = [&](uint32_t c, uint32_t m, uint32_t o) { // 2 pixels per lane -> 1
        const uint32_t E = (c & 0xffu) + 2u * (m & 0xffu) + (o & 0xffu), O = ((c >> 8) & 0xffu) + 2u * ((m >> 8) & 0xffu) + ((o >> 8) & 0xffu);
        const uint32_t left = (uint32_t)lane_shift_right((int)O);
        return ((left + 2u * E + O) >> 4) & m3;
    };
    auto level_n = [&](int k, uint32_t c, uint32_t m, uint32_t o) { // one pixel per 2^(k-4)-th lane -> one per 2^(k-3)-th
        const int sft = 1 << (k - 4);
        const uint32_t V = c + 2u * m + o; // zero in the lanes that hold no pixel
        const uint32_t left = k == 4 ? (uint32_t)lane_shift_right((int)V) : march_shift_up(V, sft, lane);
        const uint32_t right = k == 4 ? (uint32_t)lane_shift_left((int)V) : march_shift_down(V, sft, lane);
        return ((left + 2u * V + right) >> 4) & mk[k];
    };

    uint32_t e_lo, e_hi, o_lo, o_hi; // rows 2*y1, 2*y1 + 1 of the current step
    const int y1_begin = r_start >> 1, y1_end = (Ye + 1) >> 1;
    load_row(2 * y1_begin, e_lo, e_hi);
    load_row(2 * y1_begin + 1, o_lo, o_hi);
    // One step = level-1 row y1 (level-0 rows 2*y1, 2*y1 + 1).  PH = y1 mod 4 is a compile-time constant (the loop below is
    // unrolled four times; r_start is a multiple of 2^n, so y1_begin is even when level 2 exists and a multiple of 4 when
    // level 3 does): which of levels 2 and 3 a step feeds or completes is then static, and their state needs no
    // conditional updates -- written with run-time tests on y1 the compiler merged the branches with ~55 v_mov per step.
    auto step = [&](auto PHc, int y1) {
        constexpr int PH = decltype(PHc)::value;
        // fetch the next step's rows before touching this step's (consumed at the bottom of the step)
        uint32_t ne_lo, ne_hi, no_lo, no_hi;
        fetch_row(2 * y1 + 2, ne_lo, ne_hi);
        fetch_row(2 * y1 + 3, no_lo, no_hi);

        // ---- arithmetic of the step: nothing is stored yet (gfx9 counts loads and stores in one counter and only orders
        // returns within a type, so a wait for the fetched rows with this step's stores outstanding would wait for those too)
        // level 1, row y1: vertical [1 2 1] of rows 2y1-1, 2y1, 2y1+1 on the even / odd bytes, then horizontal
        uint32_t out1, out2 = 0, out3 = 0, out4 = 0, out5 = 0, out6 = 0; // outK: the level-K row this step completed (K <= n_out)
        int n_out = 1;
        {
            const uint32_t elo = (c_lo & M) + 2u * (e_lo & M) + (o_lo & M), olo = ((c_lo >> 8) & M) + 2u * ((e_lo >> 8) & M) + ((o_lo >> 8) & M);
            const uint32_t ehi = (c_hi & M) + 2u * (e_hi & M) + (o_hi & M), ohi = ((c_hi >> 8) & M) + 2u * ((e_hi >> 8) & M) + ((o_hi >> 8) & M);
            const uint32_t left = (uint32_t)lane_shift_right((int)ohi) >> 16; // the left lane's byte 7, summed
            const uint32_t a = 2u * elo + olo + ((olo << 16) | left);
            const uint32_t b = 2u * ehi + ohi + ((olo >> 16) | (ohi << 16));
            out1 = (((a >> 4) & 0xffu) | (((a >> 20) & 0xffu) << 8) | (((b >> 4) & 0xffu) << 16) | (((b >> 20) & 0xffu) << 24)) & m1;
            c_lo = o_lo;
            c_hi = o_hi;
        }
        // deeper levels: level k gets a source row every 2^(k-2) steps, an even one waits in mid, an odd one completes a row
        if constexpr ((PH & 1) == 0) {
            mid2 = out1;
        } else {
            if (n >= 2) {
                out2 = level2(carry2, mid2, out1);
                carry2 = out1;
                n_out = 2;
            }
            if constexpr (PH == 1) {
                mid3 = out2;
            } else {
                if (n >= 3) {
                    out3 = level3(carry3, mid3, out2);
                    carry3 = out2;
                    n_out = 3;
                    if (n >= 4) {
                        if (((y1 >> 2) & 1) == 0) {
                            mid4 = out3;
                        } else {
                            out4 = level_n(4, carry4, mid4, out3);
                            carry4 = out3;
                            n_out = 4;
                            if (n >= 5) {
                                if (((y1 >> 3) & 1) == 0) {
                                    mid5 = out4;
                                } else {
                                    out5 = level_n(5, carry5, mid5, out4);
                                    carry5 = out4;
                                    n_out = 5;
                                    if (n >= 6) {
                                        if (((y1 >> 4) & 1) == 0) {
                                            mid6 = out5;
                                        } else {
                                            out6 = level_n(6, carry6, mid6, out5);
                                            carry6 = out5;
                                            n_out = 6;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }

        // ---- take the fetched rows (one wait: only loads are outstanding), then store
        asm volatile("" : "+v"(ne_lo), "+v"(ne_hi), "+v"(no_lo), "+v"(no_hi)); // the fetched rows are taken here, not earlier
        // (scalar row pointer + 32-bit lane offset for every store, as in the LK march)
        if (own) {
            if (A.dst[0] != nullptr && col0 < A.w[0]) { // level-0 copy of the two rows
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r = 2 * y1 + t;
                    if (r >= Ys && r < Ye && r >= A.row0[0] && r < A.row1[0]) {
                        // 8 contiguous bytes per lane, 512 per wave; bytes beyond w are zero (masked loads) and land in the pitch
                        // padding.  (Streaming stores measured slower here: the step's single wait also covers the previous
                        // step's stores, and those take longer to complete when they bypass the L2.)
                        uint8_t *row = A.dst[0] + (size_t)(r - A.row0[0]) * (size_t)A.pitch[0];
                        pin_scalar(row);
                        if (col0 + 8 <= A.pitch[0])
                            gstore_u32x2(row, (uint32_t)col0, t ? o_lo : e_lo, t ? o_hi : e_hi);
                        else
                            gstore_u32(row, (uint32_t)col0, t ? o_lo : e_lo);
                    }
                }
            }
            if (x1 < A.w[1] && emit_ok(1, y1)) {
                uint8_t *row = A.dst[1] + (size_t)(y1 - A.row0[1]) * (size_t)A.pitch[1];
                pin_scalar(row);
                gstore_u32(row, (uint32_t)x1, out1);
            }
            auto store_level = [&](int k, uint32_t value) { // k >= 2, constant at every call site
                const int yk = y1 >> (k - 1);
                if (k > n_out || !emit_ok(k, yk)) return;
                uint8_t *row = A.dst[k] + (size_t)(yk - A.row0[k]) * (size_t)A.pitch[k];
                pin_scalar(row);
                if (k == 2) {
                    if (x2 < A.w[2]) gstore_u16(row, (uint32_t)x2, (uint16_t)value);
                } else if (k == 3) {
                    if (m3) gstore_u8(row, (uint32_t)x3, (uint8_t)value);
                } else {
                    if (mk[k]) gstore_u8(row, (uint32_t)xk[k], (uint8_t)value);
                }
            };
            if constexpr ((PH & 1) == 1) store_level(2, out2);
            if constexpr (PH == 3) {
                store_level(3, out3);
                store_level(4, out4);
                store_level(5, out5);
                store_level(6, out6);
            }
        }
        e_lo = ne_lo & m0lo;
        e_hi = ne_hi & m0hi;
        o_lo = no_lo & m0lo;
        o_hi = no_hi & m0hi;
        // (keeps the masking on this side of the step's edge: sunk into the next step it would sit behind that step's loads,
        // and the wait in front of it would be a wait for those)
        asm volatile("" : "+v"(e_lo), "+v"(e_hi), "+v"(o_lo), "+v"(o_hi));
    };
    // Steps past y1_end (the last group of a strip) are harmless: rows outside the image read as zeros and every store is
    // guarded by the strip's and the destination's row ranges.
    for (int y1 = y1_begin; y1 < y1_end; y1 += 4) {
        step(std::integral_constant<int, 0>{}, y1);
        step(std::integral_constant<int, 1>{}, y1 + 1);
        step(std::integral_constant<int, 2>{}, y1 + 2);
        step(std::integral_constant<int, 3>{}, y1 + 3);
    }
}

} // namespace ofx_dev
