// Device side of the staging kernels of a frame pair: pyramid (level by level and fused), global shift.  Shared by the
// stand-alone kernels in pyramid.hip and by the pipelined stream kernel in lk_level.hip.
#pragma once

#include "ofx_internal.h"

namespace ofx_dev {

// ---------------------------------------------------------------------------------------------------------------
// 2x decimating 3x3 Gaussian on a 1ch plane (OptFlowGpu.cu:1198-1232 / OptFlowCPU.cpp:112-148 with
// GAUS_KERNEL_3x3 = [1 2 1]^T [1 2 1] / 16, kernels.cpp:61-64).  The reference accumulates in float and truncates;
// with power-of-two weights and u8 inputs every partial sum is exact, so integer (sum >> 4) is bit-identical.
// Source taps at column/row -1 are outside the image and skipped; taps 2x+1 <= 2w-1 are always inside.
// A lane produces 4 destination pixels from source columns 8c-1 .. 8c+7.
struct DownArgs {
    const uint8_t *src;
    uint8_t *dst;
    int src_pitch, src_row0, src_row_end; // source buffer holds source rows [src_row0, src_row_end)
    int dw, dh, dst_pitch, dst_row0;      // destination level geometry
    int out_y0, out_y1;
};

// 4 destination pixels (x0 multiple of 4, row y) from source columns 2*x0-1 .. 2*x0+7, rows 2y-1 .. 2y+1.
// Source rows outside [row_lo,row_hi) and columns outside [0,sw) contribute nothing; destination columns >= dw give 0.
__device__ __forceinline__ uint32_t down4(const uint8_t *src, int src_pitch, int src_row0, int row_lo, int row_hi, int sw, int dw,
                                          int x0, int y)
{
    int col[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // vertical [1 2 1] of source columns 2*x0-1 .. 2*x0+7
    const int sx = 2 * x0;                     // multiple of 8
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int sy = 2 * y - 1 + p;
        if (sy < row_lo || sy >= row_hi) continue;
        const int wgt = (p == 1) ? 2 : 1;
        const uint8_t *row = src + (size_t)(sy - src_row0) * (size_t)src_pitch;
        uint32_t lo = 0, hi = 0;
        // the source pitch is a multiple of 4 and >= sw, so a dword starting below sw stays inside the row pitch
        if (sx < sw) lo = *reinterpret_cast<const uint32_t *>(row + sx);
        if (sx + 4 < sw) hi = *reinterpret_cast<const uint32_t *>(row + sx + 4);
        const int left = (sx > 0 && sx - 1 < sw) ? row[sx - 1] : 0;
        col[0] += wgt * left;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = (lo >> (8 * k)) & 0xff, b = (hi >> (8 * k)) & 0xff;
            col[1 + k] += wgt * ((sx + k < sw) ? a : 0);
            col[5 + k] += wgt * ((sx + 4 + k < sw) ? b : 0);
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = (col[2 * k] + 2 * col[2 * k + 1] + col[2 * k + 2]) >> 4;
        out |= (uint32_t)((x0 + k < dw) ? v : 0) << (8 * k); // pitch padding is written as zero
    }
    return out;
}

__device__ __forceinline__ void downsample_rows(const DownArgs &A, int c4 /* group of 4 destination columns */, int row)
{
    const int y = A.out_y0 + row;
    const int x0 = 4 * c4;
    if (x0 >= A.dw || y >= A.out_y1) return;
    const int lo = A.src_row0 > 0 ? A.src_row0 : 0;
    const uint32_t out = down4(A.src, A.src_pitch, A.src_row0, lo, A.src_row_end, 2 * A.dw, A.dw, x0, y);
    *reinterpret_cast<uint32_t *>(A.dst + (size_t)(y - A.dst_row0) * (size_t)A.dst_pitch + x0) = out;
}

// [1 2 1]^T [1 2 1] / 16 of three source rows for 4 destination pixels, SWAR on 16-bit halves: lo / hi = the 8 source bytes
// at columns 2*x0 .. 2*x0+7 of each row, lf = the byte at column 2*x0-1.  A dword's even bytes (b0,b2) and odd bytes
// (b1,b3) are summed down the rows (at most 1020), then out = left + 2*centre + right (at most 4080) >> 4: the same
// integers as down4.
__device__ __forceinline__ uint32_t gauss3_rows(uint32_t l0, uint32_t l1, uint32_t l2, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t f0,
                                                uint32_t f1, uint32_t f2)
{
    const uint32_t M = 0x00ff00ffu;
    const uint32_t elo = (l0 & M) + 2u * (l1 & M) + (l2 & M), olo = ((l0 >> 8) & M) + 2u * ((l1 >> 8) & M) + ((l2 >> 8) & M);
    const uint32_t ehi = (h0 & M) + 2u * (h1 & M) + (h2 & M), ohi = ((h0 >> 8) & M) + 2u * ((h1 >> 8) & M) + ((h2 >> 8) & M);
    const uint32_t left = f0 + 2u * f1 + f2;
    const uint32_t a = 2u * elo + olo + ((olo << 16) | left);        // destination pixels 0 (low half) and 1
    const uint32_t b = 2u * ehi + ohi + ((olo >> 16) | (ohi << 16)); // destination pixels 2 and 3
    return ((a >> 4) & 0xffu) | (((a >> 20) & 0xffu) << 8) | (((b >> 4) & 0xffu) << 16) | (((b >> 20) & 0xffu) << 24);
}

// Two vertically adjacent groups of 4 destination pixels (rows y and y+1, x0 a multiple of 4) from ONE set of loads:
// source rows 2y-1 .. 2y+3, columns 2*x0-1 .. 2*x0+7 of a full-height source (rows [0,sh), columns [0,sw)).  All 15 loads
// are issued before the first use.  INTERIOR: the caller guarantees that every source byte and both destination rows lie
// inside the images (no guards, no masks).
template <bool INTERIOR>
__device__ __forceinline__ void down4x2(const uint8_t *src, int src_pitch, int sh, int sw, int dw, int dh, int x0, int y, uint32_t &out0,
                                        uint32_t &out1)
{
    const int sx = 2 * x0; // multiple of 8
    uint32_t lo[5], hi[5], lf[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int sy = 2 * y - 1 + p;
        if constexpr (INTERIOR) {
            const uint8_t *row = src + (size_t)sy * (size_t)src_pitch + sx;
            lo[p] = *reinterpret_cast<const uint32_t *>(row);
            hi[p] = *reinterpret_cast<const uint32_t *>(row + 4);
            lf[p] = *reinterpret_cast<const uint32_t *>(row - 4) >> 24;
        } else {
            lo[p] = hi[p] = lf[p] = 0u;
            if (sy >= 0 && sy < sh) {
                const uint8_t *row = src + (size_t)sy * (size_t)src_pitch;
                // the source pitch is a multiple of 4 and >= sw, so a dword starting below sw stays inside the row pitch
                if (sx < sw) lo[p] = *reinterpret_cast<const uint32_t *>(row + sx);
                if (sx + 4 < sw) hi[p] = *reinterpret_cast<const uint32_t *>(row + sx + 4);
                if (sx > 0 && sx - 1 < sw) lf[p] = row[sx - 1];
            }
        }
    }
    if constexpr (INTERIOR) {
        out0 = gauss3_rows(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], lf[0], lf[1], lf[2]);
        out1 = gauss3_rows(lo[2], lo[3], lo[4], hi[2], hi[3], hi[4], lf[2], lf[3], lf[4]);
    } else {
        // bytes at columns >= sw (pitch padding) do not count
        const uint32_t mlo = sx + 4 <= sw ? 0xffffffffu : (sx < sw ? 0xffffffffu >> (8 * (sx + 4 - sw)) : 0u);
        const uint32_t mhi = sx + 8 <= sw ? 0xffffffffu : (sx + 4 < sw ? 0xffffffffu >> (8 * (sx + 8 - sw)) : 0u);
#pragma unroll
        for (int p = 0; p < 5; ++p) lo[p] &= mlo, hi[p] &= mhi;
        uint32_t v0 = gauss3_rows(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], lf[0], lf[1], lf[2]);
        uint32_t v1 = gauss3_rows(lo[2], lo[3], lo[4], hi[2], hi[3], hi[4], lf[2], lf[3], lf[4]);
        const uint32_t mx = x0 + 4 > dw ? (x0 < dw ? 0xffffffffu >> (8 * (x0 + 4 - dw)) : 0u) : 0xffffffffu; // padding is written as zero
        out0 = (y >= 0 && y < dh) ? (v0 & mx) : 0u;
        out1 = (y + 1 >= 0 && y + 1 < dh) ? (v1 & mx) : 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Whole pyramid in ONE launch.  A 256-thread workgroup takes a 64x64 tile of level 0 and produces its part of every
// coarser level, keeping the intermediate levels in LDS.  The 3x3 stencil at stride 2 reads source columns 2x-1..2x+1,
// so a tile only ever needs extra source data on its LEFT and TOP: level k+1 needs level k back to 2*X-1, i.e. a
// one-sided halo of H_k = 2*H_{k+1}+1 pixels, H_0 = 2^n - 1 for n produced levels (15 px for a 5-level pyramid).
// Pixels at negative coordinates are outside every level and read as 0 (skipped taps, OptFlowCPU.cpp:133).
constexpr int kPyrTile = 64;
constexpr int kPyrMaxProduced = 6; // levels 1..6 from level 0 in one launch (halo 63)

struct PyrArgs {
    const uint8_t *src;
    uint8_t *dst[kPyrMaxProduced + 1]; // dst[k] = level k plane, k = 1..n
    int pitch[kPyrMaxProduced + 1];    // pitch[0] = source pitch
    int w[kPyrMaxProduced + 1], h[kPyrMaxProduced + 1];
    int n;                             // produced levels
    int stride[kPyrMaxProduced + 1];   // LDS row stride of level k's region (multiple of 4)
    int lds_off[kPyrMaxProduced + 1];  // LDS byte offset of level k's region (levels alternate between two areas; 16 B of slack in front)
    int delta[kPyrMaxProduced + 1];    // LDS column of level k's pixel x is x - X_k + delta[k]; delta[k] = 2*delta[k+1], multiples of 4
    int dst0_pitch;                    // pitch of the optional level-0 copy dst[0]
    // Row window of the destination planes (sharded sessions): dst[k] holds global rows [row0[k], row1[k]) and only those
    // are stored; the source frame is always complete.  by0 = first tile row of the grid.  Whole levels: 0 / h[k] / 0.
    int row0[kPyrMaxProduced + 1], row1[kPyrMaxProduced + 1];
    int by0;
};

constexpr int kPyrThreads = 256; // small workgroups: they must find room next to the LK waves of the previous pair

// one workgroup (kPyrThreads threads, all of them must call this) = one level-0 tile (bx, by)
__device__ __forceinline__ void pyramid_block(const PyrArgs &A, int bx, int by, int tid, uint8_t *lds)
{
    const int X0 = bx * kPyrTile, Y0 = by * kPyrTile;
    const int H0 = (1 << A.n) - 1;

    // optional: keep a copy of the level-0 tile next to the levels built from it (the stream pipeline reads the
    // caller's frame only in the launch that receives it).  One 16-byte piece per thread; the load is issued here and
    // stored after the level-1 stage so that it rides along with that stage's loads.
    uint32_t cp[4] = {0u, 0u, 0u, 0u};
    const int cy = Y0 + (tid >> 2), cx = X0 + 16 * (tid & 3);
    const bool cp_any = A.dst[0] != nullptr && cy >= A.row0[0] && cy < A.row1[0] && cx < A.w[0];
    const bool cp_full = cp_any && cx + 16 <= A.w[0];
    if (cp_full) {
        __builtin_memcpy(cp, A.src + (size_t)cy * (size_t)A.pitch[0] + cx, 16); // 4-byte aligned
    } else if (cp_any) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = cx + 4 * j;
            if (x < A.w[0]) {
                cp[j] = *reinterpret_cast<const uint32_t *>(A.src + (size_t)cy * (size_t)A.pitch[0] + x);
                if (x + 3 >= A.w[0]) cp[j] &= 0xffffffffu >> (8 * (x + 4 - A.w[0])); // padding bytes stay zero
            }
        }
    }

    // LDS layout.  Level k (1 <= k < n) keeps the rows y in [Y_k - H_k, Y_k + T_k) at row y - Y_k + H_k and the pixel x at
    // column x - X_k + delta[k] of its region.  With delta[k] = 2*delta[k+1] the source bytes of 4 destination pixels that
    // start at a multiple of 4 are the bytes 8c-1 .. 8c+7 of the source rows: a thread writes its 4 pixels as ONE aligned
    // dword and reads a 3x9-byte source patch as 9 aligned dwords (byte accesses cost an instruction each, and the pyramid's
    // instructions come out of the LK stage's issue slots).  Columns left of the needed halo hold unneeded values.

    // ---- level 1 straight from HBM (the bulk of the work: no LDS staging of level 0) -----------------------------------------
    // a thread produces 4 horizontally adjacent pixels (global x a multiple of 4) of two rows from one set of loads
    {
        const int Hn = H0 >> 1, Tn = kPyrTile >> 1, Xn = X0 >> 1, Yn = Y0 >> 1;
        const bool keep = A.n > 1; // level 1 is only kept in LDS when a level 2 is built from it
        uint8_t *dstr = lds + A.lds_off[1];
        const int ds = A.stride[1], dl = A.delta[1];
        const int rn = Tn + Hn;
        const int hq = (Hn + 3) & ~3, groups = (Tn + hq + 3) / 4;
        const int pairs = (rn + 1) / 2;
        // interior tile: every level-0 byte the block reads exists -- left/top halo (columns from X0 - 2*hq - 4: the dword
        // holding the leftmost tap; rows from Y0 - 2*Hn - 1) and the two rows a trailing odd row pair reads past the tile
        const bool interior = X0 >= 2 * hq + 4 && Y0 >= 2 * Hn + 1 && X0 + kPyrTile <= A.w[0] && Y0 + kPyrTile + 2 <= A.h[0];
        for (int i = tid; i < groups * pairs; i += kPyrThreads) {
            const int ry = 2 * (i / groups), g = i % groups;
            const int y = Yn - Hn + ry, xb = Xn - hq + 4 * g; // global coordinates at level 1
            uint32_t pk[2] = {0u, 0u};
            if (interior)
                down4x2<true>(A.src, A.pitch[0], A.h[0], A.w[0], A.w[1], A.h[1], xb, y, pk[0], pk[1]);
            else if (xb >= 0 && xb < A.w[1] && y + 1 >= 0 && y < A.h[1])
                down4x2<false>(A.src, A.pitch[0], A.h[0], A.w[0], A.w[1], A.h[1], xb, y, pk[0], pk[1]);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (ry + r >= rn) continue;
                if (keep) *reinterpret_cast<uint32_t *>(dstr + (ry + r) * ds + (xb - Xn + dl)) = pk[r];
                // the tile's own part (not the halo) goes to HBM
                if (ry + r >= Hn && y + r >= A.row0[1] && y + r < A.row1[1] && xb >= Xn && xb < A.w[1]) {
                    uint8_t *row = A.dst[1] + (size_t)(y + r - A.row0[1]) * (size_t)A.pitch[1];
                    if (xb + 3 < A.w[1]) {
                        *reinterpret_cast<uint32_t *>(row + xb) = pk[r];
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (xb + q < A.w[1]) row[xb + q] = (uint8_t)(pk[r] >> (8 * q));
                    }
                }
            }
        }
    }
    if (cp_full) {
        __builtin_memcpy(A.dst[0] + (size_t)(cy - A.row0[0]) * (size_t)A.dst0_pitch + cx, cp, 16);
    } else if (cp_any) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (cx + 4 * j < A.w[0])
                *reinterpret_cast<uint32_t *>(A.dst[0] + (size_t)(cy - A.row0[0]) * (size_t)A.dst0_pitch + cx + 4 * j) = cp[j];
    }
    __syncthreads();

    // ---- deeper levels from LDS --------------------------------------------------------------------------------------
    for (int k = 1; k < A.n; ++k) {
        // producing level k+1 from level k: the taps 2x-1..2x+1 / 2y-1..2y+1 of the pixel at region row ry are the source
        // region rows 2*ry .. 2*ry+2 (H_k = 2*H_{k+1} + 1) and, for a group of 4 pixels starting at region column c, the
        // source columns 2*(c - delta[k+1]) + delta[k] - 1 ... + 7 = (8-aligned) - 1 ... + 7
        const int Hn = H0 >> (k + 1), Tn = kPyrTile >> (k + 1), Xn = X0 >> (k + 1), Yn = Y0 >> (k + 1);
        const bool keep = k + 1 < A.n;
        const uint8_t *srcr = lds + A.lds_off[k];
        uint8_t *dstr = lds + A.lds_off[k + 1];
        const int ss = A.stride[k], ds = A.stride[k + 1], sl = A.delta[k], dl = A.delta[k + 1];
        const int rn = Tn + Hn;
        const int hq = (Hn + 3) & ~3, groups = (Tn + hq + 3) / 4; // the last group may reach past the tile (T < 4): unneeded pixels
        for (int i = tid; i < groups * rn; i += kPyrThreads) {
            const int ry = i / groups, g = i % groups;
            const int y = Yn - Hn + ry, xb = Xn - hq + 4 * g; // global coordinates at level k+1
            const int sc = 2 * (4 * g - hq) + sl;             // source column of tap 2*xb (multiple of 4, >= 0)
            const uint8_t *p0 = srcr + (2 * ry) * ss + sc, *p1 = p0 + ss, *p2 = p1 + ss;
            uint32_t v = gauss3_rows(*reinterpret_cast<const uint32_t *>(p0), *reinterpret_cast<const uint32_t *>(p1),
                                     *reinterpret_cast<const uint32_t *>(p2), *reinterpret_cast<const uint32_t *>(p0 + 4),
                                     *reinterpret_cast<const uint32_t *>(p1 + 4), *reinterpret_cast<const uint32_t *>(p2 + 4),
                                     *reinterpret_cast<const uint32_t *>(p0 - 4) >> 24, *reinterpret_cast<const uint32_t *>(p1 - 4) >> 24,
                                     *reinterpret_cast<const uint32_t *>(p2 - 4) >> 24);
            // pixels outside the image are zero (they are the skipped taps of the next level)
            uint32_t m = 0u;
            if (y >= 0 && y < A.h[k + 1]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (xb + q >= 0 && xb + q < A.w[k + 1]) m |= 0xffu << (8 * q);
            }
            v &= m;
            if (keep) *reinterpret_cast<uint32_t *>(dstr + ry * ds + (xb - Xn + dl)) = v;
            // the tile's own part (not the halo) goes to HBM
            if (ry >= Hn && y >= A.row0[k + 1] && y < A.row1[k + 1] && xb >= Xn && xb < A.w[k + 1]) {
                uint8_t *row = A.dst[k + 1] + (size_t)(y - A.row0[k + 1]) * (size_t)A.pitch[k + 1];
                if (xb + 3 < Xn + Tn && xb + 3 < A.w[k + 1]) {
                    *reinterpret_cast<uint32_t *>(row + xb) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (xb + q < Xn + Tn && xb + q < A.w[k + 1]) row[xb + q] = (uint8_t)(v >> (8 * q));
                }
            }
        }
        __syncthreads();
    }
}

// cpu::shift_back_pyramid on channel 0 (OptFlowCPU.cpp:247, :268-279), destination zero-initialised:
//   target = ((int)(x+u), (int)(y+v)) with float add and truncation toward zero; inside the image -> copy;
//   outside (or non-finite) -> the byte the leading memcpy of w*h bytes left there: the pixel's own value when
//   3*(y*w+x) < w*h, else 0.
struct ShiftArgs {
    const uint8_t *src;
    uint8_t *dst;
    const float *uv;
    int w, h, pitch, row0, row_end, out_y0, out_y1, blocks_x;
};

struct ShiftTable { // (level, pair) items: up to the levels of every pair of a stream tick
    ShiftArgs lv[OFX_MAX_LK_ITEMS];
    int first_block[OFX_MAX_LK_ITEMS + 1];
    int n;
};

// one 256-thread block of the multi-level shift: a thread moves 16 adjacent pixels (four groups of four) of kShiftRows rows.
// Nothing branches between the loads (round 4, second session):
//   * the column map does not depend on the row and is formed once per thread: per group of four pixels a base column (the
//     -- generally unaligned -- dword that holds the targets of its in-image pixels), a byte selector and a mask;
//   * all rows' dwords are issued together; a wave that holds a border of the image (some pixel's target lies outside: that
//     pixel keeps its own value where cpu::shift_back_pyramid's memcpy of w * h bytes put one, OptFlowCPU.cpp:247,270-273 -- the
//     first third of the 3-channel positions -- and is zero after it) also fetches the pixels' own dwords and merges by v_perm;
//   * a column map that does not fit that form (never seen: it needs |u| beyond 2^23) is served byte by byte.
// Measured, and recorded because it was not what the rewrite was for: in front of a tick of eight 4K pairs with iterations this
// launch takes 65 us for its 177 MB (0.34 of the roofline) in EVERY form tried -- one row per thread with a branch per group (the
// form of rounds 1-3: 65-69 us, 138 k waves), four rows per thread (68), this one (65, 17 k waves, 32 dwords in flight per lane)
// -- so neither its wave count nor what a lane keeps in flight bounds it; it runs right behind an accumulating launch's 0.7 GB of
// streaming stores, whose lines the memory side is still writing back when its cold reads arrive.
constexpr int kShiftRows = 8;
__device__ __forceinline__ void shift_block(const ShiftTable &T, int blk, int tid)
{
    if (blk >= T.first_block[T.n]) return;
    int level = 0, hi = T.n; // (binary search: a linear scan is a chain of dependent scalar loads, ~20 items deep in a stream tick)
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (blk >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    const ShiftArgs &A = T.lv[level];
    const int block = blk - T.first_block[level];
    // (the division runs on the vector unit: readfirstlane puts its -- uniform -- results back into scalar registers, so that the
    // row pointers below are scalar and every load is `uniform base + 32-bit lane offset`, one address register instead of two)
    const int by = __builtin_amdgcn_readfirstlane(block / A.blocks_x), bx = __builtin_amdgcn_readfirstlane(block - by * A.blocks_x);
    const int xt = 16 * (bx * 256 + tid);
    const int y0 = A.out_y0 + by * kShiftRows;
    if (xt >= A.pitch || y0 >= A.out_y1) return;
    // (the shift vector comes through a vector load: uniform, but in vector registers until readfirstlane)
    const float u = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, A.uv[0])));
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, A.uv[1])));
    const long long third = ((long long)A.w * (long long)A.h + 2) / 3; // 3 * pos < w * h  <=>  pos < ceil(w * h / 3)
    // the column map of the thread's 16 pixels
    uint32_t base[4], own0[4], sel[4], vmask[4];
    bool weird = false, partial = false;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int x0 = xt + 4 * g;
        int nx[4];
        bool xin[4];
        int ref = x0;
        bool have = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tx = (float)(x0 + k) + u;
            xin[k] = (x0 + k) < A.w && tx > -1.0f && tx < (float)A.w;
            nx[k] = xin[k] ? (int)tx : 0;
            if (xin[k] && !have) ref = nx[k] - k, have = true;
        }
        const int b0 = min(max(ref, 0), A.pitch - 4);
        uint32_t sg = 0u, vm = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = nx[k] - b0;
            if (xin[k]) {
                weird = weird || b < 0 || b > 3;
                sg |= (uint32_t)(b & 3) << (8 * k);
                vm |= 0xffu << (8 * k);
            } else {
                sg |= (uint32_t)(4 + k) << (8 * k); // the pixel's own byte (second operand of the permute)
            }
        }
        base[g] = (uint32_t)b0, own0[g] = (uint32_t)min(x0, A.pitch - 4), sel[g] = sg, vmask[g] = vm;
        if (x0 < A.pitch) partial = partial || vm != 0xffffffffu;
    }
    const bool wide = xt + 16 <= A.pitch && (A.pitch & 15) == 0 && ((uintptr_t)A.dst & 15) == 0;
    // the row map: where row y's pixels come from (block-uniform)
    auto row_src = [&](int y, bool &ok) -> const uint8_t * {
        const float ty = (float)y + v;
        const bool yin = ty > -1.0f && ty < (float)A.h;
        const int ny = yin ? (int)ty : 0;
        ok = yin && ny >= A.row0 && ny < A.row_end;
        return A.src + (size_t)((ok ? ny : y) - A.row0) * (size_t)A.pitch;
    };
    // bytes of the dword at (y, x0) that keep the pixel's own value when nothing is shifted onto it: x < w and inside the first third
    auto own_mask = [&](int y, int x0) -> uint32_t {
        long long lead = third - ((long long)y * A.w + x0);
        const long long inw = (long long)A.w - x0;
        lead = lead < inw ? lead : inw;
        return lead >= 4 ? 0xffffffffu : (lead <= 0 ? 0u : (1u << (8 * (int)lead)) - 1u);
    };
    auto store_row = [&](int y, const uint32_t (&res)[4]) {
        uint8_t *drow = A.dst + (size_t)(y - A.row0) * (size_t)A.pitch + xt;
        if (wide) {
            *reinterpret_cast<uint4 *>(drow) = make_uint4(res[0], res[1], res[2], res[3]);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (xt + 4 * g < A.pitch) *reinterpret_cast<uint32_t *>(drow + 4 * g) = res[g];
        }
    };
    if (!__any(weird)) {
        const bool border = __any(partial) != 0; // (wave-uniform)
        uint32_t S[kShiftRows][4], O[kShiftRows][4];
#pragma unroll
        for (int r = 0; r < kShiftRows; ++r) {
            const int y = min(y0 + r, A.out_y1 - 1); // (rows past the end repeat the last one and are not stored)
            bool ok;
            const uint8_t *srow = row_src(y, ok);
#pragma unroll
            for (int g = 0; g < 4; ++g) __builtin_memcpy(&S[r][g], srow + (ok ? base[g] : own0[g]), 4); // (uniform row base + 32-bit lane offset)
        }
        if (border) {
#pragma unroll
            for (int r = 0; r < kShiftRows; ++r) {
                const int y = min(y0 + r, A.out_y1 - 1);
                const uint8_t *own = A.src + (size_t)(y - A.row0) * (size_t)A.pitch;
#pragma unroll
                for (int g = 0; g < 4; ++g) __builtin_memcpy(&O[r][g], own + own0[g], 4);
            }
        }
#pragma unroll
        for (int r = 0; r < kShiftRows; ++r) {
            const int y = y0 + r;
            if (y >= A.out_y1) break; // (block-uniform)
            bool ok;
            (void)row_src(y, ok);
            uint32_t res[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!ok) res[g] = S[r][g] & own_mask(y, xt + 4 * g); // (S is the pixels' own dword then)
                else if (border) res[g] = __builtin_amdgcn_perm(O[r][g], S[r][g], sel[g]) & (vmask[g] | own_mask(y, xt + 4 * g));
                else res[g] = __builtin_amdgcn_perm(0u, S[r][g], sel[g]);
            }
            store_row(y, res);
        }
        return;
    }
    // any other column map: byte by byte, a row at a time
    for (int r = 0; r < kShiftRows; ++r) {
        const int y = y0 + r;
        if (y >= A.out_y1) break;
        bool ok;
        const uint8_t *srow = row_src(y, ok);
        const uint8_t *own = A.src + (size_t)(y - A.row0) * (size_t)A.pitch;
        uint32_t res[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t out = 0u;
#pragma nounroll
            for (int k = 0; k < 4; ++k) {
                const int x = xt + 4 * g + k;
                const float tx = (float)x + u;
                const bool shifted = ok && x < A.w && tx > -1.0f && tx < (float)A.w;
                const bool keep = x < A.w && (shifted || (long long)y * A.w + x < third);
                const uint32_t b = shifted ? srow[(int)tx] : own[min(x, A.pitch - 1)];
                out |= (keep ? b : 0u) << (8 * k);
            }
            res[g] = out;
        }
        store_row(y, res);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Extension (no reference twin; DESIGN.md "lk_iter"): dst(x,y) = round_u8(bilinear(src, x + s*u, y + s*v)),
// replicate border, non-finite flow = no warp.  Every float operation is written out in the order of
// the CPU restatement used by the tests (DESIGN.md "lk_iter") and the file is built with -ffp-contract=off, so the bytes match.
struct WarpArgs {
    const uint8_t *src;
    uint8_t *dst;
    const float *flow; // interleaved (u,v), row (y - flow_row0)
    float scale;
    int w, h, pitch, row0, out_y0, out_y1, flow_row0, blocks_x;
    int row_end;  // the planes hold image rows [row0, row_end)
    int *status;  // optional: bit status_bit is set when a source row inside the image lay outside the planes
    int status_bit;
};

struct WarpTable {
    WarpArgs lv[OFX_MAX_LK_ITEMS];
    int first_block[OFX_MAX_LK_ITEMS + 1];
    int n;
};

// The warp of 4 adjacent pixels whose 16 taps lie inside a window of 3 rows x 8 bytes, in ~200 instructions (the general
// form below, which also serves row windows, ragged ends and non-finite flows, needs ~330; PMC: 332 -> 226 VALU
// instructions per wave, kernel cycles -6.5 %).  Same arithmetic per pixel as the restatement; what is lean is
// the bookkeeping: |flow| summed once for the finiteness test, clamps as v_med3, the fraction as v_fract (exact for the
// clamped, non-negative coordinate), the window extents as min3 / max3, and the taps picked by byte permutes -- per pixel
// one selector (xi - xbase, x1 - xbase) applied to the three window rows, the six bytes gathered into a register pair and
// the two rows (yi, y1) picked by a second permute whose selector is 0x0202 * row.  Returns false (nothing written) when
// the thread does not qualify.
struct WarpLean { // a row of this thread between its two stages
    float fx[4], fy[4];
    uint32_t selx[4], sela[4], selb[4];
    uint32_t lo[3], hi[3]; // the 3 x 8-byte window (loads in flight between the stages)
};
// stage 1: coordinates, window test, selectors; issues the six window loads.  false: the thread does not qualify.
__device__ __forceinline__ bool warp4_lean_prepare(const WarpArgs &A, int x0, int y, const float (&fu)[4], const float (&fv)[4], WarpLean &M)
{
    // every |s*u|, |s*v| <= 0.54e8 when the sum of the magnitudes is <= 1e8 (NaN / Inf fail the test): all four pixels finite
    const float mag = ((__builtin_fabsf(fu[0]) + __builtin_fabsf(fv[0])) + (__builtin_fabsf(fu[1]) + __builtin_fabsf(fv[1]))) +
                      ((__builtin_fabsf(fu[2]) + __builtin_fabsf(fv[2])) + (__builtin_fabsf(fu[3]) + __builtin_fabsf(fv[3])));
    if (!(mag <= 1e8f)) return false;
    const float xf0 = (float)x0, yf = (float)y, wmaxf = (float)(A.w - 1), hmaxf = (float)(A.h - 1);
    int xi[4], yi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sx = __builtin_amdgcn_fmed3f((xf0 + (float)k) + A.scale * fu[k], 0.0f, wmaxf);
        const float sy = __builtin_amdgcn_fmed3f(yf + A.scale * fv[k], 0.0f, hmaxf);
        xi[k] = (int)sx;
        yi[k] = (int)sy;
        M.fx[k] = __builtin_amdgcn_fractf(sx); // == sx - (float)xi: sx >= 0, the difference is exact
        M.fy[k] = __builtin_amdgcn_fractf(sy);
    }
    const int xmin = min(min(xi[0], xi[1]), min(xi[2], xi[3])), ximax = max(max(xi[0], xi[1]), max(xi[2], xi[3]));
    const int ymin = min(min(yi[0], yi[1]), min(yi[2], yi[3])), yimax = max(max(yi[0], yi[1]), max(yi[2], yi[3]));
    const int wmax = A.w - 1, hmax = A.h - 1;
    const int xbase = min(xmin, A.pitch - 8); // the 8-byte window stays inside the row pitch
    if (A.pitch < 8 || min(ximax + 1, wmax) - xbase > 7 || min(yimax + 1, hmax) - ymin > 2) return false;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint8_t *row = A.src + (size_t)(uint32_t)(min(ymin + r, hmax) * A.pitch + xbase); // (rows past y1 are never selected)
        __builtin_memcpy(&M.lo[r], row, 4);
        __builtin_memcpy(&M.hi[r], row + 4, 4);
    }
    const int cx = (int)0x0c0c0000 - xbase, c1 = 1 - xbase, cw = wmax - xbase;
    const int cy = (int)0x0c0c0100 - 0x0202 * ymin;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // bytes (xi - xbase, x1 - xbase) of a window row into bytes 0, 1 (selector bytes 0-3: lo, 4-7: hi, 0x0c: zero)
        M.selx[k] = ((uint32_t)min(xi[k] + c1, cw) << 8) | (uint32_t)(xi[k] + cx);
        // row r's pair will sit at bytes 2r, 2r + 1 of a register pair (stage 2)
        M.sela[k] = (uint32_t)(0x0202 * yi[k] + cy);
        M.selb[k] = (uint32_t)(0x0202 * min(yi[k] + 1, hmax) + cy);
    }
    return true;
}
// stage 2: the window has arrived
__device__ __forceinline__ uint32_t warp4_lean_finish(const WarpLean &M)
{
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t0 = __builtin_amdgcn_perm(M.hi[0], M.lo[0], M.selx[k]), t1 = __builtin_amdgcn_perm(M.hi[1], M.lo[1], M.selx[k]),
                       t2 = __builtin_amdgcn_perm(M.hi[2], M.lo[2], M.selx[k]);
        const uint32_t t01 = __builtin_amdgcn_perm(t1, t0, 0x05040100u); // row 0's pair in bytes 0-1, row 1's in bytes 2-3
        const uint32_t pa = __builtin_amdgcn_perm(t2, t01, M.sela[k]), pb = __builtin_amdgcn_perm(t2, t01, M.selb[k]);
        const float p00 = (float)(pa & 0xffu), p01 = (float)((pa >> 8) & 0xffu);
        const float p10 = (float)(pb & 0xffu), p11 = (float)((pb >> 8) & 0xffu);
        const float a = p00 + M.fx[k] * (p01 - p00);
        const float b = p10 + M.fx[k] * (p11 - p10);
        const float v = a + M.fy[k] * (b - a);
        out |= ((uint32_t)(int)(v + 0.5f) & 0xffu) << (8 * k);
    }
    return out;
}

// Bilinear warp of 4 adjacent pixels per thread.  The arithmetic per pixel is the restatement's (clamp to the image,
// a = p00 + fx*(p01-p00), b = p10 + fx*(p11-p10), v = a + fy*(b-a), round half up); what differs is how the four taps are
// fetched.  Flow fields are smooth, so the taps of a thread's 4 pixels almost always lie inside a window of 3 rows x 8
// bytes: the thread then loads that window with 6 (unaligned) dword loads that are contiguous across the lanes of a wave
// and picks the taps with v_perm_b32, instead of gathering 16 single bytes (the gather form is TA-bound: 43 us for a 4K
// pyramid).  A thread whose taps do not fit, or that has a non-finite flow, takes the gather path.
__device__ __forceinline__ uint32_t warp4_general(const WarpArgs &A, int x0, int y, int npx, const float (&fu)[4], const float (&fv)[4]);

// One thread = four adjacent pixels of one row; a block = 256 threads of one row.  (Rows per thread were tried both ways in
// round 2 -- 2 / 4 rows one after the other: 35.1 / 37.4 us per 4K pyramid against 33.8; 2 ... 16 rows with the next row's
// flow prefetched during the current one: 39.3 ... 51.6 us -- the kernel wants many short waves.)
__device__ __forceinline__ void warp_block(const WarpTable &T, int blk, int tid)
{
    if (blk >= T.first_block[T.n]) return;
    int level = 0, hi = T.n; // (binary search: a linear scan is a chain of dependent scalar loads, ~20 items deep in a stream tick)
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (blk >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    const WarpArgs &A = T.lv[level];
    const int block = blk - T.first_block[level];
    const int bx = block % A.blocks_x, by = block / A.blocks_x;
    const int x0 = 4 * (bx * 256 + tid);
    const int y = A.out_y0 + by;
    if (x0 >= A.pitch || y >= A.out_y1) return;
    const float *frow = A.flow + 2 * ((size_t)(y - A.flow_row0) * (size_t)A.w);
    const int npx = A.w - x0 < 4 ? (A.w - x0 > 0 ? A.w - x0 : 0) : 4;

    float fu[4] = {0, 0, 0, 0}, fv[4] = {0, 0, 0, 0};
    if (npx == 4) {
        float tmp[8];
        __builtin_memcpy(tmp, frow + 2 * x0, 32); // 8-byte aligned: two 16-byte loads
#pragma unroll
        for (int k = 0; k < 4; ++k) fu[k] = tmp[2 * k], fv[k] = tmp[2 * k + 1];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < npx) fu[k] = frow[2 * (x0 + k)], fv[k] = frow[2 * (x0 + k) + 1];
    }
    uint32_t out;
    WarpLean M;
    // the lean form of the common case: whole-level planes, four pixels, taps inside a 3 x 8 window
    if (npx == 4 && A.row0 == 0 && A.row_end >= A.h && warp4_lean_prepare(A, x0, y, fu, fv, M)) out = warp4_lean_finish(M);
    else out = warp4_general(A, x0, y, npx, fu, fv);
    *reinterpret_cast<uint32_t *>(A.dst + (size_t)(y - A.row0) * (size_t)A.pitch + x0) = out;
}

// The general form of a thread's four pixels (row windows, ragged ends, non-finite flows, taps outside a 3 x 8 window).
__device__ __forceinline__ uint32_t warp4_general(const WarpArgs &A, int x0, int y, int npx, const float (&fu)[4], const float (&fv)[4])
{
    int xi[4], yi[4], x1[4], y1[4];
    float fx[4], fy[4];
    bool finite = true;
    int xmin = 0x7fffffff, xmax = -1, ymin = 0x7fffffff, ymax = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float sx = (float)(x0 + k) + A.scale * fu[k];
        float sy = (float)y + A.scale * fv[k];
        const bool ok = (sx >= -1e9f && sx <= 1e9f) && (sy >= -1e9f && sy <= 1e9f);
        if (k < npx) finite = finite && ok;
        sx = sx < 0.0f ? 0.0f : (sx > (float)(A.w - 1) ? (float)(A.w - 1) : sx);
        sy = sy < 0.0f ? 0.0f : (sy > (float)(A.h - 1) ? (float)(A.h - 1) : sy);
        if (!ok) sx = 0.0f, sy = 0.0f;
        xi[k] = (int)sx;
        yi[k] = (int)sy;
        x1[k] = xi[k] + 1 < A.w ? xi[k] + 1 : A.w - 1;
        y1[k] = yi[k] + 1 < A.h ? yi[k] + 1 : A.h - 1;
        fx[k] = sx - (float)xi[k];
        fy[k] = sy - (float)yi[k];
        if (k < npx) {
            xmin = min(xmin, xi[k]);
            xmax = max(xmax, x1[k]);
            ymin = min(ymin, yi[k]);
            ymax = max(ymax, y1[k]);
        }
    }
    if (A.row0 > 0 || A.row_end < A.h) { // wave-uniform: a row window of the level (row-sharded caller)
        bool miss = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < npx && (yi[k] < A.row0 || y1[k] >= A.row_end)) miss = true;
            yi[k] = min(max(yi[k], A.row0), A.row_end - 1);
            y1[k] = min(max(y1[k], A.row0), A.row_end - 1);
        }
        ymin = min(max(ymin, A.row0), A.row_end - 1);
        ymax = min(max(ymax, A.row0), A.row_end - 1);
        if (miss && finite && A.status != nullptr) atomicOr(A.status, 1 << A.status_bit);
    }
    uint32_t out = 0;
    const int xbase = min(xmin, A.pitch - 8); // the 8-byte window stays inside the row pitch
    if (npx > 0 && finite && A.pitch >= 8 && xmax - xbase <= 7 && ymax - ymin <= 2) {
        uint32_t lo[3], hi[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yr = min(ymin + r, min(A.h, A.row_end) - 1); // rows past ymax are never selected
            const uint8_t *row = A.src + (size_t)(yr - A.row0) * (size_t)A.pitch + xbase;
            __builtin_memcpy(&lo[r], row, 4);
            __builtin_memcpy(&hi[r], row + 4, 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= npx) break;
            const int ra = yi[k] - ymin, rb = y1[k] - ymin; // 0..2
            const uint32_t alo = ra == 0 ? lo[0] : (ra == 1 ? lo[1] : lo[2]), ahi = ra == 0 ? hi[0] : (ra == 1 ? hi[1] : hi[2]);
            const uint32_t blo = rb == 0 ? lo[0] : (rb == 1 ? lo[1] : lo[2]), bhi = rb == 0 ? hi[0] : (rb == 1 ? hi[1] : hi[2]);
            // byte c of the 8-byte window (hi:lo) into byte 0, zeros above: selector bytes 0-3 = lo, 4-7 = hi, 0x0c = 0
            const uint32_t s0 = 0x0c0c0c00u | (uint32_t)(xi[k] - xbase), s1 = 0x0c0c0c00u | (uint32_t)(x1[k] - xbase);
            const float p00 = (float)__builtin_amdgcn_perm(ahi, alo, s0), p01 = (float)__builtin_amdgcn_perm(ahi, alo, s1);
            const float p10 = (float)__builtin_amdgcn_perm(bhi, blo, s0), p11 = (float)__builtin_amdgcn_perm(bhi, blo, s1);
            const float a = p00 + fx[k] * (p01 - p00);
            const float b = p10 + fx[k] * (p11 - p10);
            const float v = a + fy[k] * (b - a);
            out |= ((uint32_t)(int)(v + 0.5f) & 0xffu) << (8 * k);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= npx) break;
            const int x = x0 + k;
            const float sxr = (float)x + A.scale * fu[k], syr = (float)y + A.scale * fv[k];
            uint32_t val;
            if (!(sxr >= -1e9f && sxr <= 1e9f) || !(syr >= -1e9f && syr <= 1e9f)) {
                val = A.src[(size_t)(y - A.row0) * (size_t)A.pitch + x];
            } else {
                const uint8_t *r0 = A.src + (size_t)(yi[k] - A.row0) * (size_t)A.pitch;
                const uint8_t *r1 = A.src + (size_t)(y1[k] - A.row0) * (size_t)A.pitch;
                const float p00 = r0[xi[k]], p01 = r0[x1[k]], p10 = r1[xi[k]], p11 = r1[x1[k]];
                const float a = p00 + fx[k] * (p01 - p00);
                const float b = p10 + fx[k] * (p11 - p10);
                const float v = a + fy[k] * (b - a);
                val = (uint32_t)(int)(v + 0.5f);
            }
            out |= (val & 0xffu) << (8 * k);
        }
    }
    return out;
}

} // namespace ofx_dev
