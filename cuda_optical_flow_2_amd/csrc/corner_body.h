// Device side of the corner kernel (see corner.hip).  Shared with the pipelined stream kernel.
#pragma once

#include "lk_solve.h"
#include "ofx_internal.h"
#include "stages_body.h"

namespace ofx_dev {

struct CornerLevel {
    const uint8_t *prev;
    const uint8_t *next; // unshifted
    float *flow;         // pixel 0 is written when flow_row0 == 0
    int w, h, pitch, row_end, flow_row0;
    int col_end; // the planes hold columns [0, col_end) and rows [0, row_end) of the w x h level (a top-left patch, or all of it)
    // row-sharded sessions (all zero: unchecked): the image rows [need0, need1) the level kernel's stencils of this shard
    // touch before the shift, and the rows [valid0, valid1) its buffers hold
    int need0, need1, valid0, valid1;
};

// A corner chain: its header and, separately, its levels -- the stream kernel keeps the levels of all the chains of a tick in
// one flat table ((pair, level) items, like the LK stage) so that sixteen chains fit its argument block.
struct CornerHead {
    float *uv;   // 2 floats per level
    int *status; // optional: bit k is set when level k needed a pixel inside the image but outside its planes, bit 8 + k
                 // when level k's vertical shift sends the shard's reads to image rows its buffers do not hold
    int levels, radius;
    float min_det; // determinant guard of the solve (lk_solve.h), as the level kernel of the same pair applies it
    int lv0;       // stream kernel: index of the chain's level 0 in the flat table
    // optional: THIS pair's word, written (not OR-ed) when the chain ends -- the bits the chain raised in `status` plus
    // kCornerRepaired when a level had to be read through a relocated patch (the result is then exact all the same)
    int *pair_status;
    // non-null: the block may rebuild the next frame's patch pyramid AROUND a shifted corner that left the top-left patch
    // (corner_block): planes of levels 1 .. levels-1 laid out like one frame of PatchBuild.  Level 0's planes must then be the
    // whole frames.
    uint8_t *reloc;
};
constexpr int kCornerRepaired = 1 << 24;
struct CornerArgs { // the stand-alone corner kernel
    CornerHead hd;
    CornerLevel lv[OFX_MAX_LEVELS];
};

// ---- patch pyramids built by the corner block itself (two-stage stream pipeline) --------------------------------------------
// Levels 1 .. n of the top-left pw[0] x ph[0] corner of two frames (a pair's previous and next frame), each level from the
// one below with the pyramid's own stencil (down4, stages_body.h): the pyramid of such a patch IS the top-left part of the
// frame's pyramid (the stencil 2x-1 .. 2x+1 never reaches past column / row 2 * w_k - 1).  One block of 256 threads, a
// barrier per level; the planes are global memory private to the block's chain (written and read by this block only:
// the workgroup's L1 sees its own stores).  256 x 256 level-0 pixels: ~43 groups of four pixels per thread and frame.
struct PatchBuild {
    int n;                                   // produced levels (0: nothing to build)
    int pw[OFX_MAX_LEVELS], ph[OFX_MAX_LEVELS], pitch[OFX_MAX_LEVELS]; // [0]: the patch of the frame itself (pitch: per slot)
    int off[OFX_MAX_LEVELS];                 // byte offset of level k inside a frame's patch planes
    int frame_stride;                        // bytes between the planes of the slot's two frames
    int first;                               // 0: build both frames' planes; 1: only the second's (pyr_corner.hip: the first's are at hand)
};
struct PatchBuildSlot {
    const uint8_t *src[2]; // the frames (level 0, whole rows from column 0)
    int src_pitch[2];
    uint8_t *base;         // this slot's planes: frame f's level k at base + f * frame_stride + off[k]
};

#ifndef OFX_PATCH_ITEMS
#define OFX_PATCH_ITEMS 2 // work items (a group of four pixels on two rows: fifteen loads) a thread has in flight per pass (4: the stream kernels spill)
#endif
__device__ __forceinline__ void patch_build_block(const PatchBuild &P, const PatchBuildSlot &S, int tid)
{
    for (int k = 1; k <= P.n; ++k) {
        // an item = four adjacent pixels of two rows from ONE set of fifteen loads, all issued before the first use (down4x2; until
        // round 4 this was one row at a time through the generic down4, whose guarded loads and per-tap arithmetic made the two
        // patches of a 4K pair cost ~45 us of a block that then walks a 10 us chain -- profiles/r04_ablation.txt batch 10)
        const int groups = (P.pw[k] + 3) / 4; // (the pitch covers the last group: it is the width rounded up to 64)
        const int per_frame = groups * ((P.ph[k] + 1) / 2);
        for (int i0 = P.first * per_frame + tid; i0 < 2 * per_frame; i0 += OFX_PATCH_ITEMS * 256) {
            uint32_t v[OFX_PATCH_ITEMS][2];
            uint8_t *dst[OFX_PATCH_ITEMS];
            bool two[OFX_PATCH_ITEMS];
#pragma unroll
            for (int u = 0; u < OFX_PATCH_ITEMS; ++u) {
                const int i = i0 + 256 * u;
                const bool on = i < 2 * per_frame;
                const int ic = on ? i : i0;
                const int f = ic >= per_frame ? 1 : 0, j = ic - f * per_frame;
                const int yp = j / groups, x0 = 4 * (j - yp * groups), y = 2 * yp;
                const uint8_t *src = k == 1 ? S.src[f] : S.base + (size_t)f * (size_t)P.frame_stride + P.off[k - 1];
                const int sp = k == 1 ? S.src_pitch[f] : P.pitch[k - 1];
                down4x2<false>(src, sp, P.ph[k - 1], P.pw[k - 1], P.pw[k], P.ph[k], x0, y, v[u][0], v[u][1]);
                dst[u] = on ? S.base + (size_t)f * (size_t)P.frame_stride + P.off[k] + (size_t)y * (size_t)P.pitch[k] + x0 : nullptr;
                two[u] = y + 1 < P.ph[k];
            }
#pragma unroll
            for (int u = 0; u < OFX_PATCH_ITEMS; ++u)
                if (dst[u]) {
                    *reinterpret_cast<uint32_t *>(dst[u]) = v[u][0];
                    if (two[u]) *reinterpret_cast<uint32_t *>(dst[u] + P.pitch[k]) = v[u][1];
                }
        }
        __syncthreads();
    }
}

// ---- a shifted corner that leaves the top-left patch: the patch pyramid of the NEXT frame, rebuilt around it -------------------
// cpu::shift_back_pyramid defines the shift for every input (OptFlowCPU.cpp:255-273): a flat or nearly singular corner gives
// pixel 0 an arbitrarily large flow, and the shifted corner of the next finer level then lies anywhere in the image.  The
// chain's planes only hold the top-left patch, so the block rebuilds the same small pyramid around the target: origin
// (ox, oy) at level 0, a multiple of max(8, 2^n), level k's plane holding columns [ox >> k, (ox >> k) + pw[k]) of the level.
// Level 1 is formed from the whole frame with its true neighbours; from level 2 on a plane's first column / row would need a
// source column / row the plane below does not hold (2x - 1 < 0), so with ox > 0 (oy > 0) local column (row) 0 of levels >= 2
// is not valid -- one pixel, not one per level: local column 1 reads local columns 1 .. 3 of the level below.  Every pixel the
// chain takes from such a plane is therefore exactly the pixel of the frame's own pyramid.
struct RelocState {
    int on;     // planes built
    int ox, oy; // level-0 origin
};

__device__ __forceinline__ void reloc_valid_box(const PatchBuild &P, const RelocState &B, int k, int w_k, int h_k, int &x0, int &x1, int &y0, int &y1)
{
    x0 = (B.ox >> k) + ((k >= 2 && B.ox > 0) ? 1 : 0);
    y0 = (B.oy >> k) + ((k >= 2 && B.oy > 0) ? 1 : 0);
    x1 = min((B.ox >> k) + P.pw[k], w_k);
    y1 = min((B.oy >> k) + P.ph[k], h_k);
}

// all 256 threads of the block; ends with a barrier.  frame: the whole level-0 next frame (w0 x h0, pitch0).
__device__ __forceinline__ void patch_build_reloc(const PatchBuild &P, const uint8_t *frame, int pitch0, int w0, int h0, uint8_t *base, int ox,
                                                  int oy, int tid)
{
    for (int k = 1; k <= P.n; ++k) {
        const int groups = (P.pw[k] + 3) / 4;
        const int total = groups * P.ph[k];
        for (int i0 = tid; i0 < total; i0 += 4 * 256) {
            uint32_t v[4];
            uint8_t *dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                const bool on = i < total;
                const int j = on ? i : i0;
                const int y = j / groups, x0 = 4 * (j - y * groups);
                if (k == 1) // global coordinates on the whole frame: true neighbours on every side, the image border where it is
                    v[u] = down4(frame, pitch0, 0, 0, h0, w0, w0 >> 1, (ox >> 1) + x0, (oy >> 1) + y);
                else
                    v[u] = down4(base + P.off[k - 1], P.pitch[k - 1], 0, 0, P.ph[k - 1], P.pw[k - 1], P.pw[k], x0, y);
                dst[u] = on ? base + P.off[k] + (size_t)y * (size_t)P.pitch[k] + x0 : nullptr;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u]) *reinterpret_cast<uint32_t *>(dst[u]) = v[u];
        }
        __syncthreads();
    }
}

// The chain's pixels come from LDS: before the walk, the wave copies the top-left corner of every level -- 16 x 16 bytes of
// prev (the window and its 3x3 stencils reach column/row radius + 1 <= 13) and 32 x 32 bytes of next (the same plus
// >= 18 pixels of shift) -- with all its loads in flight at once, i.e. ONE memory round trip instead of one per level
// (the chain was 13 us of mostly memory latency; a rank of an 8-way sharded 4K pair spends less than that on LK).  A
// shifted target outside the cached corner falls back to a global load.
constexpr int kCornerPrevDim = 16, kCornerNextDim = 32;
constexpr int kCornerCacheBytes = kCornerPrevDim * kCornerPrevDim + kCornerNextDim * kCornerNextDim; // per level
constexpr int kCornerTileBytes = 2 * kCornerPrevDim * kCornerPrevDim;                                 // the resolved tiles
constexpr int kCornerScratch = 128; // LDS in front of a corner block's cache: the chain's floats, then corner_block's hand-over words
constexpr int kCornerMaxRadius = kCornerPrevDim - 3; // the cached prev corner must hold the window and its stencils

struct CornerCache {
    const uint8_t *prev; // [16][16]
    const uint8_t *next; // [32][32]
};

// a pixel of the window or of its 3x3 stencils (x, y <= radius + 1): always inside the cached corner when the planes hold it
__device__ __forceinline__ int pix(const uint8_t *cached, int dim, const CornerLevel &L, int x, int y, int &miss)
{
    const bool inside = x >= 0 && x < L.w && y >= 0 && y < L.h;
    const bool in = inside && y < L.row_end && x < L.col_end;
    miss |= (inside && !in) ? 1 : 0;
    const int cx = min(max(x, 0), dim - 1), cy = min(max(y, 0), dim - 1);
    const int v = (int)cached[cy * dim + cx];
    return in ? v : 0;
}

// cpu::shift_back_pyramid on channel 0 for one pixel (same rule as shift_1ch_kernel in pyramid.hip)
// Rl: the relocated planes of this level (plane == nullptr: none): columns [x0, x1) x rows [y0, y1) are valid, the plane's
// first byte is pixel (px, py)
struct RelocLevel {
    const uint8_t *plane;
    int pitch, px, py, x0, x1, y0, y1;
};
__device__ __forceinline__ int shifted_next(const CornerCache &C, const CornerLevel &L, const RelocLevel &Rl, int x, int y, bool shifted, float u,
                                            float v, int &miss, int &used_reloc)
{
    const bool inside = x >= 0 && x < L.w && y >= 0 && y < L.h;
    const int own = pix(C.next, kCornerNextDim, L, x, y, miss);
    const float ty = (float)y + v, tx = (float)x + u;
    const bool yin = ty > -1.0f && ty < (float)L.h;
    const int ny = yin ? (int)ty : 0;
    const bool target = shifted && yin && tx > -1.0f && tx < (float)L.w; // the target pixel exists in the image
    const int tnx = target ? (int)tx : 0;
    const bool hit = target && ny < L.row_end && tnx < L.col_end;
    const bool hit2 = target && !hit && Rl.plane != nullptr && tnx >= Rl.x0 && tnx < Rl.x1 && ny >= Rl.y0 && ny < Rl.y1;
    miss |= (inside && target && !hit && !hit2) ? 1 : 0;
    used_reloc |= (inside && hit2) ? 1 : 0;
    int moved = 0;
    if (hit) {
        if (tnx < kCornerNextDim && ny < kCornerNextDim)
            moved = (int)C.next[ny * kCornerNextDim + tnx];
        else
            moved = (int)L.next[(size_t)ny * (size_t)L.pitch + tnx]; // far shift: outside the cached corner
    } else if (hit2) {
        moved = (int)Rl.plane[(size_t)(ny - Rl.py) * (size_t)Rl.pitch + (tnx - Rl.px)];
    }
    const bool keep = 3ll * ((long long)y * L.w + x) < (long long)L.w * (long long)L.h;
    const int val = !shifted ? own : ((hit || hit2) ? moved : (keep ? own : 0));
    return inside ? val : 0;
}

// Copies the corners of every level into `cache` (levels * kCornerCacheBytes bytes of LDS private to this wave).  Rows
// and dwords outside the planes are stored as zero (they are never selected: pix() tests the extents).
__device__ __forceinline__ void corner_prefetch(const CornerHead &A, const CornerLevel *lv, int lane, uint8_t *cache)
{
    // Phase 1 issues every load of every level (branch-free: address clamped into the planes); phase 2, behind a
    // scheduling barrier, masks the values and stores them.  Without the split hipcc puts a wait behind each load.
    uint32_t pv[OFX_MAX_LEVELS], nv[OFX_MAX_LEVELS][4];
#pragma unroll
    for (int k = 0; k < OFX_MAX_LEVELS; ++k) {
        pv[k] = nv[k][0] = nv[k][1] = nv[k][2] = nv[k][3] = 0u;
        if (k >= A.levels) continue; // uniform
        const CornerLevel &L = lv[k];
        const int rows = min(L.h, L.row_end), last_c = L.pitch - 4;
        pv[k] = *reinterpret_cast<const uint32_t *>(L.prev + (size_t)min(lane >> 2, rows - 1) * (size_t)L.pitch + min((lane & 3) * 4, last_c));
#pragma unroll
        for (int i = 0; i < 4; ++i) { // 32 rows x 8 dwords
            const int idx = lane + 64 * i;
            nv[k][i] = *reinterpret_cast<const uint32_t *>(L.next + (size_t)min(idx >> 3, rows - 1) * (size_t)L.pitch + min((idx & 7) * 4, last_c));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < OFX_MAX_LEVELS; ++k) {
        if (k >= A.levels) continue;
        const CornerLevel &L = lv[k];
        const int rows = min(L.h, L.row_end), last_c = L.pitch - 4;
        uint8_t *pc = cache + k * kCornerCacheBytes, *nc = pc + kCornerPrevDim * kCornerPrevDim;
        {
            const int r = lane >> 2, c = (lane & 3) * 4; // 16 rows x 4 dwords
            *reinterpret_cast<uint32_t *>(pc + r * kCornerPrevDim + c) = (r < rows && c <= last_c) ? pv[k] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, r = idx >> 3, c = (idx & 7) * 4;
            *reinterpret_cast<uint32_t *>(nc + r * kCornerNextDim + c) = (r < rows && c <= last_c) ? nv[k][i] : 0u;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One level of the chain, by one wave: shift vector from the corner flows found so far, the resolved tiles, the window sums of
// pixel 0, the solve.  Returns (wave-uniform) whether a pixel inside the image was needed that neither the planes nor the
// relocated planes Rl hold.  `final` = false: the caller can repair a miss -- the level then stops at the miss (nothing but the
// shift vector is published, which does not depend on the pixels) and is run again on relocated planes; true: a miss raises
// bit k of the status words.  u_out, v_out = the level's shift (every lane holds the same).
template <int MODE, bool FAST>
__device__ __forceinline__ bool corner_level(const CornerHead &A, const CornerLevel *lv, int k, int lane, float *f0, uint8_t *tileP, uint8_t *tileQ,
                                             uint8_t *cache, const RelocLevel &Rl, bool final, int &word, float &u_out, float &v_out)
{
    const CornerLevel &L = lv[k];
    const CornerCache C{cache + k * kCornerCacheBytes, cache + k * kCornerCacheBytes + kCornerPrevDim * kCornerPrevDim};
    // shift vector: float accumulation, coarsest level first (OptFlowCPU.cpp:257-266)
    float u = 0.0f, v = 0.0f;
    for (int j = A.levels - 1; j > k; --j) {
        const float mult = (float)(1 << (j - k));
        u += mult * f0[2 * j];
        v += mult * f0[2 * j + 1];
    }
    u_out = u;
    v_out = v;
    const bool shifted = k != A.levels - 1;
    if (shifted && lane == 0) {
        A.uv[2 * k] = u;
        A.uv[2 * k + 1] = v;
        // a row-sharded level kernel reads next at row (int)(y + v) for the rows y its stencils touch: targets inside the
        // image must be rows the shard holds (targets outside the image read nothing; a NaN shift moves nothing)
        if (L.need1 > L.need0 && v == v) {
            const float t0 = (float)L.need0 + v, t1 = (float)(L.need1 - 1) + v; // the map is monotone: its two ends decide
            if (t1 > -1.0f && t0 < (float)L.h) {
                const int lo = max(0, (int)floorf(t0)), hi = min(L.h - 1, (int)floorf(t1));
                if (lo <= hi && (lo < L.valid0 || hi >= L.valid1)) {
                    word |= 1 << (8 + k);
                    if (A.status != nullptr) atomicOr(A.status, 1 << (8 + k));
                }
            }
        }
    }
    // Stage 1: every pixel the window's 3x3 stencils can touch -- x, y in [-1, radius+1] -- is resolved ONCE (border
    // rule, shift, patch extents) into two small LDS tiles, prev and shifted next, indexed by coordinate + 1.  A lone wave
    // issues an instruction every ~6 cycles, so the chain is bound by its instruction count: resolving the 9 neighbours
    // inside every tap cost ~1.7k instructions per level, this costs ~0.4k.
    const int rdim = A.radius + 3; // <= 16
    int miss = 0, used = 0;
    for (int i = lane; i < rdim * rdim; i += 64) {
        const int rx = i % rdim, ry = i / rdim;
        tileP[ry * kCornerPrevDim + rx] = (uint8_t)pix(C.prev, kCornerPrevDim, L, rx - 1, ry - 1, miss);
        tileQ[ry * kCornerPrevDim + rx] = (uint8_t)shifted_next(C, L, Rl, rx - 1, ry - 1, shifted, u, v, miss, used);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool any_miss = __any(miss != 0) != 0;
    if (!final && any_miss) return true; // (the level is run again on relocated planes: no need to finish this pass)
    // Stage 2: window of pixel 0, clipped to the image: taps [0..R] x [0..R]
    const int tw = min(A.radius + 1, L.w), th = min(A.radius + 1, L.h);
    int sxx = 0, syy = 0, sxy = 0, sxt = 0, syt = 0;
    for (int t = lane; t < tw * th; t += 64) {
        const int x = t % tw, y = t / tw;
        int p[3][3], q[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                p[i][j] = tileP[(y + i) * kCornerPrevDim + x + j]; // pixel (x - 1 + j, y - 1 + i)
                q[i][j] = tileQ[(y + i) * kCornerPrevDim + x + j];
            }
        int ix = (p[0][2] + 2 * p[1][2] + p[2][2]) - (p[0][0] + 2 * p[1][0] + p[2][0]); // Dx_3x3, kernels.cpp:6-10
        int iy = (p[2][0] + 2 * p[2][1] + p[2][2]) - (p[0][0] + 2 * p[0][1] + p[0][2]); // Dy_3x3, kernels.cpp:15-19
        int it;
        if constexpr (MODE == OFX_MODE_LK_FLOAT) {
            // Dt_3x3 (kernels.cpp:20-24) on next - prev
            int d[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) d[i][j] = q[i][j] - p[i][j];
            it = (d[0][0] + d[0][2] + d[2][0] + d[2][2]) + 2 * (d[0][1] + d[1][0] + d[1][2] + d[2][1]) + 3 * d[1][1];
        } else {
            // per-tap truncated Gaussian (OptFlowCPU.cpp:102 with GAUS_KERNEL_3x3), u8 wrap (:106, :15)
            const int gp = (p[0][0] >> 4) + (p[0][2] >> 4) + (p[2][0] >> 4) + (p[2][2] >> 4) + (p[0][1] >> 3) + (p[1][0] >> 3) +
                           (p[1][2] >> 3) + (p[2][1] >> 3) + (p[1][1] >> 2);
            const int gq = (q[0][0] >> 4) + (q[0][2] >> 4) + (q[2][0] >> 4) + (q[2][2] >> 4) + (q[0][1] >> 3) + (q[1][0] >> 3) +
                           (q[1][2] >> 3) + (q[2][1] >> 3) + (q[1][1] >> 2);
            ix &= 0xff;
            iy &= 0xff;
            it = (gq - gp) & 0xff;
        }
        sxx += ix * ix;
        syy += iy * iy;
        sxy += ix * iy;
        sxt += ix * it;
        syt += iy * it;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sxx += __shfl_xor(sxx, m);
        syy += __shfl_xor(syy, m);
        sxy += __shfl_xor(sxy, m);
        sxt += __shfl_xor(sxt, m);
        syt += __shfl_xor(syt, m);
    }
    float fu, fv;
    solve2x2<MODE, FAST>(sxx, syy, sxy, sxt, syt, SolveOpts{A.min_det}, fu, fv); // every lane holds the same sums
    if (any_miss) {
        word |= 1 << k;
        if (A.status != nullptr && lane == 0) atomicOr(A.status, 1 << k);
    }
    if (__any(used != 0)) word |= kCornerRepaired;
    if (lane == 0) {
        f0[2 * k] = fu;
        f0[2 * k + 1] = fv;
        if (L.flow != nullptr && L.flow_row0 == 0) {
            L.flow[0] = fu;
            L.flow[1] = fv;
        }
    }
    // LDS operations of one wave execute in order; the fence only stops the compiler from moving the next
    // level's reads of f0 above the store
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return any_miss;
}

// One wave walks the pyramid coarse to fine.  LDS private to this wave: f0 = 2*OFX_MAX_LEVELS floats (the corner flows
// found so far) and cache = kCornerTileBytes + levels * kCornerCacheBytes bytes (two resolved tiles, then corner_prefetch's
// corners).  Only wave-level ordering is needed, so the function can run inside a larger workgroup.  A pixel the planes do
// not hold raises bit k of the status words (no repair: see corner_block).
template <int MODE, bool FAST = false>
__device__ __forceinline__ void corner_wave(const CornerHead &A, const CornerLevel *lv, int lane, float *f0, uint8_t *cache)
{
    uint8_t *tileP = cache, *tileQ = cache + kCornerPrevDim * kCornerPrevDim;
    cache += kCornerTileBytes;
    corner_prefetch(A, lv, lane, cache);
    int word = 0;
    float u, v;
    const RelocLevel none{nullptr, 0, 0, 0, 0, 0, 0, 0};
    for (int k = A.levels - 1; k >= 0; --k) (void)corner_level<MODE, FAST>(A, lv, k, lane, f0, tileP, tileQ, cache, none, true, word, u, v);
    if (A.pair_status != nullptr && lane == 0) *A.pair_status = word;
}

// The chain inside a block of 256 threads that can REPAIR a miss (A.reloc != nullptr; two-stage stream pipeline and
// local_corner sessions on borrowed frames): wave 0 walks the chain; when a level needs a pixel of the next frame's pyramid
// that the top-left patch does not hold, all four waves rebuild that pyramid around the shifted corner (patch_build_reloc) and
// wave 0 runs the level again on it.  The result is then the reference's for every input, with no host round trip.
// xch: 4 ints of LDS shared by the block (beyond the wave's own scratch).
template <int MODE, bool FAST = false>
__device__ __forceinline__ void corner_block(const CornerHead &A, const CornerLevel *lv, const PatchBuild &P, int tid, int wv, float *f0, uint8_t *cache,
                                             int *xch)
{
    const int lane = tid & 63; // (wv: the wave's index in the block, wave-uniform -- the caller's readfirstlane)
    uint8_t *tileP = cache, *tileQ = cache + kCornerPrevDim * kCornerPrevDim;
    cache += kCornerTileBytes;
    if (wv == 0) corner_prefetch(A, lv, lane, cache);
    RelocState B{0, 0, 0};
    int word = 0;
    const CornerLevel &L0 = lv[0];
    const int align = max(8, 1 << P.n);
    // One call site for the level (a state machine instead of "run, repair, run again"): hipcc 7.2 fails with "illegal VGPR to
    // SGPR copy" in the stream kernel's LK branch when the chain is inlined twice into the kernel.
    int k = A.levels - 1, pass = 0;
    bool retried = false;
    while (k >= 0) {
        int *const slot = xch + 4 * (pass & 1); // (two hand-over slots in turn: one barrier per pass is enough)
        ++pass;
        if (wv == 0) {
            RelocLevel r{nullptr, 0, 0, 0, 0, 0, 0, 0};
            if (B.on && k >= 1) {
                r.plane = A.reloc + P.off[k];
                r.pitch = P.pitch[k];
                r.px = B.ox >> k;
                r.py = B.oy >> k;
                reloc_valid_box(P, B, k, lv[k].w, lv[k].h, r.x0, r.x1, r.y0, r.y1);
            }
            // (level 0 is the whole frame when the repair is on: nothing to repair there; the top level is not shifted)
            const bool can_repair = A.reloc != nullptr && !retried && k >= 1 && k < A.levels - 1;
            float u = 0.0f, v = 0.0f;
            const bool miss = corner_level<MODE, FAST>(A, lv, k, lane, f0, tileP, tileQ, cache, r, !can_repair, word, u, v);
            if (lane == 0) {
                slot[0] = (miss && can_repair) ? 1 : 0;
                slot[1] = __float_as_int(u);
                slot[2] = __float_as_int(v);
            }
        }
        __syncthreads();
        const bool repair = __builtin_amdgcn_readfirstlane(slot[0]) != 0; // block-uniform
        if (!repair) {
            --k;
            retried = false;
            continue;
        }
        // centre the patch on the middle of the shifted corner: target columns (int)(x + u) for x in [-1, radius + 1]
        const float fu = __int_as_float(__builtin_amdgcn_readfirstlane(slot[1])), fv = __int_as_float(__builtin_amdgcn_readfirstlane(slot[2]));
        const float half = 0.5f * (float)A.radius;
        const float cx = fminf(fmaxf((fu + half) * (float)(1 << k), -1.0e9f), 1.0e9f), cy = fminf(fmaxf((fv + half) * (float)(1 << k), -1.0e9f), 1.0e9f);
        int ox = (int)cx - P.pw[0] / 2, oy = (int)cy - P.ph[0] / 2;
        ox = __builtin_amdgcn_readfirstlane(min(max(ox, 0), L0.w - P.pw[0]) / align * align);
        oy = __builtin_amdgcn_readfirstlane(min(max(oy, 0), L0.h - P.ph[0]) / align * align);
        patch_build_reloc(P, L0.next, L0.pitch, L0.w, L0.h, A.reloc, ox, oy, tid); // ends with a barrier
        B = RelocState{1, ox, oy};
        retried = true; // the same level again, on the relocated planes; a miss is final now
    }
    if (wv == 0 && A.pair_status != nullptr && lane == 0) *A.pair_status = word;
}

} // namespace ofx_dev
