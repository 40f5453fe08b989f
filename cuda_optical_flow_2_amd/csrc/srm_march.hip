// gpu::srm_1ch / cpu::srm_1ch as a MARCH (round 4): dst(y, x) = sum over the ww x wh window around (y, x), clipped
// at the image border, of a * b (OptFlowGpu.cu:1463-1502, OptFlowCPU.cpp:162-200).  The stand-alone entry point used to be the
// reference's own shape -- one thread per pixel, ww * wh taps each, 2 * ww * wh byte loads per pixel: 317 us per 4K plane at 9x9, 0.02
// of the HBM roofline on its 6 B/px.  Integer sums are exact and do not depend on the order of their terms (int32 wraps like the
// reference's int accumulator), so the window slides:
//   * a wave walks down a strip of rows of a 256-column tile, a lane owns 4 adjacent columns; per step it fetches the row entering
//     the vertical window and the row leaving it (one dword of each plane per lane, a step ahead) and updates four running sums
//     V[c] += a_in * b_in - a_out * b_out (byte products by SDWA multiplies);
//   * the horizontal window goes through the wave's private LDS row: every lane writes its four V, reads the ww + 3 values its four
//     outputs cover and slides -- out0 = the first ww, out1 = out0 - v[0] + v[ww], ... -- in plain 32-bit adds;
//   * one 16-byte store per lane and row: 1 KB per wave, gap-free.
// Algorithmic bytes per call: 2 u8 read + one int32 written = 6 B/px; five calls per level = 30.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "ofx.h"
#include "ofx_internal.h"

namespace {

constexpr int kSrmPad = 64;                           // ints behind a wave's LDS row: the reads of the lanes past the tile's last output
constexpr int kSrmWaveInts = 256 + kSrmPad;
constexpr int kSrmOob = (int)0x80000000;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t srm_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00027000);
}

// byte j of x times byte j of y
template <int J>
__device__ __forceinline__ int mul_byte(uint32_t x, uint32_t y)
{
    return (int)(((x >> (8 * J)) & 0xffu) * ((y >> (8 * J)) & 0xffu)); // (selects as v_mul_u32_u24_sdwa BYTE_J x BYTE_J)
}

// lane l receives lane l - 1's / lane l + 1's value (zero at the wave's ends): DPP wave shifts, as the LK march's derivative stage
__device__ __forceinline__ int srm_from_left(int x)
{
    int r = __builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int srm_from_right(int x)
{
    int r = __builtin_amdgcn_update_dpp(0, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}

struct SrmArgs {
    const uint8_t *a, *b;
    int32_t *dst;
    int w, h, ww, wh;
    int tiles_x, strips, strip_h, out_w; // out_w: output columns of a tile (a multiple of 4)
};

// WW > 0: the horizontal window's width at compile time (its loop unrolled); 0: A.ww at run time
template <int WW>
__global__ __launch_bounds__(256) void srm_u8_march_kernel(const SrmArgs A)
{
    __shared__ __attribute__((aligned(16))) int lds[4 * kSrmWaveInts];
    const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int item = (int)blockIdx.x * 4 + wv;
    if (item >= A.tiles_x * A.strips) return;
    const int tile = item % A.tiles_x, strip = item / A.tiles_x;
    const int ww = WW > 0 ? WW : A.ww, wh = A.wh, w = A.w, h = A.h;
    const int ox = ww >> 1, oy = wh >> 1, ry = wh - 1 - oy;
    const int x0 = tile * A.out_w - ox; // image column of the wave's LDS index 0
    const int cb = x0 + 4 * lane;       // this lane's first column
    const int ys = strip * A.strip_h, ye = min(ys + A.strip_h, h);
    int *row = lds + wv * kSrmWaveInts;
    if (lane < kSrmPad / 4) *(int4 *)(row + 256 + 4 * lane) = int4{0, 0, 0, 0};

    const __amdgpu_buffer_rsrc_t ra = srm_rsrc(A.a, (unsigned)w * (unsigned)h), rb = srm_rsrc(A.b, (unsigned)w * (unsigned)h);
    const __amdgpu_buffer_rsrc_t rd = srm_rsrc(A.dst, (unsigned)w * (unsigned)h * 4u);
    // The planes are tightly packed (w bytes per row, any w >= 4): the dword is fetched from a base column clamped into the row and
    // its bytes put in place by a per-lane selector, zero where the column lies outside the image.
    const int cbl = min(max(cb, 0), w - 4);
    uint32_t sel = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = cb + j;
        const uint32_t sj = (x >= 0 && x < w && x - cbl >= 0 && x - cbl < 4) ? (uint32_t)(x - cbl) : 0x0cu;
        sel |= sj << (8 * j);
    }
    const uint32_t voff = (uint32_t)cbl;
    auto row_off = [&](int y) -> int { return (uint32_t)y < (uint32_t)h ? y * w : kSrmOob; };
    auto fetch = [&](int y, uint32_t &pa, uint32_t &pb) {
        const int o = row_off(y);
        pa = __builtin_amdgcn_raw_buffer_load_b32(ra, voff, o, 0);
        pb = __builtin_amdgcn_raw_buffer_load_b32(rb, voff, o, 0);
    };
    auto place = [&](uint32_t raw) -> uint32_t { return __builtin_amdgcn_perm(0u, raw, sel); };

    int V[4] = {0, 0, 0, 0};
    auto add_row = [&](uint32_t pa, uint32_t pb) {
        const uint32_t xa = place(pa), xb = place(pb);
        V[0] += mul_byte<0>(xa, xb);
        V[1] += mul_byte<1>(xa, xb);
        V[2] += mul_byte<2>(xa, xb);
        V[3] += mul_byte<3>(xa, xb);
    };
    auto sub_row = [&](uint32_t pa, uint32_t pb) {
        const uint32_t xa = place(pa), xb = place(pb);
        V[0] -= mul_byte<0>(xa, xb);
        V[1] -= mul_byte<1>(xa, xb);
        V[2] -= mul_byte<2>(xa, xb);
        V[3] -= mul_byte<3>(xa, xb);
    };
    // The wave is alone with its memory round trips (two or three waves per SIMD at 4K, a step's arithmetic is ~60 instructions), so
    // what it waits for is the NUMBER of round trips, not the bytes: the rows of DEPTH steps are in flight at once -- a group of
    // DEPTH steps takes its rows from registers while the next group's 4 * DEPTH loads are on their way (round 4, second session:
    // one step ahead and one load per priming row was 13.4 us per 4K plane at 9x9; strips of 16 rows then took 8 + 16 round trips).
    constexpr int DEPTH = 8;
    // the store: outputs x0 + ox + 4 lane + j, j < nval
    const int xo = x0 + ox + 4 * lane;
    const int nval = max(0, min(4, min(A.out_w - 4 * lane, w - xo)));
    const uint32_t st_off = nval == 4 ? (uint32_t)xo * 4u : (uint32_t)kSrmOob;
    const bool ragged = __any(nval > 0 && nval < 4) != 0;
    uint32_t ia[DEPTH], ib[DEPTH], oa[DEPTH], ob[DEPTH];
    // the first group's rows are issued BEFORE the priming rows are waited for: everything a strip needs up to its first DEPTH
    // outputs is then one batch of loads when the window has no more than DEPTH + 1 rows
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) fetch(ys + t + ry, ia[t], ib[t]), fetch(ys + t - oy, oa[t], ob[t]);
    // priming: rows ys - oy .. ys + ry - 1 (rows outside the image read as zeros), DEPTH rows' loads in flight at a time
    for (int y = ys - oy; y < ys + ry; y += DEPTH) {
        uint32_t pa[DEPTH], pb[DEPTH];
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) fetch(y + t < ys + ry ? y + t : -1, pa[t], pb[t]); // (row -1 reads zeros: adds nothing)
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) add_row(pa[t], pb[t]);
    }
    for (int yg = ys; yg < ye; yg += DEPTH) {
        uint32_t nia[DEPTH], nib[DEPTH], noa[DEPTH], nob[DEPTH]; // the rows of the next group
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) fetch(yg + DEPTH + t + ry, nia[t], nib[t]), fetch(yg + DEPTH + t - oy, noa[t], nob[t]);
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) {
        const int y = yg + t;
        if (y >= ye) break; // (wave-uniform)
        add_row(ia[t], ib[t]); // V = the vertical window of row y
        *(int4 *)(row + 4 * lane) = int4{V[0], V[1], V[2], V[3]};
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int *src = row + 4 * lane; // this lane's four windows start at LDS indices 4 lane + j
        int acc = 0;
        if constexpr (WW > 0) {
#pragma unroll
            for (int k = 0; k < WW; ++k) acc += src[k];
        } else {
            for (int k = 0; k < ww; ++k) acc += src[k];
        }
        const int o0 = acc;
        const int o1 = o0 - src[0] + src[ww];
        const int o2 = o1 - src[1] + src[ww + 1];
        const int o3 = o2 - src[2] + src[ww + 2];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier(); // (the next step's write must not overtake these reads: in order within a wave anyway)
        const int so = __builtin_amdgcn_readfirstlane(y * w * 4);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o0, (uint32_t)o1, (uint32_t)o2, (uint32_t)o3}, rd, st_off, so, 2 /* nt */);
        if (__builtin_expect(ragged, 0)) { // the lane at the image's right edge: one to three pixels
            const int ov[4] = {o0, o1, o2, o3};
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_raw_buffer_store_b32((uint32_t)ov[j], rd, (nval < 4 && j < nval) ? (uint32_t)(xo + j) * 4u : (uint32_t)kSrmOob, so, 2);
        }
        sub_row(oa[t], ob[t]); // row y - oy leaves
        }
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) ia[t] = nia[t], ib[t] = nib[t], oa[t] = noa[t], ob[t] = nob[t];
    }
}

// The same sums with a strip's rows fetched ALL AT ONCE (square windows of 3 ... 21, the only ones the reference calls with:
// OptFlowCPU.cpp:344-345 9x9, OptFlowGpu.cu:1944-1945 19x19).  A strip is kSrmRows - (WH - 1) output rows, so the rows its windows
// cover are exactly kSrmRows: 2 x 32 dword loads go out back to back, the wave pays ONE memory round trip instead of one per group
// of eight rows, and the row that leaves a window is the register that entered it WH steps earlier -- no second fetch.  With the
// window's height at compile time the whole strip is straight-line code.
constexpr int kSrmRows = 32;
#ifndef OFX_SRM_NEIGH
#define OFX_SRM_NEIGH 1 // (0: every window through LDS)
#endif

template <int WW, int WH>
__global__ __launch_bounds__(256) void srm_u8_strip_kernel(const SrmArgs A)
{
    __shared__ __attribute__((aligned(16))) int lds[4 * kSrmWaveInts];
    const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int item = (int)blockIdx.x * 4 + wv;
    if (item >= A.tiles_x * A.strips) return;
    const int tile = item % A.tiles_x, strip = item / A.tiles_x;
    constexpr int ww = WW, wh = WH, SH = kSrmRows - (WH - 1);
    const int w = A.w, h = A.h;
    constexpr int ox = ww >> 1, oy = wh >> 1;
    // NEIGH (windows up to 9 columns): a lane's four outputs are the columns of its own four sums, and their windows reach no further
    // than the neighbouring lanes' -- eight DPP moves instead of the round trip through LDS (a write, a wait, three 16-byte reads: the
    // longest dependent stretch of a step, with one or two waves per SIMD to hide it).  Lanes 1 .. 62 produce outputs: 248 columns.
    constexpr bool NEIGH = OFX_SRM_NEIGH && WW <= 9;
    const int x0 = tile * A.out_w - (NEIGH ? 4 : ox);
    const int cb = x0 + 4 * lane;
    const int ys = strip * SH, ye = min(ys + SH, h);
    int *row = lds + wv * kSrmWaveInts;
    if (lane < kSrmPad / 4) *(int4 *)(row + 256 + 4 * lane) = int4{0, 0, 0, 0};

    const __amdgpu_buffer_rsrc_t ra = srm_rsrc(A.a, (unsigned)w * (unsigned)h), rb = srm_rsrc(A.b, (unsigned)w * (unsigned)h);
    const __amdgpu_buffer_rsrc_t rd = srm_rsrc(A.dst, (unsigned)w * (unsigned)h * 4u);
    const int cbl = min(max(cb, 0), w - 4);
    uint32_t sel = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = cb + j;
        const uint32_t sj = (x >= 0 && x < w && x - cbl >= 0 && x - cbl < 4) ? (uint32_t)(x - cbl) : 0x0cu;
        sel |= sj << (8 * j);
    }
    const uint32_t voff = (uint32_t)cbl;
    uint32_t pa[kSrmRows], pb[kSrmRows];
#pragma unroll
    for (int t = 0; t < kSrmRows; ++t) {
        const int y = ys - oy + t;
        const int o = (uint32_t)y < (uint32_t)h ? y * w : kSrmOob; // (a row outside the image reads zeros)
        pa[t] = __builtin_amdgcn_raw_buffer_load_b32(ra, voff, o, 0);
        pb[t] = __builtin_amdgcn_raw_buffer_load_b32(rb, voff, o, 0);
    }
    const int xo = NEIGH ? x0 + 4 * lane : x0 + ox + 4 * lane; // the lane's first output column
    const int nval = NEIGH ? ((lane >= 1 && lane <= 62) ? max(0, min(4, min(A.out_w - 4 * (lane - 1), w - xo))) : 0)
                           : max(0, min(4, min(A.out_w - 4 * lane, w - xo)));
    const uint32_t st_off = nval == 4 ? (uint32_t)xo * 4u : (uint32_t)kSrmOob;
    const bool ragged = __any(nval > 0 && nval < 4) != 0;
    int V[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < kSrmRows; ++t) {
        {
            const uint32_t xa = __builtin_amdgcn_perm(0u, pa[t], sel), xb = __builtin_amdgcn_perm(0u, pb[t], sel);
            V[0] += mul_byte<0>(xa, xb), V[1] += mul_byte<1>(xa, xb), V[2] += mul_byte<2>(xa, xb), V[3] += mul_byte<3>(xa, xb);
        }
        if (t < wh - 1) continue; // (compile time) the window is not complete yet
        const int y = ys + t - (wh - 1);
        if (y >= ye) break; // (wave-uniform)
        int o0, o1, o2, o3;
        if constexpr (NEIGH) {
            // E[e] = the vertical sum of column (4 lane - 4 + e): the left neighbour's four, this lane's, the right neighbour's
            int E[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) E[j] = srm_from_left(V[j]), E[4 + j] = V[j], E[8 + j] = srm_from_right(V[j]);
            int acc = 0;
#pragma unroll
            for (int e = 4 - ox; e <= 4 + ox; ++e) acc += E[e];
            o0 = acc;
            o1 = o0 - E[4 - ox] + E[5 + ox];
            o2 = o1 - E[5 - ox] + E[6 + ox];
            o3 = o2 - E[6 - ox] + E[7 + ox];
        } else {
        *(int4 *)(row + 4 * lane) = int4{V[0], V[1], V[2], V[3]};
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int *src = row + 4 * lane;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < WW; ++k) acc += src[k];
        o0 = acc;
        o1 = o0 - src[0] + src[ww];
        o2 = o1 - src[1] + src[ww + 1];
        o3 = o2 - src[2] + src[ww + 2];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }
        const int so = __builtin_amdgcn_readfirstlane(y * w * 4);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o0, (uint32_t)o1, (uint32_t)o2, (uint32_t)o3}, rd, st_off, so, 2 /* nt */);
        if (__builtin_expect(ragged, 0)) {
            const int ov[4] = {o0, o1, o2, o3};
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_raw_buffer_store_b32((uint32_t)ov[j], rd, (nval < 4 && j < nval) ? (uint32_t)(xo + j) * 4u : (uint32_t)kSrmOob, so, 2);
        }
        {
            const uint32_t xa = __builtin_amdgcn_perm(0u, pa[t - (wh - 1)], sel), xb = __builtin_amdgcn_perm(0u, pb[t - (wh - 1)], sel);
            V[0] -= mul_byte<0>(xa, xb), V[1] -= mul_byte<1>(xa, xb), V[2] -= mul_byte<2>(xa, xb), V[3] -= mul_byte<3>(xa, xb);
        }
    }
}

// ---- gpu::srm_1ch_float (OptFlowGpu.cu:1549-1588): float planes, a float accumulator fed in ROW-MAJOR tap order -----------------------
// The order is part of the result (float addition does not associate; sums beyond 2^24 are the rule: SURVEY 8a row 8), so nothing
// slides here: every output still adds its ww * wh products one after the other, top row first, left to right -- but the products are
// formed ONCE per pixel (the plain kernel forms each of them ww * wh times and fetches 2 * ww * wh floats per output from global
// memory).  A wave walks down a strip of a 256-column tile; the products of the newest image row go into slot (row mod wh) of a ring
// of wh rows in the wave's LDS; a lane then accumulates its four adjacent outputs: per window row it reads the ww + 3 products they
// cover and feeds four independent accumulators in tap order (plain v_add_f32: the cheap issue class); four output rows per step share
// the product rows they have in common (each row of products is read from LDS once per step).  A tap outside the image is
// skipped by the reference; here it adds the +0.0f stored for it, which leaves a float accumulator that started at +0.0f unchanged
// bit for bit (it can never be -0.0f).  12 B/px per call (2 floats read, one written).
struct SrmFArgs {
    const float *a, *b;
    float *dst;
    int w, h, ww, wh;
    int tiles_x, strips, strip_h, out_w;
};

constexpr int kSrmFRps = 4;          // output rows per step of the float march
constexpr int kSrmFRow = 256 + 64; // floats per ring row (the reads of the lanes past the tile's last output land in the pad)

// NW = 1: one wave per block.  NW = 2 (round 4, second session; windows of 13 rows and more): TWO waves share one ring -- a block's LDS
// is what limits how many waves a CU holds (the ring of a 19x19 window is 28 KB: one wave per SIMD, which issues v_add_f32 at half the
// rate two do), and two waves on one ring of wh + 2 * RPS - 1 rows need 33 KB together.  A step then produces 2 * RPS output rows: wave v
// puts the product rows y + ry + v * RPS + o and accumulates the output rows y + v * RPS + o; a block barrier separates the puts from
// the accumulation and the accumulation from the next step's puts.
template <int WW, int NW = 1>
__global__ __launch_bounds__(64 * NW) void srm_f32_march_kernel(const SrmFArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float ring_all[]; // (wh + NW * RPS - 1) rows x kSrmFRow
    const int lane = (int)threadIdx.x & 63, wv = NW > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int item = (int)blockIdx.x;
    if (item >= A.tiles_x * A.strips) return; // (block-uniform)
    auto block_sync = [&] {
        if constexpr (NW > 1) __syncthreads();
    };
    const int tile = item % A.tiles_x, strip = item / A.tiles_x;
    const int ww = WW > 0 ? WW : A.ww, wh = A.wh, w = A.w, h = A.h;
    const int ox = ww >> 1, oy = wh >> 1, ry = wh - 1 - oy;
    const int x0 = tile * A.out_w - ox;
    const int cb = x0 + 4 * lane;
    const int ys = strip * A.strip_h, ye = min(ys + A.strip_h, h);
    float *ring = ring_all; // (one ring per block)
    for (int r = wv; r < wh + NW * kSrmFRps - 1; r += NW)
        if (lane < 16) *(float4 *)(ring + r * kSrmFRow + 256 + 4 * lane) = float4{0.0f, 0.0f, 0.0f, 0.0f};

    const __amdgpu_buffer_rsrc_t ra = srm_rsrc(A.a, (unsigned)w * (unsigned)h * 4u), rb = srm_rsrc(A.b, (unsigned)w * (unsigned)h * 4u);
    const __amdgpu_buffer_rsrc_t rd = srm_rsrc(A.dst, (unsigned)w * (unsigned)h * 4u);
    // per column: its byte offset in a row, or the out-of-range marker (reads 0) where the column lies outside the image
    uint32_t co[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) co[j] = (cb + j >= 0 && cb + j < w) ? (uint32_t)(cb + j) * 4u : (uint32_t)kSrmOob;
    const bool whole = cb >= 0 && cb + 4 <= w; // (one 16-byte load instead of four)
    auto row_off = [&](int y) -> int { return (uint32_t)y < (uint32_t)h ? y * w * 4 : kSrmOob; };
    auto fetch = [&](const __amdgpu_buffer_rsrc_t &rs, int y) -> float4 {
        const int o = row_off(y);
        if (whole) return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, co[0], o, 0));
        float4 v;
        v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, co[0], o, 0));
        v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, co[1], o, 0));
        v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, co[2], o, 0));
        v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, co[3], o, 0));
        return v;
    };
    // RPS output rows per step share their product rows: a row of products is read from LDS once and feeds the accumulators of every
    // output row whose window holds it -- each output still receives its own taps in row-major order
    constexpr int RPS = kSrmFRps;
    const int nslots = wh + NW * RPS - 1; // rows of the ring
    const int nrows_w = wh + RPS - 1;     // product rows one wave's RPS output rows cover
    auto slot_of = [&](int r) {
        int sl = r % nslots;
        return sl < 0 ? sl + nslots : sl;
    };
    auto put_row_s = [&](int y, const float4 pa, const float4 pb) {
        const bool in_img = (uint32_t)y < (uint32_t)h;
        float4 p{pa.x * pb.x, pa.y * pb.y, pa.z * pb.z, pa.w * pb.w};
        if (!in_img) p = float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co[j] == (uint32_t)kSrmOob) (&p.x)[j] = 0.0f; // (a NaN / Inf next to the border must not leak into a skipped tap)
        *(float4 *)(ring + slot_of(y) * kSrmFRow + 4 * lane) = p;
    };
    // priming, eight rows' loads in flight at a time (one wave per SIMD is all the ring's LDS allows: a load per iteration would
    // expose a memory round trip per primed row -- 18 of them for a 19-row window)
    for (int y = ys - oy + 8 * wv; y < ys + ry; y += 8 * NW) {
        float4 pa[8], pb[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) pa[t] = fetch(ra, y + t), pb[t] = fetch(rb, y + t);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (y + t < ys + ry) put_row_s(y + t, pa[t], pb[t]);
    }
    const int xo = x0 + ox + 4 * lane;
    const int nval = max(0, min(4, min(A.out_w - 4 * lane, w - xo)));
    const uint32_t st_off = nval == 4 ? (uint32_t)xo * 4u : (uint32_t)kSrmOob;
    const bool ragged = __any(nval > 0 && nval < 4) != 0;
    float4 na[RPS], nb[RPS];
#pragma unroll
    for (int o = 0; o < RPS; ++o) na[o] = fetch(ra, ys + wv * RPS + ry + o), nb[o] = fetch(rb, ys + wv * RPS + ry + o);
    for (int yb = ys; yb < ye; yb += NW * RPS) {
        const int y = yb + wv * RPS; // this wave's first output row of the step
#pragma unroll
        for (int o = 0; o < RPS; ++o) put_row_s(y + ry + o, na[o], nb[o]);
#pragma unroll
        for (int o = 0; o < RPS; ++o) na[o] = fetch(ra, y + NW * RPS + ry + o), nb[o] = fetch(rb, y + NW * RPS + ry + o); // the next step's rows, a step ahead
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        block_sync(); // (the other wave's product rows)
        float acc[RPS][4];
#pragma unroll
        for (int o = 0; o < RPS; ++o) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 0.0f;
        int slot = slot_of(y - oy);
        for (int p = 0; p < nrows_w; ++p) { // product rows y - oy + p, top to bottom
            const float *src = ring + slot * kSrmFRow + 4 * lane;
            if constexpr (WW > 0) {
                float t[WW + 3];
#pragma unroll
                for (int k = 0; k < (WW + 3 + 3) / 4; ++k) {
                    const float4 q = *(const float4 *)(src + 4 * k);
                    if (4 * k + 0 < WW + 3) t[4 * k + 0] = q.x;
                    if (4 * k + 1 < WW + 3) t[4 * k + 1] = q.y;
                    if (4 * k + 2 < WW + 3) t[4 * k + 2] = q.z;
                    if (4 * k + 3 < WW + 3) t[4 * k + 3] = q.w;
                }
                if (p >= RPS - 1 && p < wh) {
                    // the row lies in every output row's window: sixteen independent accumulators take tap q before any takes tap
                    // q + 1 (an accumulator's own order stays left to right; the distance between its dependent adds is what a lone
                    // wave needs to issue at rate)
#pragma unroll
                    for (int q = 0; q < WW; ++q) {
#pragma unroll
                        for (int o = 0; o < RPS; ++o) {
                            acc[o][0] += t[q];
                            acc[o][1] += t[q + 1];
                            acc[o][2] += t[q + 2];
                            acc[o][3] += t[q + 3];
                        }
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < RPS; ++o) {
                        if (p >= o && p - o < wh) { // (wave-uniform) this product row lies in output row y + o's window
#pragma unroll
                            for (int q = 0; q < WW; ++q) { // taps left to right, four outputs side by side
                                acc[o][0] += t[q];
                                acc[o][1] += t[q + 1];
                                acc[o][2] += t[q + 2];
                                acc[o][3] += t[q + 3];
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int o = 0; o < RPS; ++o) {
                    if (p >= o && p - o < wh) {
                        for (int q = 0; q < ww; ++q) {
                            acc[o][0] += src[q];
                            acc[o][1] += src[q + 1];
                            acc[o][2] += src[q + 2];
                            acc[o][3] += src[q + 3];
                        }
                    }
                }
            }
            slot = slot + 1 == nslots ? 0 : slot + 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        block_sync(); // (the next step's puts overwrite rows the other wave may still be reading)
#pragma unroll
        for (int o = 0; o < RPS; ++o) {
            if (y + o >= ye) break;
            const int so = __builtin_amdgcn_readfirstlane((y + o) * w * 4);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__builtin_bit_cast(uint32_t, acc[o][0]), __builtin_bit_cast(uint32_t, acc[o][1]),
                                                           __builtin_bit_cast(uint32_t, acc[o][2]), __builtin_bit_cast(uint32_t, acc[o][3])}, rd, st_off, so, 2 /* nt */);
            if (__builtin_expect(ragged, 0)) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc[o][j]), rd, (nval < 4 && j < nval) ? (uint32_t)(xo + j) * 4u : (uint32_t)kSrmOob, so, 2);
            }
        }
    }
}

int env_pos(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && atoi(e) > 0 ? atoi(e) : dflt;
}

} // namespace

// returns OFX_E_UNSUPPORTED when the shape is not one the march handles (the caller keeps the plain kernel for those)
int ofx_srm_u8_march(const uint8_t *d_a, const uint8_t *d_b, int w, int h, int ww, int wh, int32_t *d_dst, hipStream_t st)
{
    if (w < 8 || ww < 1 || wh < 1 || ww > 57 || (long long)w * h * 4 >= (1ll << 32) || (long long)w * h >= (1ll << 31)) return OFX_E_UNSUPPORTED;
    SrmArgs A{};
    A.a = d_a, A.b = d_b, A.dst = d_dst, A.w = w, A.h = h, A.ww = ww, A.wh = wh;
    A.out_w = (256 - (ww - 1)) & ~3;
    A.tiles_x = ofx_div_up(w, A.out_w);
    // square windows up to 21: a strip's rows in one batch of loads (srm_u8_strip_kernel).  OFX_SRM_STRIP=0 keeps the grouped march.
    static const bool strip_form = [] { const char *e = getenv("OFX_SRM_STRIP"); return !e || atoi(e) != 0; }();
    if (strip_form && ww == wh && (ww & 1) && ww >= 3 && ww <= 21) {
        if (OFX_SRM_NEIGH && ww <= 9) A.out_w = 248, A.tiles_x = ofx_div_up(w, A.out_w); // (lanes 1 .. 62 of a wave produce outputs)
        A.strip_h = kSrmRows - (wh - 1);
        A.strips = ofx_div_up(h, A.strip_h);
        const int nblocks = ofx_div_up(A.tiles_x * A.strips, 4);
        switch (ww) {
#define OFX_SRM_CASE(N) case N: hipLaunchKernelGGL((srm_u8_strip_kernel<N, N>), dim3((unsigned)nblocks), dim3(256), 0, st, A); break;
            OFX_SRM_CASE(3) OFX_SRM_CASE(5) OFX_SRM_CASE(7) OFX_SRM_CASE(9) OFX_SRM_CASE(11) OFX_SRM_CASE(13) OFX_SRM_CASE(15) OFX_SRM_CASE(17)
            OFX_SRM_CASE(19) OFX_SRM_CASE(21)
#undef OFX_SRM_CASE
        default: break;
        }
        OFX_HIP(hipGetLastError());
        return OFX_OK;
    }
    // strips: enough waves to fill the chip several times over (the kernel is latency-bound per wave), but long enough that the
    // wh - 1 priming rows stay a minor cost
    static const int target = env_pos("OFX_SRM_WAVES", 256 * 4 * 8);
    int strips = target / A.tiles_x;
    if (strips < 1) strips = 1;
    int strip_h = ofx_div_up(h, strips);
    static const int min_strip = env_pos("OFX_SRM_MIN_STRIP", 0);
    const int min_h = min_strip ? min_strip : 16; // (measured at 4K: 9x9 14.6 / 13.4 / 16.4 us, 19x19 19.5 / 19.4 / 24.4 us with strips of 8 / 16 / 32 rows)
    if (strip_h < min_h) strip_h = min_h;
    A.strip_h = strip_h;
    A.strips = ofx_div_up(h, strip_h);
    const int blocks = ofx_div_up(A.tiles_x * A.strips, 4);
    switch (ww) {
#define OFX_SRM_CASE(N) case N: hipLaunchKernelGGL(srm_u8_march_kernel<N>, dim3((unsigned)blocks), dim3(256), 0, st, A); break;
        OFX_SRM_CASE(3) OFX_SRM_CASE(5) OFX_SRM_CASE(7) OFX_SRM_CASE(9) OFX_SRM_CASE(11) OFX_SRM_CASE(13) OFX_SRM_CASE(15) OFX_SRM_CASE(17)
        OFX_SRM_CASE(19) OFX_SRM_CASE(21) OFX_SRM_CASE(23)
#undef OFX_SRM_CASE
    default: hipLaunchKernelGGL(srm_u8_march_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, A); break;
    }
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

int ofx_srm_f32_march(const float *d_a, const float *d_b, int w, int h, int ww, int wh, float *d_dst, hipStream_t st)
{
    if (w < 8 || ww < 1 || wh < 1 || ww > 57 || wh > 45 || (long long)w * h * 4 >= (1ll << 31)) return OFX_E_UNSUPPORTED;
    SrmFArgs A{};
    A.a = d_a, A.b = d_b, A.dst = d_dst, A.w = w, A.h = h, A.ww = ww, A.wh = wh;
    A.out_w = (256 - (ww - 1)) & ~3;
    A.tiles_x = ofx_div_up(w, A.out_w);
    static const int target = env_pos("OFX_SRM_WAVES", 256 * 4 * 8);
    int strips = target / A.tiles_x;
    if (strips < 1) strips = 1;
    int strip_h = ofx_div_up(h, strips);
    // (priming a strip costs products only -- no accumulation --, and the accumulation is what the kernel spends its time on: short
    // strips, several waves per SIMD)
    static const int min_strip = env_pos("OFX_SRMF_MIN_STRIP", 0);
    const int min_h = min_strip ? min_strip : 8;
    if (strip_h < min_h) strip_h = min_h;
    // two waves on one ring where one wave's ring would leave fewer than two waves per SIMD (OFX_SRMF_NW = 1 / 2 overrides)
    static const int nw_forced = env_pos("OFX_SRMF_NW", 0);
    int nw = nw_forced == 1 || nw_forced == 2 ? nw_forced : (wh >= 5 ? 2 : 1);
    if ((size_t)(wh + 2 * kSrmFRps - 1) * kSrmFRow * sizeof(float) > 64u * 1024u) nw = 1; // (a block's dynamic LDS)
    const int step_rows = nw * kSrmFRps;
    if (nw == 2 && strip_h < 2 * step_rows) strip_h = 2 * step_rows;
    // ONE residency round: the blocks a CU holds are set by the ring's LDS (four at 19x19), a block lives for its whole strip, and a
    // second round that fills a fraction of the chip costs a whole strip's time (measured at 4K 19x19, strips of 16 / 32 / 48 rows =
    // 2.2 / 1.1 / 0.75 rounds: 128 / 143 / 105 us) -- so the strips are made just tall enough for every block to be resident at once
    // (OFX_SRMF_MIN_STRIP set: that height instead)
    if (!min_strip) {
        static const int cus = [] {
            int dev = 0, n = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
            return n;
        }();
        const size_t lds_block = (size_t)(wh + step_rows - 1) * kSrmFRow * sizeof(float);
        int per_cu = (int)((size_t)160 * 1024 / lds_block);
        per_cu = per_cu < 1 ? 1 : (per_cu * nw > 32 ? 32 / nw : per_cu); // (at most eight waves per SIMD)
        const int strips_max = cus * per_cu / A.tiles_x;
        if (strips_max >= 1) {
            const int need = ofx_div_up(h, strips_max);
            if (need > strip_h) strip_h = need;
        }
    }
    strip_h = (strip_h + step_rows - 1) / step_rows * step_rows; // whole steps
    A.strip_h = strip_h;
    A.strips = ofx_div_up(h, strip_h);
    const int blocks = A.tiles_x * A.strips;
    const size_t lds = (size_t)(wh + step_rows - 1) * kSrmFRow * sizeof(float);
    if (nw == 2) {
        switch (ww) {
#define OFX_SRM_CASE(N) case N: hipLaunchKernelGGL((srm_f32_march_kernel<N, 2>), dim3((unsigned)blocks), dim3(128), lds, st, A); break;
            OFX_SRM_CASE(3) OFX_SRM_CASE(5) OFX_SRM_CASE(7) OFX_SRM_CASE(9) OFX_SRM_CASE(11) OFX_SRM_CASE(13) OFX_SRM_CASE(15) OFX_SRM_CASE(17)
            OFX_SRM_CASE(19) OFX_SRM_CASE(21) OFX_SRM_CASE(23)
#undef OFX_SRM_CASE
        default: hipLaunchKernelGGL((srm_f32_march_kernel<0, 2>), dim3((unsigned)blocks), dim3(128), lds, st, A); break;
        }
        OFX_HIP(hipGetLastError());
        return OFX_OK;
    }
    switch (ww) {
#define OFX_SRM_CASE(N) case N: hipLaunchKernelGGL(srm_f32_march_kernel<N>, dim3((unsigned)blocks), dim3(64), lds, st, A); break;
        OFX_SRM_CASE(3) OFX_SRM_CASE(5) OFX_SRM_CASE(7) OFX_SRM_CASE(9) OFX_SRM_CASE(11) OFX_SRM_CASE(13) OFX_SRM_CASE(15) OFX_SRM_CASE(17)
        OFX_SRM_CASE(19) OFX_SRM_CASE(21) OFX_SRM_CASE(23)
#undef OFX_SRM_CASE
    default: hipLaunchKernelGGL(srm_f32_march_kernel<0>, dim3((unsigned)blocks), dim3(64), lds, st, A); break;
    }
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
