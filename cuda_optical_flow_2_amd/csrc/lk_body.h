// Device side of the fused dense Lucas-Kanade level kernel (see lk_level.hip for the design notes).  Shared by the
// stand-alone level kernel and by the pipelined "stream" kernel, which runs it next to the other stages of a frame pair.
#pragma once

#include <type_traits>

#include "lk_solve.h"
#include "ofx_internal.h"

namespace ofx_dev {

struct LkArgs {
    const uint8_t *prev;
    const uint8_t *next;
    const float *uv; // non-null: `next` is read through the reference's global shift by (uv[0], uv[1]) (fused shift)
    float *flow;   // interleaved (u,v), 2*w floats per row, row (y - flow_row0)
    int32_t *sums; // optional: 5 planes of w ints per row (test/inspection variant)
    size_t sums_plane;
    int w, h, pitch, row0, row_end; // buffer holds global rows [row0,row_end)
    int out_y0, out_y1, flow_row0;
    int strip_h, tiles_x;
    int accumulate; // non-zero: flow += result (refinement iterations) instead of flow = result
    float min_det;  // > 0: determinant guard of the solve (lk_solve.h); <= 0: the reference
    // refinement iterations (lk_body_buf.h, ITER == 2): the march also writes warp_out = warp(warp_src, warp_scale * new flow), the
    // image the next iteration reads as `next`
    const uint8_t *warp_src;
    uint8_t *warp_out;
    float warp_scale;
    int *warp_status;    // row windows (ITER 4, 5): bit warp_status_bit is set when a tap row lay outside the window (optional)
    int warp_status_bit;
};

// one launch covers several pyramid levels: block b belongs to the last level whose first_block <= b
struct LkTable {
    LkArgs lv[OFX_MAX_LK_ITEMS];
    int first_block[OFX_MAX_LK_ITEMS + 1];
    int n;
};

// Global-memory accessors.  The level's pointers pass through pin_scalar (below), after which the compiler no longer
// knows they came from the kernarg segment and would fall back to flat_* instructions with 64-bit VALU address
// arithmetic; these casts put them back into the global address space (scalar base + 32-bit lane offset).
#define OFX_GLOBAL __attribute__((address_space(1)))
#ifndef OFX_LK_NT_STORES
#define OFX_LK_NT_STORES 1
#endif
#ifndef OFX_LK_INTERIOR_VARIANT
#define OFX_LK_INTERIOR_VARIANT 1
#endif
#ifndef OFX_LK_HBOX_SLIDE
#define OFX_LK_HBOX_SLIDE 1
#endif
#ifndef OFX_LK_FOLD_PRIMING
#define OFX_LK_FOLD_PRIMING 1 // the strip's first R - 1 rows enter through the high halves (lk_wave_impl, "Priming, folded")
#endif
#ifndef OFX_LK_HBOX_SLIDE_MAX_R
#define OFX_LK_HBOX_SLIDE_MAX_R 12 // (4: the sliding box sums only where the neighbour columns are one lane away)
#endif
#ifndef OFX_LK_PROGRESS_PRIORITY
#define OFX_LK_PROGRESS_PRIORITY 1
#endif
struct __attribute__((packed)) UnalignedU32 {
    uint32_t v;
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// (base: wave-uniform pointer; off: per-lane unsigned byte / element offset -- the cast is applied to the base so that
// the access selects as "SGPR base + 32-bit VGPR offset"; the offset is made opaque at the access because instruction
// selection only forms that addressing mode when the 32->64-bit extension sits in the same block as the access, and
// loop-invariant code motion would otherwise hoist it out of the march)
__device__ __forceinline__ uint32_t lane_off(uint32_t off)
{
    asm volatile("" : "+v"(off));
    return off;
}
// the same for an offset the caller keeps in a variable of its own for the whole march: the variable itself is made opaque,
// so no copy of it is needed per access (by value, the asm's output needs a register of its own: one v_mov per load)
__device__ __forceinline__ uint32_t lane_off_var(uint32_t &off)
{
    asm volatile("" : "+v"(off));
    return off;
}
__device__ __forceinline__ uint32_t gload_u32_var(const uint8_t *base, uint32_t &off)
{
    return *(const OFX_GLOBAL uint32_t *)((const OFX_GLOBAL uint8_t *)base + lane_off_var(off));
}
__device__ __forceinline__ uint32_t gload_u32_unaligned_var(const uint8_t *base, uint32_t &off)
{
    return ((const OFX_GLOBAL UnalignedU32 *)((const OFX_GLOBAL uint8_t *)base + lane_off_var(off)))->v;
}
__device__ __forceinline__ uint32_t gload_u32(const uint8_t *base, uint32_t off)
{
    return *(const OFX_GLOBAL uint32_t *)((const OFX_GLOBAL uint8_t *)base + lane_off(off));
}
// streaming read: data this launch (and the following ones) will not touch again
__device__ __forceinline__ uint32_t gload_u32_nt(const uint8_t *base, uint32_t off)
{
    return __builtin_nontemporal_load((const OFX_GLOBAL uint32_t *)((const OFX_GLOBAL uint8_t *)base + lane_off(off)));
}
__device__ __forceinline__ uint32_t gload_u32_unaligned(const uint8_t *base, uint32_t off)
{
    return ((const OFX_GLOBAL UnalignedU32 *)((const OFX_GLOBAL uint8_t *)base + lane_off(off)))->v;
}
typedef OFX_GLOBAL float *gfloat_ptr;
__device__ __forceinline__ gfloat_ptr gptr_f32(float *base, uint32_t idx) { return (gfloat_ptr)base + lane_off(idx); }
// base + a BYTE offset kept by the caller (see lane_off_var): selects as scalar base + 32-bit lane offset
__device__ __forceinline__ gfloat_ptr gptr_f32_var(float *base, uint32_t &byte_off)
{
    return (gfloat_ptr)((OFX_GLOBAL uint8_t *)base + lane_off_var(byte_off));
}
// the flow is written once and never read back by this launch: streaming stores keep it from displacing the image rows
// the trailing window re-reads from L2
__device__ __forceinline__ void gstore_f32x2(gfloat_ptr p, float a, float b)
{
#if OFX_LK_NT_STORES
    __builtin_nontemporal_store(f32x2{a, b}, (OFX_GLOBAL f32x2 *)p);
#else
    *(OFX_GLOBAL f32x2 *)p = f32x2{a, b};
#endif
}
__device__ __forceinline__ void gstore_f32x4(gfloat_ptr p, f32x4 v)
{
    typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
#if OFX_LK_NT_STORES
    __builtin_nontemporal_store((f32x4_a8)v, (OFX_GLOBAL f32x4_a8 *)p);
#else
    *(OFX_GLOBAL f32x4_a8 *)p = (f32x4_a8)v;
#endif
}
__device__ __forceinline__ void gstore_i32(int32_t *p, int32_t v) { *(OFX_GLOBAL int32_t *)p = v; }
// stores at (wave-uniform base) + (per-lane byte offset)
__device__ __forceinline__ void gstore_u32(uint8_t *base, uint32_t off, uint32_t v)
{
    *(OFX_GLOBAL uint32_t *)((OFX_GLOBAL uint8_t *)base + lane_off(off)) = v;
}
__device__ __forceinline__ void gstore_u32x2(uint8_t *base, uint32_t off, uint32_t a, uint32_t b)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
    *(OFX_GLOBAL u32x2 *)((OFX_GLOBAL uint8_t *)base + lane_off(off)) = u32x2{a, b};
}
__device__ __forceinline__ void gstore_u16(uint8_t *base, uint32_t off, uint16_t v)
{
    *(OFX_GLOBAL uint16_t *)((OFX_GLOBAL uint8_t *)base + lane_off(off)) = v;
}
__device__ __forceinline__ void gstore_u8(uint8_t *base, uint32_t off, uint8_t v) { *((OFX_GLOBAL uint8_t *)base + lane_off(off)) = v; }

// a wave-uniform value, made opaque in an SGPR (no instruction is emitted)
template <typename T>
__device__ __forceinline__ void pin_scalar(T &x)
{
    asm volatile("" : "+s"(x));
}

// value of x held by lane (lane + D); 0 where that lane does not exist.  gfx9 DPP whole-wave shifts.
// One whole-wave shift by a single lane.  The empty asm makes the moved value opaque so that hipcc's DPP combiner
// cannot fold the move into its consumer: with ROCm 7.2 the folded form (v_subrev_u32_dpp) gave results shifted by
// one lane in the compat_cpu derivative stage (found by the parity test; tools/dbg.py shows the impulse response).
__device__ __forceinline__ int lane_shift_right(int x) // lane l receives lane l-1's value
{
    int r = __builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int lane_shift_left(int x) // lane l receives lane l+1's value
{
    int r = __builtin_amdgcn_update_dpp(0, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    asm volatile("" : "+v"(r));
    return r;
}

template <int D>
__device__ __forceinline__ int lane_from(int x)
{
    if constexpr (D == 0) {
        return x;
    } else if constexpr (D > 0) {
        return lane_from<D - 1>(lane_shift_left(x));
    } else {
        return lane_from<D + 1>(lane_shift_right(x));
    }
}

// acc + (the value of x held D lanes away), for the horizontal box.  The LAST one-lane shift is left unfenced and the sum
// is kept a two-operand add (the fence after it stops the compiler from merging two of them into a v_add3_u32, which
// cannot carry a DPP operand), so that the combiner folds move and add into ONE v_add_u32_dpp -- 2 instructions per
// neighbour term instead of 3 for two.  Addition is commutative, so the operand mix-up that broke the subtract form
// (above) cannot change the result.  OFX_LK_FOLD_DPP_ADDS=0 restores the fenced moves + add3.
#ifndef OFX_LK_FOLD_DPP_ADDS
#define OFX_LK_FOLD_DPP_ADDS 1
#endif
template <int D>
__device__ __forceinline__ int add_from(int acc, int x)
{
    if constexpr (D == 0) {
        return acc + x;
    } else {
#if OFX_LK_FOLD_DPP_ADDS
        const int near = lane_from<(D > 0 ? D - 1 : D + 1)>(x); // all but the last lane step: fenced moves
        int r = acc + __builtin_amdgcn_update_dpp(0, near, D > 0 ? 0x130 /* wave_shl:1 */ : 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        asm volatile("" : "+v"(r));
        return r;
#else
        return acc + lane_from<D>(x);
#endif
    }
}

// ---- horizontal box sum over columns [c-R, c+R] for the 4 columns of a lane -------------------------------
// q[k] = a0+..+ak, s[k] = ak+..+a3 (q[3] == s[0] == lane total).  Each helper adds its lanes' terms to acc.
template <int R, int I, int D>
__device__ __forceinline__ int hbox_right(const int (&q)[4], int acc)
{
    constexpr int hi = I + R; // last relative column of the window; lane +D holds relative columns 4D..4D+3
    if constexpr (hi < 4 * D) {
        return acc;
    } else if constexpr (hi >= 4 * D + 3) {
        return hbox_right<R, I, D + 1>(q, add_from<D>(acc, q[3]));
    } else {
        return add_from<D>(acc, q[hi - 4 * D]);
    }
}

template <int R, int I, int D>
__device__ __forceinline__ int hbox_left(const int (&s)[4], int acc)
{
    constexpr int lo = I - R; // first relative column; lane -D holds relative columns -4D..-4D+3
    if constexpr (lo > -4 * D + 3) {
        return acc;
    } else if constexpr (lo <= -4 * D) {
        return hbox_left<R, I, D + 1>(s, add_from<-D>(acc, s[0]));
    } else {
        return add_from<-D>(acc, s[lo + 4 * D]);
    }
}

template <int R, int I>
__device__ __forceinline__ int hbox_one(const int (&q)[4], const int (&s)[4])
{
    constexpr int lo = I - R, hi = I + R;
    constexpr int olo = lo > 0 ? lo : 0, ohi = hi < 3 ? hi : 3;
    int own;
    if constexpr (olo == 0) {
        own = q[ohi];
    } else if constexpr (ohi == 3) {
        own = s[olo];
    } else {
        own = q[ohi] - q[olo - 1];
    }
    return hbox_left<R, I, 1>(s, hbox_right<R, I, 1>(q, own));
}

// (column C's value) - y, C relative to this lane's first column: C in 0..3 is this lane's a[C], C < 0 lies in a lane to the
// left, C > 3 in one to the right.  All lane steps but the last are fenced moves (lane_from); the last one is the DPP
// operand of the subtraction, in the instruction's first source (dst = dpp(src0) - src1 is v_sub_u32_dpp, the form hipcc
// folds correctly; the "rev" form is the one it gets wrong, see lane_shift_right): a running value is kept negated every
// other step so that only that form occurs.
template <int C>
__device__ __forceinline__ int col_minus(const int (&a)[4], int y)
{
    if constexpr (C >= 0 && C <= 3) {
        return a[C] - y;
    } else {
        constexpr int idx = ((C % 4) + 4) % 4;
        constexpr int dist = (C - idx) / 4; // lanes away: > 0 to the right
        const int near = lane_from<(dist > 0 ? dist - 1 : dist + 1)>(a[idx]);
        int r = __builtin_amdgcn_update_dpp(0, near, dist > 0 ? 0x130 /* wave_shl:1 */ : 0x138 /* wave_shr:1 */, 0xf, 0xf, true) - y;
        asm volatile("" : "+v"(r));
        return r;
    }
}

template <int R>
__device__ __forceinline__ void hbox4(const int (&a)[4], int (&out)[4])
{
    if constexpr (OFX_LK_HBOX_SLIDE && R <= OFX_LK_HBOX_SLIDE_MAX_R) {
        // Windows of neighbouring columns differ by one column leaving and one entering (for R <= 4 both lie in this lane or the
        // one next to it, for R <= 8 up to two lanes away, else three): out[0] and out[3] in full, then out[1] = out[0] - a(-R) + a(R + 1) and
        // out[2] = out[3] - a(R + 3) + a(2 - R), each difference two instructions -- n = a(leaving) - out; out' = a(entering) - n
        // -- two short dependent chains, 11-12 instructions per quantity instead of 13.
        int q[4], s[4];
        q[0] = a[0];
        q[1] = q[0] + a[1];
        q[2] = q[1] + a[2];
        q[3] = q[2] + a[3];
        s[3] = a[3];
        s[2] = s[3] + a[2];
        s[1] = s[2] + a[1];
        s[0] = q[3];
        out[0] = hbox_one<R, 0>(q, s);
        out[3] = hbox_one<R, 3>(q, s);
        out[1] = col_minus<1 + R>(a, col_minus<0 - R>(a, out[0]));
        out[2] = col_minus<2 - R>(a, col_minus<3 + R>(a, out[3]));
    } else {
        int q[4], s[4];
        q[0] = a[0];
        q[1] = q[0] + a[1];
        q[2] = q[1] + a[2];
        q[3] = q[2] + a[3];
        s[3] = a[3];
        s[2] = s[3] + a[2];
        s[1] = s[2] + a[1];
        s[0] = q[3];
        out[0] = hbox_one<R, 0>(q, s);
        out[1] = hbox_one<R, 1>(q, s);
        out[2] = hbox_one<R, 2>(q, s);
        out[3] = hbox_one<R, 3>(q, s);
    }
}

// ---- the five box sums of a step in lockstep -----------------------------------------------------------------------------------
// hbox4 above handles one quantity: a chain of DPP operations, each fenced by an empty asm so that hipcc's DPP combiner cannot
// fold it wrongly (lane_shift_right).  Fenced, the chain's instructions stay in source order, every one reading the register the
// one before has just written -- and a DPP operand needs two wait states after the VALU write of its register: ~30 s_nop per
// row step (a tenth of the march's scalar instructions).  The five quantities of a step are independent, so hbox4x5 walks the
// same chain for all five at once, stage by stage: between two dependent instructions of one quantity lie the four of the others.
// Same operations, same operands, same results.
template <int K>
__device__ __forceinline__ void lane_from5(const int (&x)[5], int (&r)[5])
{
    if constexpr (K == 0) {
#pragma unroll
        for (int n = 0; n < 5; ++n) r[n] = x[n];
    } else {
        int t[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) t[n] = K > 0 ? lane_shift_left(x[n]) : lane_shift_right(x[n]);
        lane_from5<(K > 0 ? K - 1 : K + 1)>(t, r);
    }
}

template <int D>
__device__ __forceinline__ void add_from5(int (&acc)[5], const int (&x)[5])
{
    if constexpr (D == 0) {
#pragma unroll
        for (int n = 0; n < 5; ++n) acc[n] += x[n];
    } else {
        int near[5];
        lane_from5<(D > 0 ? D - 1 : D + 1)>(x, near); // all but the last lane step: fenced moves
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            int r = acc[n] + __builtin_amdgcn_update_dpp(0, near[n], D > 0 ? 0x130 /* wave_shl:1 */ : 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
            asm volatile("" : "+v"(r));
            acc[n] = r;
        }
    }
}

template <int C>
__device__ __forceinline__ void column5(const int (&a)[5][4], int (&x)[5])
{
#pragma unroll
    for (int n = 0; n < 5; ++n) x[n] = a[n][C];
}

template <int R, int I, int D>
__device__ __forceinline__ void hbox_right5(const int (&q)[5][4], int (&acc)[5])
{
    constexpr int hi = I + R;
    if constexpr (hi < 4 * D) {
        return;
    } else if constexpr (hi >= 4 * D + 3) {
        int x[5];
        column5<3>(q, x);
        add_from5<D>(acc, x);
        hbox_right5<R, I, D + 1>(q, acc);
    } else {
        int x[5];
        column5<hi - 4 * D>(q, x);
        add_from5<D>(acc, x);
    }
}

template <int R, int I, int D>
__device__ __forceinline__ void hbox_left5(const int (&s)[5][4], int (&acc)[5])
{
    constexpr int lo = I - R;
    if constexpr (lo > -4 * D + 3) {
        return;
    } else if constexpr (lo <= -4 * D) {
        int x[5];
        column5<0>(s, x);
        add_from5<-D>(acc, x);
        hbox_left5<R, I, D + 1>(s, acc);
    } else {
        int x[5];
        column5<lo + 4 * D>(s, x);
        add_from5<-D>(acc, x);
    }
}

template <int R, int I>
__device__ __forceinline__ void hbox_one5(const int (&q)[5][4], const int (&s)[5][4], int (&out)[5])
{
    constexpr int lo = I - R, hi = I + R;
    constexpr int olo = lo > 0 ? lo : 0, ohi = hi < 3 ? hi : 3;
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        if constexpr (olo == 0) out[n] = q[n][ohi];
        else if constexpr (ohi == 3) out[n] = s[n][olo];
        else out[n] = q[n][ohi] - q[n][olo - 1];
    }
    hbox_right5<R, I, 1>(q, out);
    hbox_left5<R, I, 1>(s, out);
}

// r = (column C's value) - y for the five quantities (col_minus, staged)
template <int C>
__device__ __forceinline__ void col_minus5(const int (&a)[5][4], const int (&y)[5], int (&r)[5])
{
    if constexpr (C >= 0 && C <= 3) {
#pragma unroll
        for (int n = 0; n < 5; ++n) r[n] = a[n][C] - y[n];
    } else {
        constexpr int idx = ((C % 4) + 4) % 4;
        constexpr int dist = (C - idx) / 4; // lanes away: > 0 to the right
        int x[5], near[5];
        column5<idx>(a, x);
        lane_from5<(dist > 0 ? dist - 1 : dist + 1)>(x, near);
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            int t = __builtin_amdgcn_update_dpp(0, near[n], dist > 0 ? 0x130 /* wave_shl:1 */ : 0x138 /* wave_shr:1 */, 0xf, 0xf, true) - y[n];
            asm volatile("" : "+v"(t));
            r[n] = t;
        }
    }
}

// a[n][j]: vertical sum of quantity n in this lane's column j; out[n][j]: its box sum over columns [c - R, c + R]
template <int R>
__device__ __forceinline__ void hbox4x5(const int (&a)[5][4], int (&out)[5][4])
{
    int q[5][4], s[5][4];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        q[n][0] = a[n][0];
        q[n][1] = q[n][0] + a[n][1];
        q[n][2] = q[n][1] + a[n][2];
        q[n][3] = q[n][2] + a[n][3];
        s[n][3] = a[n][3];
        s[n][2] = s[n][3] + a[n][2];
        s[n][1] = s[n][2] + a[n][1];
        s[n][0] = q[n][3];
    }
    int o0[5], o1[5], o2[5], o3[5];
    hbox_one5<R, 0>(q, s, o0);
    hbox_one5<R, 3>(q, s, o3);
    if constexpr (OFX_LK_HBOX_SLIDE && R <= OFX_LK_HBOX_SLIDE_MAX_R) {
        int t[5];
        col_minus5<0 - R>(a, o0, t);
        col_minus5<1 + R>(a, t, o1);
        col_minus5<3 + R>(a, o3, t);
        col_minus5<2 - R>(a, t, o2);
    } else {
        hbox_one5<R, 1>(q, s, o1);
        hbox_one5<R, 2>(q, s, o2);
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) out[n][0] = o0[n], out[n][1] = o1[n], out[n][2] = o2[n], out[n][3] = o3[n];
}

// geometry of a wave tile for radius R (also used by the host)
template <int R>
struct TileGeom {
    static constexpr int LO_LANE = (R + 1 + 3) / 4; // first lane whose 4 outputs have all their taps inside the wave
    static constexpr int HI_LANE = (251 - R) / 4;   // last such lane (derivatives are valid for wave columns 1..254)
    static constexpr int OUT_W = (HI_LANE - LO_LANE + 1) * 4;
};

// ---- packed (in | out) form -------------------------------------------------------------------------------------------
// The march keeps TWO derivative rows in flight: the one entering the vertical window and the one leaving it.  They go
// through exactly the same arithmetic, on values that fit 16 bits (|Ix|,|Iy| <= 1020, |It| <= 4335 before the centre tap
// is removed), so both are carried in ONE register per quantity: low half = entering row, high half = leaving row.
// v_pk_*_i16 then does both rows per instruction, one DPP move carries both neighbours, and
//     V += Ix_in*Iy_in - Ix_out*Iy_out      is ONE   v_dot2_i32_i16( (Ix_in,Ix_out), (Iy_in,-Iy_out), V ).
// It also halves the row state in registers.
typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 lane_shift_s2(s2 v, bool from_left)
{
    return as_s2((uint32_t)(from_left ? lane_shift_right((int)as_u32(v)) : lane_shift_left((int)as_u32(v))));
}

__device__ __forceinline__ s2 pk_two()
{
    uint32_t v = 0x00020002u; // opaque (2,2): keeps the [1 2 1] combinations as one v_pk_mad_u16 instead of shift + add
    asm volatile("" : "+s"(v));
    return as_s2(v);
}

template <int MODE>
struct RowPk; // one image row of this lane's 4 columns for both marching windows (lo: entering, hi: leaving)
template <>
struct RowPk<OFX_MODE_LK_FLOAT> {
    s2 p[4]; // prev
    s2 d[4]; // next - prev
};
template <>
struct RowPk<OFX_MODE_COMPAT_CPU> {
    s2 p[4]; // prev
    s2 n[4]; // next
};

// makes the row's registers opaque at this point of the program (no instruction is emitted)
__device__ __forceinline__ void pin_row(RowPk<OFX_MODE_LK_FLOAT> &r)
{
    asm volatile("" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]), "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]));
}
__device__ __forceinline__ void pin_row(RowPk<OFX_MODE_COMPAT_CPU> &r)
{
    asm volatile("" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]), "+v"(r.n[0]), "+v"(r.n[1]), "+v"(r.n[2]), "+v"(r.n[3]));
}

// byte j of `in_raw` -> low half, byte j of `out_raw` -> high half (zero-extended): one v_perm_b32 per column
template <int J>
__device__ __forceinline__ s2 pair_bytes(uint32_t in_raw, uint32_t out_raw)
{
    constexpr uint32_t sel = (uint32_t)J | (0x0cu << 8) | ((uint32_t)(4 + J) << 16) | (0x0cu << 24);
    return as_s2(__builtin_amdgcn_perm(out_raw, in_raw, sel));
}

__device__ __forceinline__ void unpack_pk(uint32_t pi, uint32_t ni, uint32_t po, uint32_t no, RowPk<OFX_MODE_LK_FLOAT> &r)
{
    r.p[0] = pair_bytes<0>(pi, po);
    r.p[1] = pair_bytes<1>(pi, po);
    r.p[2] = pair_bytes<2>(pi, po);
    r.p[3] = pair_bytes<3>(pi, po);
    r.d[0] = pair_bytes<0>(ni, no) - r.p[0];
    r.d[1] = pair_bytes<1>(ni, no) - r.p[1];
    r.d[2] = pair_bytes<2>(ni, no) - r.p[2];
    r.d[3] = pair_bytes<3>(ni, no) - r.p[3];
}

__device__ __forceinline__ void unpack_pk(uint32_t pi, uint32_t ni, uint32_t po, uint32_t no, RowPk<OFX_MODE_COMPAT_CPU> &r)
{
    r.p[0] = pair_bytes<0>(pi, po);
    r.p[1] = pair_bytes<1>(pi, po);
    r.p[2] = pair_bytes<2>(pi, po);
    r.p[3] = pair_bytes<3>(pi, po);
    r.n[0] = pair_bytes<0>(ni, no);
    r.n[1] = pair_bytes<1>(ni, no);
    r.n[2] = pair_bytes<2>(ni, no);
    r.n[3] = pair_bytes<3>(ni, no);
}

// derivatives of the middle rows of both windows, UNMASKED: pixels outside the image or the strip are removed by
// accumulate_pk's multipliers.  `two` = pk_two(), made once per wave.
__device__ __forceinline__ void derivs_pk(const RowPk<OFX_MODE_LK_FLOAT> &t, const RowPk<OFX_MODE_LK_FLOAT> &m,
                                          const RowPk<OFX_MODE_LK_FLOAT> &b, const s2 two, s2 (&ix)[4], s2 (&iy)[4], s2 (&it)[4])
{
    s2 sm[6], df[6], g[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sm[j + 1] = m.p[j] * two + (t.p[j] + b.p[j]); // [1 2 1]^T (kernels.cpp:6-19)
        df[j + 1] = b.p[j] - t.p[j];                  // [-1 0 1]^T
        g[j + 1] = m.d[j] * two + (t.d[j] + b.d[j]);  // Dt_3x3 = [1 2 1]^T[1 2 1] - centre (kernels.cpp:20-24) on next - prev
    }
    sm[0] = lane_shift_s2(sm[4], true);
    sm[5] = lane_shift_s2(sm[1], false);
    df[0] = lane_shift_s2(df[4], true);
    df[5] = lane_shift_s2(df[1], false);
    g[0] = lane_shift_s2(g[4], true);
    g[5] = lane_shift_s2(g[1], false);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ix[j] = sm[j + 2] - sm[j];
        iy[j] = df[j + 1] * two + (df[j] + df[j + 2]);
        it[j] = g[j + 1] * two + (g[j] + g[j + 2]) - m.d[j];
    }
}

__device__ __forceinline__ void derivs_pk(const RowPk<OFX_MODE_COMPAT_CPU> &t, const RowPk<OFX_MODE_COMPAT_CPU> &m,
                                          const RowPk<OFX_MODE_COMPAT_CPU> &b, const s2, s2 (&ix)[4], s2 (&iy)[4], s2 (&it)[4])
{
    // cpu path: int accumulator truncated after every tap (OptFlowCPU.cpp:102) => each Gaussian tap contributes
    // floor(px * w): corner px>>4, edge px>>3, centre px>>2 (GAUS_KERNEL_3x3, kernels.cpp:61-64); u8 wrap (:106, :15)
    s2 sm[6], df[6], sp[6], sn[6], mp[4], mn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sm[j + 1] = (m.p[j] + m.p[j]) + t.p[j] + b.p[j];
        df[j + 1] = b.p[j] - t.p[j];
        sp[j + 1] = (t.p[j] >> 4) + (m.p[j] >> 3) + (b.p[j] >> 4); // column contribution left/right of the centre
        sn[j + 1] = (t.n[j] >> 4) + (m.n[j] >> 3) + (b.n[j] >> 4);
        mp[j] = (t.p[j] >> 3) + (m.p[j] >> 2) + (b.p[j] >> 3);     // ... as the centre column
        mn[j] = (t.n[j] >> 3) + (m.n[j] >> 2) + (b.n[j] >> 3);
    }
    sm[0] = lane_shift_s2(sm[4], true);
    sm[5] = lane_shift_s2(sm[1], false);
    df[0] = lane_shift_s2(df[4], true);
    df[5] = lane_shift_s2(df[1], false);
    sp[0] = lane_shift_s2(sp[4], true);
    sp[5] = lane_shift_s2(sp[1], false);
    sn[0] = lane_shift_s2(sn[4], true);
    sn[5] = lane_shift_s2(sn[1], false);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        constexpr uint32_t m8 = 0x00ff00ffu; // the (unsigned char) wrap of OptFlowCPU.cpp:106
        ix[j] = as_s2(as_u32(sm[j + 2] - sm[j]) & m8);
        iy[j] = as_s2(as_u32((df[j + 1] + df[j + 1]) + df[j] + df[j + 2]) & m8);
        const s2 gp = sp[j] + mp[j] + sp[j + 2], gn = sn[j] + mn[j] + sn[j + 2];
        it[j] = as_s2(as_u32(gn - gp) & 0x00ff00ffu);
    }
}

// V += P(entering) - P(leaving) for the five products, order of the planes: OptFlowCPU.cpp:347-358.
// mm[j] = per-column multiplier pair: low half 1 when the entering row's pixel counts (column inside the image, row inside
// the image), high half -1 when the leaving row's pixel counts (... and that row has entered this strip), else 0.  One
// v_pk_mul_lo_u16 by it both removes the pixels that do not count and negates the leaving row's factor; every product
// below has nx or ny as a factor, so Ix, Iy and It themselves need no masking (rows and columns outside the image read as
// zero bytes, so every value stays inside its 16-bit range).
__device__ __forceinline__ void accumulate_pk(const s2 (&ix)[4], const s2 (&iy)[4], const s2 (&it)[4], const uint32_t (&mm)[4],
                                              int (&vxx)[4], int (&vyy)[4], int (&vxy)[4], int (&vxt)[4], int (&vyt)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const s2 nx = ix[j] * as_s2(mm[j]), ny = iy[j] * as_s2(mm[j]);
        vxx[j] = __builtin_amdgcn_sdot2(ix[j], nx, vxx[j], false);
        vyy[j] = __builtin_amdgcn_sdot2(iy[j], ny, vyy[j], false);
        vxy[j] = __builtin_amdgcn_sdot2(ix[j], ny, vxy[j], false);
        vxt[j] = __builtin_amdgcn_sdot2(nx, it[j], vxt[j], false);
        vyt[j] = __builtin_amdgcn_sdot2(ny, it[j], vyt[j], false);
    }
}

// ---- the flow stores ------------------------------------------------------------------------------------------------------
// A lane ends a step with the (u,v) pairs of its 4 pixels: 32 contiguous bytes, 2 KB per wave and row.  Written as they
// lie -- two dwordx4 per lane -- each store instruction covers every other 16 bytes of that range, and streaming
// (non-temporal) stores, which the kernel needs so that the flow does not evict the image rows from L2, are then written
// to memory as half-filled sectors: tools/ubench/store_patterns.hip measures 2.5 TB/s for that shape against 5.3-5.6 TB/s
// when every instruction covers whole 128-byte lines (runs of >= 8 lanes x 16 B), and 354 MB of flow per four 4K pairs at
// 2.6 TB/s is the 135 us the launch took however little arithmetic was left in it (profiles/r02_*).  So the wave first
// exchanges the row through LDS -- each lane writes its 32 bytes, then reads the two 16-byte chunks l and l + 64 of the
// row (counted from the first valid byte) -- and both store instructions cover 1 KB without gaps, starting on the tile's
// first output byte, which is 128-byte aligned whenever the level's row pitch is (every BASELINE level).  LDS instructions
// issue beside the VALU, a wave's LDS operations execute in order, and the region is private to the wave: no barrier.
constexpr int kLkWaveLds = 2048 + 128; // one row of a wave + the reads of the lanes past its valid end
typedef __attribute__((address_space(3))) uint8_t *lds_ptr;

// One wave of the fused level kernel: `wave` indexes the (level, tile, strip) work items of the table, `lane` is 0..63.
// MAY_ACC: the launch may contain accumulating items (refinement iterations); false compiles that path out (the stream
// kernel never has any, and the extra live registers would push it over its 96-VGPR budget)
// FAST: the <= 1 ulp solve (lk_solve.h) instead of the replay of the reference's operation order
// xlds: kLkWaveLds bytes of LDS private to this wave (the exchange in front of the flow stores, below)
// INTERIOR: all 256 columns of the wave's tile lie inside the image (every tile of a row but its first and its last; the
// dispatcher lk_wave below decides): the column masks are compile-time constants then -- no byte mask on the loaded rows, one
// scalar multiplier pair for all four columns instead of four per-lane selects per step.
template <int R, int MODE, bool SUMS, bool MAY_ACC, bool FAST, bool INTERIOR>
__device__ __forceinline__ void lk_wave_impl(const LkTable &T, int wave, int lane, uint8_t *xlds)
{
    using G = TileGeom<R>;
    constexpr int NS = 2 * R + 1;

    if (wave >= T.first_block[T.n]) return;
    // the item this wave belongs to: the last one whose first_block <= wave (up to 40 items: bisection, 6 scalar loads)
    int level = 0, hi = T.n;
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (wave >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    // The level's arguments, held in SGPRs for the whole march: left as references into the kernarg table the compiler
    // re-reads them from memory inside the loop (~25 s_load + wait per step).
    LkArgs A = T.lv[level];
    pin_scalar(A.prev);
    pin_scalar(A.next);
    pin_scalar(A.flow);
    pin_scalar(A.w);
    pin_scalar(A.h);
    pin_scalar(A.pitch);
    pin_scalar(A.row0);
    pin_scalar(A.row_end);
    pin_scalar(A.flow_row0);
    pin_scalar(A.accumulate);
    pin_scalar(A.min_det);
    const SolveOpts sopt{A.min_det};
    if constexpr (SUMS) {
        pin_scalar(A.sums);
        pin_scalar(A.sums_plane);
    }
    const int block = wave - T.first_block[level];
    const int tile = block % A.tiles_x;
    const int strip = block / A.tiles_x;
    const int cb = tile * G::OUT_W - G::LO_LANE * 4 + 4 * lane; // first of this lane's 4 image columns
    const int ys = A.out_y0 + strip * A.strip_h;
    const int ye = min(ys + A.strip_h, A.out_y1);

    // column validity: bytes outside [0,w) read as zero, derivatives there are zero
    const bool ld_ok = INTERIOR || (cb >= 0 && cb < A.w);
    uint32_t bmask = INTERIOR ? 0xffffffffu : 0u;
    int cm[4] = {-1, -1, -1, -1};
    if constexpr (!INTERIOR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = (cb + j) >= 0 && (cb + j) < A.w;
            cm[j] = in ? -1 : 0;
            bmask |= in ? (0xffu << (8 * j)) : 0u;
        }
    }
    uint32_t col_off = ld_ok ? (uint32_t)cb : 0u; // 32-bit lane offset on top of a wave-uniform row pointer
    uint32_t flow_off = 8u * (uint32_t)(cb > 0 ? cb : 0); // byte offset of this lane's first (u,v) pair in a flow row
    // the exchanged layout: chunk c of the row (16 bytes = pixels x0 + 2c, x0 + 2c + 1, x0 = the tile's first output column)
    // is written by lane c (first store) or c - 64 (second store).  Everything per lane derives from 16 * lane, so that the
    // exchange costs two VGPRs: chunk c holds min(max(nv - 2c, 0), 2) valid pixels, nv = the tile's output columns.
    const int x0 = tile * G::OUT_W;
    const int nv = min(x0 + G::OUT_W, A.w) - x0;
    uint32_t l16 = 16u * (uint32_t)lane;
    const int lim = 8 * nv - 16, c16 = 16 * lane;
    const bool st_lo4 = c16 <= lim, st_lo2 = c16 == lim + 8, st_hi4 = c16 <= lim - 1024, st_hi2 = c16 == lim + 8 - 1024;
    // LDS: lane l's 32 bytes go to offset 32 l; the tile's first output byte is lane LO_LANE's, so chunk c sits at 32 LO + 16 c
    const lds_ptr xl_w = (lds_ptr)xlds + 32 * lane;
    const lds_ptr xl_base = (lds_ptr)xlds + 32 * G::LO_LANE;

    // rows outside the image are the zero border; rows past the last one this strip needs (the loop prefetches one
    // row ahead) or outside the buffer are never dereferenced.  The row test is wave-uniform (scalar branch); lanes
    // whose columns lie outside the image read column 0 of the row and are zeroed by bmask.
    const int y_lim = min(min(ye + R + 1, A.h), A.row_end);
    const int y_min = max(0, A.row0);
    // Loads are split into fetch (issue the memory reads, touch nothing) and finish (mask / permute): the march fetches
    // one step ahead and finishes at the top of the next step, so that no wave waits on a load it has just issued.
    auto fetch_row = [&](const uint8_t *img, int y) -> uint32_t {
        uint32_t v = 0u;
        if (y >= y_min && y < y_lim) {
            const uint8_t *row = img + (size_t)(uint32_t)(y - A.row0) * (size_t)(uint32_t)A.pitch;
            pin_scalar(row); // scalar base + 32-bit lane offset: no VALU address arithmetic
            v = gload_u32_var(row, col_off);
        }
        return v;
    };
    auto finish_row = [&](uint32_t raw) -> uint32_t {
        if constexpr (INTERIOR) return raw;
        else return raw & bmask;
    };

    // ---- fused shift ---------------------------------------------------------------------------------------------------
    // Below the top level the reference replaces next by cpu::shift_back_pyramid(next) (OptFlowCPU.cpp:241-282): pixel
    // (x,y) takes next((int)(x+u), (int)(y+v)) when that lands inside the image, else the byte the leading memcpy of
    // w*h bytes left there (its own value if 3*(y*w+x) < w*h, else 0).  The column map (int)((float)x + u) does not
    // depend on the row and is monotone with steps of 0 or 1, so the in-image targets of a lane's 4 columns always lie in
    // 4 consecutive source bytes.  Per lane, once: the base column `nb` of those bytes and a v_perm_b32 selector that
    // picks, for each of its 4 pixels, a byte of the shifted dword, the pixel's own byte (target outside the image), or
    // zero (column outside the image).  Per row: one unaligned dword at (row (int)(y+v), column nb), one aligned dword of
    // the row itself, one v_perm -- for every lane alike, so that image-edge tiles cost the same as interior tiles (all
    // waves of the launch run for its whole duration: one slow tile would set the kernel's time).
    // No shift (top level) is the same code with (u,v) = (0,0): every target is the pixel itself.
    const float su = A.uv ? A.uv[0] : 0.0f, sv = A.uv ? A.uv[1] : 0.0f;
    uint32_t nb_off = 0u, sel = 0u, own_sel = 0u; // own_sel: every in-image pixel takes its own byte (row target outside the image)
    bool all_in; // wave-uniform: no pixel of this wave has its column target outside the image (the interior-tile case)
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
        sel = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = cb + j;
            // selector byte: 0..3 = byte of the shifted dword (src1), 4..7 = byte of the own dword (src0), 0x0c = 0x00
            const uint32_t sj = !(x >= 0 && x < A.w) ? 0x0cu : (in[j] ? (uint32_t)(n[j] - nb) : (uint32_t)(4 + j));
            sel |= sj << (8 * j);
            own_sel |= (!(x >= 0 && x < A.w) ? 0x0cu : (uint32_t)(4 + j)) << (8 * j);
            if (x >= 0 && x < A.w && !in[j]) lane_all_in = false;
        }
        nb_off = (uint32_t)nb;
        all_in = __all(lane_all_in) != 0;
    }
    // a pixel whose target is outside the image keeps its own byte iff 3*(y*w+x) < w*h, i.e. iff 3*x < w*(h-3*y):
    // all of row y for y < h/3, none for y >= ceil(h/3), and x < w*(h%3)/3 in the one row in between (if h%3 != 0)
    const int y_none = (A.h + 2) / 3, y_part = (A.h % 3) ? A.h / 3 : -1;
    struct NextRaw {
        uint32_t own, sh; // the row's own dword / the dword at the shifted position (0 where not needed)
        uint32_t have;    // wave-uniform: ~0 when the shifted row exists
    };
    // The row map y -> (int)((float)y + v) is wave-uniform but float arithmetic, i.e. VALU work (8 instructions per row,
    // two rows per step).  It is evaluated for 64 consecutive rows at a time instead -- lane i holds the target of row
    // map_base + i, or -1 when that target is outside the image or the buffer -- and a step reads its two entries with
    // v_readlane_b32.  The march refreshes the table when the entering row runs off its end (refresh_map).
    int map_base = 0, row_map = -1;
    auto refresh_map = [&](int y0) {
        map_base = y0;
        const float ty = (float)(y0 + lane) + sv;
        const bool yin = ty > -1.0f && ty < (float)A.h;
        const int ny = yin ? (int)ty : 0;
        row_map = (yin && ny >= A.row0 && ny < A.row_end) ? ny : -1;
    };
    auto fetch_next = [&](int y) -> NextRaw {
        NextRaw r = {0u, 0u, 0u};
        if (y >= y_min && y < y_lim) {
            const int ny = __builtin_amdgcn_readlane(row_map, y - map_base);
            const bool have = ny >= 0;
            r.have = have ? ~0u : 0u;
            if (have) {
                const uint8_t *srow = A.next + (size_t)(uint32_t)(ny - A.row0) * (size_t)(uint32_t)A.pitch;
                pin_scalar(srow);
                r.sh = gload_u32_unaligned_var(srow, nb_off);
            }
            // interior tile with its target row inside the image: every byte comes from the shifted dword
            if (!(have && all_in) && y < y_none) {
                const uint8_t *orow = A.next + (size_t)(uint32_t)(y - A.row0) * (size_t)(uint32_t)A.pitch;
                pin_scalar(orow);
                r.own = gload_u32_var(orow, col_off);
                if (y == y_part) {
                    uint32_t km = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (3ll * (cb + j) < (long long)A.w * (A.h % 3)) km |= 0xffu << (8 * j);
                    r.own &= km;
                }
            }
        }
        return r;
    };
    // both selectors give 0 for columns outside the image, and own == sh == 0 for rows outside it
    auto finish_next = [&](const NextRaw &r) -> uint32_t {
        return __builtin_amdgcn_perm(r.own, r.sh, (sel & r.have) | (own_sel & ~r.have));
    };
    auto load_row = [&](const uint8_t *img, int y) -> uint32_t { return finish_row(fetch_row(img, y)); };
    auto load_next = [&](int y) -> uint32_t { return finish_next(fetch_next(y)); };

    // Two 3-row windows march down the strip NS rows apart: `in` around the derivative row entering the vertical
    // window, `out` around the row leaving it.  The leaving row's derivatives are recomputed from the image (its rows
    // are L2-resident: this wave read them NS steps ago) instead of being kept in an LDS ring: that costs one more
    // derivative stage per step but no LDS, no pack/unpack of 16-bit fields, and it lets occupancy follow VGPRs only.
    // Both windows share their registers (low halves: entering, high halves: leaving) and live in three slots, the rows
    // (t, m, b) of step s in slots (s, s + 1, s + 2) mod 3: the loop is unrolled three times and every slot index is a
    // compile-time constant (no register-to-register rotation).
    //
    // Priming, folded: the first output row ys needs the 2R + 1 derivative rows ys - R .. ys + R in the running sums and
    // nothing leaves before output row ys + 1, so during the first steps the high halves have no leaving row to carry.  They
    // carry ENTERING rows instead, with multiplier +1: the H = R - 1 rows ys - R .. ys - 2 ride in the high halves of steps
    // 0 .. H - 1 while the low halves start at row ys - 1.  The high stream then jumps back to become the leaving window:
    // its fetches of steps R - 1, R, R + 1 are rows ys - R - 1, ys - R, ys - R + 1 -- which is what `yo + 2` gives there, so
    // the restart needs no code of its own -- its multiplier is 0 in steps H .. R + 1, and from step R + 2 on it is the row
    // leaving.  The first row comes out at step PR = R + 1 instead of 2R: R - 1 steps less per strip (3 of 8 for the 9x9
    // window, 8 of 18 for 19x19), which is 17 % -> 11 % of the steps of a 4K pair launched alone and what bounds the
    // rows a rank of a sharded pair recomputes.  OFX_LK_FOLD_PRIMING=0 (H = 0) is the plain 2R-step priming.
    constexpr int H = OFX_LK_FOLD_PRIMING ? R - 1 : 0;
    constexpr int PR = 2 * R - H;    // steps before the first output row
    const int y_first = ys - R;      // first derivative row this strip needs
    const int y_lo0 = y_first + H;   // derivative row of the low halves at step 0
    const int nsteps = (ye - ys) + PR;
    RowPk<MODE> wp[3];
    const s2 two = pk_two();
    // rows of the leaving window before y_first - 1 are never used: fed as zeros
    auto fetch_out_prev = [&](int r) -> uint32_t { return r < y_first - 1 ? 0u : fetch_row(A.prev, r); };
    auto fetch_out_next = [&](int r) -> NextRaw { return r < y_first - 1 ? NextRaw{0u, 0u, 0u} : fetch_next(r); };
    refresh_map(y_first - 1);
    if constexpr (H > 0) {
        unpack_pk(load_row(A.prev, y_lo0 - 1), load_next(y_lo0 - 1), load_row(A.prev, y_first - 1), load_next(y_first - 1), wp[0]);
        unpack_pk(load_row(A.prev, y_lo0), load_next(y_lo0), load_row(A.prev, y_first), load_next(y_first), wp[1]);
        unpack_pk(load_row(A.prev, y_lo0 + 1), load_next(y_lo0 + 1), load_row(A.prev, y_first + 1), load_next(y_first + 1), wp[2]);
    } else {
        unpack_pk(load_row(A.prev, y_lo0 - 1), load_next(y_lo0 - 1), 0u, 0u, wp[0]);
        unpack_pk(load_row(A.prev, y_lo0), load_next(y_lo0), 0u, 0u, wp[1]);
        unpack_pk(load_row(A.prev, y_lo0 + 1), load_next(y_lo0 + 1), 0u, 0u, wp[2]);
    }

    int vxx[4] = {0, 0, 0, 0}, vyy[4] = {0, 0, 0, 0}, vxy[4] = {0, 0, 0, 0}, vxt[4] = {0, 0, 0, 0}, vyt[4] = {0, 0, 0, 0};
#ifdef OFX_X_SKELETON // timing experiment: row loads, the wait, the LDS exchange and the streaming stores -- nothing else
    uint32_t skel = 0u;
#endif

    auto body = [&](auto K, int s) {
        constexpr int k = decltype(K)::value; // s mod 3
        const int yy = y_lo0 + s;             // derivative row entering the window (low halves)
        const int yo = yy - NS;               // derivative row leaving it (high halves, once the folded priming is over)
        const bool folded = H > 0 && s < H;   // high halves: the entering row y_first + s
        const int yh = folded ? y_first + s : yo;

#if defined(OFX_X_EXTRA_SALU) || defined(OFX_X_EXTRA_VALU) // sensitivity experiments: N more scalar / vector instructions per row step
        {
#ifdef OFX_X_EXTRA_SALU
            int sx = s;
#pragma unroll
            for (int e = 0; e < OFX_X_EXTRA_SALU; ++e) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
#endif
#ifdef OFX_X_EXTRA_VALU
            int vx = lane;
#pragma unroll
            for (int e = 0; e < OFX_X_EXTRA_VALU; ++e) asm volatile("v_add_u32 %0, %0, %0" : "+v"(vx));
#endif
        }
#endif
        // Issue the loads of the rows the next step adds (yy + 2 and the high stream's).  They are finished (mask / permute /
        // unpack) at the end of this step, before its flow stores: gfx9 counts loads and stores in one vmcnt and only orders
        // returns within a type, so a wait for a load that has younger stores outstanding is a wait for those stores too.
        if (yy + 2 - map_base >= 64) refresh_map(yo + 2); // (yo + 2 is the lowest row still to be looked up)
        const int ro = (H > 1 && s + 1 < H) ? y_first + s + 2 : yo + 2; // b row of the high stream's next step
        const uint32_t pf_ip = fetch_row(A.prev, yy + 2), pf_op = fetch_out_prev(ro);
        const NextRaw pf_in = fetch_next(yy + 2), pf_on = fetch_out_next(ro);
        // a refinement launch adds to the flow already there: its 8 floats are fetched with the rows and waited for once
        const bool emit = s >= PR;
        const int y = yy - R;
        const bool out_lane = lane >= G::LO_LANE && lane <= G::HI_LANE && cb < A.w;
        const size_t rowpix = (size_t)(y - A.flow_row0) * (size_t)A.w; // scalar
        float old_uv[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (!SUMS && MAY_ACC) {
            if (A.accumulate && emit && out_lane) {
                float *frow = A.flow + 2 * rowpix;
                pin_scalar(frow);
                const gfloat_ptr src = gptr_f32_var(frow, flow_off);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cb + j < A.w) old_uv[2 * j] = src[2 * j], old_uv[2 * j + 1] = src[2 * j + 1];
            }
        }

        // rows outside the image have no derivatives (their window taps are skipped, OptFlowCPU.cpp:182); the high halves
        // count +1 while they carry an entering row, -1 once they carry the leaving one (yo >= y_first), 0 in between
        const uint32_t him = folded ? 0x00010000u : (yo >= y_first ? 0xffff0000u : 0u);
        uint32_t rowm = ((yy >= 0 && yy < A.h) ? 0x00000001u : 0u) | ((yh >= 0 && yh < A.h) ? him : 0u);
        if constexpr (INTERIOR) rowm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowm); // one scalar multiplier pair for all columns
        const uint32_t mm[4] = {(uint32_t)cm[0] & rowm, (uint32_t)cm[1] & rowm, (uint32_t)cm[2] & rowm, (uint32_t)cm[3] & rowm};
#ifndef OFX_X_SKELETON
        s2 ix[4], iy[4], it[4];
        derivs_pk(wp[k], wp[(k + 1) % 3], wp[(k + 2) % 3], two, ix, iy, it);
        accumulate_pk(ix, iy, it, mm, vxx, vyy, vxy, vxt, vyt);
#else
        (void)mm;
#endif
        // row yy - 1 is done with: its slot takes row yy + 2 once the step's arithmetic is over
        // (the barrier keeps the scheduler from hoisting these few ALU ops -- and with them the wait -- up to the loads;
        // pin_row keeps the sink passes from moving them down into the next step, below its loads)
        auto take_rows = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#ifdef OFX_X_SKELETON
            skel ^= finish_row(pf_ip) ^ finish_next(pf_in) ^ finish_row(pf_op) ^ finish_next(pf_on);
            asm volatile("" : "+v"(skel));
#else
            unpack_pk(finish_row(pf_ip), finish_next(pf_in), finish_row(pf_op), finish_next(pf_on), wp[k]);
            pin_row(wp[k]);
#endif
        };

        // ---- emit output row y = yy - R ------------------------------------------------------------------------
        // (one call site for take_rows, after the arithmetic and before the stores)
        int hxx[4], hyy[4], hxy[4], hxt[4], hyt[4];
        float uv[8];
        f32x4 xlo, xhi; // this lane's two chunks of the exchanged row (only defined, and only used, in emitting steps: the
        asm("" : "=v"(xlo), "=v"(xhi)); // empty asm stands in for an initialisation that would cost 8 v_mov per step)
        if (emit) {
#ifdef OFX_X_SKELETON
#pragma unroll
            for (int j = 0; j < 4; ++j) hxx[j] = hyy[j] = hxy[j] = hxt[j] = hyt[j] = (int)skel + j;
#elif defined(OFX_X_NOHBOX) // timing experiments (OFX_BUILD_DEFS): what a stage costs is what the launch gains without it
#pragma unroll
            for (int j = 0; j < 4; ++j) hxx[j] = vxx[j], hyy[j] = vyy[j], hxy[j] = vxy[j], hxt[j] = vxt[j], hyt[j] = vyt[j];
#else
            hbox4<R>(vxx, hxx);
            hbox4<R>(vyy, hyy);
            hbox4<R>(vxy, hxy);
            hbox4<R>(vxt, hxt);
            hbox4<R>(vyt, hyt);
#endif
            if constexpr (!SUMS) {
                // every lane solves (the halo lanes' results are dropped): no divergence before the rows are taken
#if defined(OFX_X_NOSOLVE) || defined(OFX_X_SKELETON)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uv[2 * j] = __int_as_float(hxx[j] ^ hxy[j] ^ hxt[j]);
                    uv[2 * j + 1] = __int_as_float(hyy[j] ^ hyt[j]);
                }
#else
                solve_lane<MODE, FAST>(hxx, hyy, hxy, hxt, hyt, sopt, uv);
#endif
                if constexpr (MAY_ACC) {
                    if (A.accumulate) { // (old_uv is zero in the lanes that do not store)
#pragma unroll
                        for (int j = 0; j < 8; ++j) uv[j] = old_uv[j] + uv[j];
                    }
                }
                // exchange the row through LDS (see "the flow stores" above): in as it lies, out as two gap-free runs
                *(__attribute__((address_space(3))) f32x4 *)(xl_w) = f32x4{uv[0], uv[1], uv[2], uv[3]};
                *(__attribute__((address_space(3))) f32x4 *)(xl_w + 16) = f32x4{uv[4], uv[5], uv[6], uv[7]};
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const lds_ptr xl_r = xl_base + lane_off_var(l16);
                xlo = *(__attribute__((address_space(3))) f32x4 *)(xl_r);
                xhi = *(__attribute__((address_space(3))) f32x4 *)(xl_r + 1024);
            }
        }
        take_rows();
        if (emit) {
            if constexpr (SUMS) {
                if (out_lane) {
                    const size_t pix = rowpix + (uint32_t)cb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (cb + j < A.w) {
                            gstore_i32(A.sums + pix + j, hxx[j]);
                            gstore_i32(A.sums + A.sums_plane + pix + j, hyy[j]);
                            gstore_i32(A.sums + 2 * A.sums_plane + pix + j, hxy[j]);
                            gstore_i32(A.sums + 3 * A.sums_plane + pix + j, hxt[j]);
                            gstore_i32(A.sums + 4 * A.sums_plane + pix + j, hyt[j]);
                        }
                    }
                }
            } else {
                float *frow = A.flow + 2 * (rowpix + (size_t)x0); // the tile's first output pixel of this row
                pin_scalar(frow);
                // chunk c: both pixels valid <=> 16 c <= 8 nv - 16; only the first (a level of odd width ends inside it) <=>
                // 16 c == 8 nv - 8.  (The address is formed inside each branch: instruction selection only picks the
                // scalar-base form when the offset's extension sits in the block of the access; it is only 8-byte aligned
                // in general -- odd w*y.)
                // The four predicates are loop-invariant lane masks: computed once, applied as exec masks.
                if (st_lo4) {
                    gstore_f32x4(gptr_f32_var(frow, l16), xlo);
                } else if (st_lo2) {
                    gstore_f32x2(gptr_f32_var(frow, l16), xlo.x, xlo.y);
                }
                if (st_hi4) {
                    gstore_f32x4(gptr_f32_var(frow, l16) + 256, xhi);
                } else if (st_hi2) {
                    gstore_f32x2(gptr_f32_var(frow, l16) + 256, xhi.x, xhi.y);
                }
            }
        }
    };

    // The waves of a SIMD are served oldest first, not in turn: left alone, the four LK waves of a SIMD finish one after the
    // other (70 / 95 / 125 / 165 us of a 186 us launch, tools/stream_timeline.py) and the last one runs its final quarter
    // alone, at the ~60 % issue rate of a lone wave.  Each wave therefore lowers its own priority as it advances through its
    // strip (3 -> 0 at the quarter marks): whoever is behind is served first, and the waves of a SIMD finish together.
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
}


} // namespace ofx_dev
#include "lk_body_warp.h" // the warp of lk_iter in two stages (ITER == 2 below)
#include "lk_body_buf.h" // lk_wave_buf: the same march on buffer resources (a fifth of the scalar instructions)
#include "lk_body_wide.h" // lk_wave_wide: eight columns per lane (round 4)
namespace ofx_dev {

// One wave of the fused level kernel: picks the variant for its tile (wave-uniform: two complete copies of the march, nothing
// merges after them).
// LDS_ROWS: xlds holds kLkWaveLdsDma bytes and the march fetches its rows through LDS (lk_body_buf.h, DMA)
// ITER: lk_wave_buf's (0, or 3 = also write the warped image of the pair's second iteration); buffer march only
template <int R, int MODE, bool SUMS, bool MAY_ACC = true, bool FAST = false, bool LDS_ROWS = false, int ITER = 0>
__device__ __forceinline__ void lk_wave(const LkTable &T, int wave, int lane, uint8_t *xlds)
{
    if (wave >= T.first_block[T.n]) return;
    int level = 0, hi = T.n;
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (wave >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    const int tile = (wave - T.first_block[level]) % T.lv[level].tiles_x;
    const int cb0 = tile * TileGeom<R>::OUT_W - TileGeom<R>::LO_LANE * 4;
    if constexpr (OFX_LK_BUFFER_PATH && !SUMS && !MAY_ACC) { // (the host keeps levels of 2 GB and more out of such launches)
#if OFX_LK_INTERIOR_VARIANT
        if (cb0 >= 0 && cb0 + 256 <= T.lv[level].w) lk_wave_buf<R, MODE, FAST, true, LDS_ROWS, ITER>(T, wave, lane, xlds);
        else
#endif
            lk_wave_buf<R, MODE, FAST, false, LDS_ROWS, ITER>(T, wave, lane, xlds);
        return;
    }
#if OFX_LK_INTERIOR_VARIANT
    if (cb0 >= 0 && cb0 + 256 <= T.lv[level].w) lk_wave_impl<R, MODE, SUMS, MAY_ACC, FAST, true>(T, wave, lane, xlds);
    else
#endif
        lk_wave_impl<R, MODE, SUMS, MAY_ACC, FAST, false>(T, wave, lane, xlds);
}

} // namespace ofx_dev
