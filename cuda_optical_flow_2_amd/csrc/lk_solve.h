// The per-pixel 2x2 solve shared by the fused level kernel and the corner kernel (both must produce bit-identical
// flow for pixel 0 of a level).
#pragma once

#include <hip/hip_runtime.h>

#include "ofx.h"

// ---- 2x2 solve ---------------------------------------------------------------------------------------------
// MODE 1: gpu::inverse_matrix_float, OptFlowGpu.cu:1833-1845 -- the sums are float planes there, so each exact
//         integer sum is rounded once to float first.
// MODE 0: inline loop of cpu::calc_optical_flow, OptFlowCPU.cpp:369-382 -- int sums, `c` left unscaled.
// Same operation order as the reference, in double, with IEEE division; this file is built with
// -ffp-contract=off so no product/sum pair is fused.
// 1/x in double, correctly rounded, for x an integer-valued double (|x| < 2^63: no subnormals, no overflow).
// v_rcp_f64 seed p0 with relative error e0, |e0| < 2^-22; one third-order step p1 = p0*(1 + e + e^2) with
// e = fma(-x,p0,1): the exact p0*(1+e0+e0^2) misses 1/x by e0^3/x < 2^-66/x, so after the fma's rounding p1 is within
// 1 ulp of 1/x; then Markstein's correction r = fma(-x,p1,1); p = fma(p1,r,p1), which turns an approximation within
// 1 ulp into the correctly rounded quotient unless x's significand is all ones (53 one bits: |x| >= 2^52 and a 2^-52
// coincidence on top; the quotient is then still within 1 ulp of a double, which the final rounding to float absorbs).
// x == 0 gives +-Inf like the IEEE division the reference performs.  7 instructions instead of the ~13 of the generic
// division expansion (div_scale/div_fmas handle ranges that cannot occur here).
__device__ __forceinline__ double recip_f64(double x)
{
    double p = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, p, 1.0);
    const double t = __builtin_fma(e, e, e);
    p = __builtin_fma(p, t, p);
    e = __builtin_fma(-x, p, 1.0);
    p = __builtin_fma(p, e, p);
    // x == 0: the steps above turned the +-Inf seed into NaN; v_div_fixup_f64 puts the IEEE special cases of 1/x back
    // (+-Inf for +-0) and passes every other quotient through -- one instruction instead of a compare and two selects
    return __builtin_amdgcn_div_fixup(p, x, 1.0);
}

// What a launch asks of the solve besides the mode's arithmetic.
//   fast (OFX_MODE_LK_FLOAT_FAST): the <= 1 ulp formulation below instead of the replay of the reference's operation order.
//   min_det > 0 (extension): a pixel whose determinant, rounded to float, is below it gets the flow (0, 0)
//   instead of the reference's unguarded quotient (NaN / Inf / huge where the window has no texture or only an edge).
struct SolveOpts {
    float min_det; // <= 0: the reference (no guard)
};

// the reference's operation order from the converted operands and the determinant on (exact replay)
template <int MODE>
__device__ __forceinline__ void solve_tail_exact(double a, double b, double d, double xt, double yt, double det, float &u, float &v)
{
    const double pre = recip_f64(det);
    double c = b;
    a *= pre;
    b *= pre;
    if constexpr (MODE == OFX_MODE_LK_FLOAT) c *= pre;
    d *= pre;
    u = (float)(-d * xt + b * yt);
    v = (float)(c * xt - a * yt);
}

// FAST: lk_float only.  The reference computes u = -(d/det)*xt + (b/det)*yt in double and rounds to float.  Here the
// numerators come first: with 24-bit operands d*xt and b*yt are exact in double, so nu = fma(b, yt, -(d*xt)) is the exact
// numerator rounded ONCE, and u = float(nu * p) with p = 1/det good to 2^-44 (seed + one Newton step) -- 24 instructions
// per pixel instead of 30.  Against the replay this is within 1 float ulp (measured: identical bits for 99.999 % of the
// pixels of every test image, 1 ulp for the rest; the bound fails only where the reference's own rounding noise,
// 3 * 2^-53 * (|d' xt| + |b' yt|), exceeds a float ulp of the result, i.e. a numerator that cancels to less than 2^-27 of
// its terms).  A pixel with det == 0 takes the replay's path, so NaN / Inf appear exactly where the reference produces
// them; the choice is per pixel (a lane's result never depends on its neighbours in the wave), which keeps sharded and
// tiled runs bit-identical to whole-frame ones.  The test for it costs one compare per pixel: solve_fast reports the lanes
// with det == 0 as a wave mask (scalar registers) and the caller branches on it.
template <int MODE>
__device__ __forceinline__ void solve_operands(int sxx, int syy, int sxy, int sxt, int syt, double &a, double &b, double &d, double &xt,
                                               double &yt, double &det)
{
    if constexpr (MODE == OFX_MODE_LK_FLOAT) {
        a = (double)(float)sxx;
        b = (double)(float)sxy;
        d = (double)(float)syy;
        xt = (double)(float)sxt;
        yt = (double)(float)syt;
        // a, b, d carry 24 significant bits, so a*d and b*c are exact in double and (a*d) - (b*c) rounds once: the fused
        // form rounds the same exact difference once -- identical bits, one instruction less
        det = __builtin_fma(a, d, -(b * b));
    } else {
        a = (double)sxx;
        b = (double)sxy;
        d = (double)syy;
        xt = (double)sxt;
        yt = (double)syt;
        det = a * d - b * b; // 31-bit factors: the products themselves round, keep the reference's three operations
    }
}

// returns the wave mask of the lanes whose determinant is zero (their u, v are not the reference's yet: solve_fix_singular)
__device__ __forceinline__ unsigned long long solve_fast(int sxx, int syy, int sxy, int sxt, int syt, float &u, float &v)
{
    double a, b, d, xt, yt, det;
    solve_operands<OFX_MODE_LK_FLOAT>(sxx, syy, sxy, sxt, syt, a, b, d, xt, yt, det);
    const double nu = __builtin_fma(b, yt, -(d * xt));
    const double nv = __builtin_fma(b, xt, -(a * yt));
    double p = __builtin_amdgcn_rcp(det);
    const double e = __builtin_fma(-det, p, 1.0);
    p = __builtin_fma(p, e, p);
    u = (float)(nu * p);
    v = (float)(nv * p);
    return __ballot(det == 0.0);
}

// the rare path behind solve_fast: the lanes of `zero` take the replay's result
__device__ __forceinline__ void solve_fix_singular(int sxx, int syy, int sxy, int sxt, int syt, float &u, float &v)
{
    double a, b, d, xt, yt, det;
    solve_operands<OFX_MODE_LK_FLOAT>(sxx, syy, sxy, sxt, syt, a, b, d, xt, yt, det);
    float eu, ev;
    solve_tail_exact<OFX_MODE_LK_FLOAT>(a, b, d, xt, yt, det, eu, ev);
    if (det == 0.0) {
        u = eu;
        v = ev;
    }
}

// the determinant guard (SolveOpts::min_det), applied after the solve where it is on
template <int MODE>
__device__ __forceinline__ void solve_guard(int sxx, int syy, int sxy, float min_det, float &u, float &v)
{
    double a, b, d, xt, yt, det;
    solve_operands<MODE>(sxx, syy, sxy, 0, 0, a, b, d, xt, yt, det);
    if (!((float)det >= min_det)) {
        u = 0.0f;
        v = 0.0f;
    }
}

// One pixel, complete (corner kernel).
template <int MODE, bool FAST>
__device__ __forceinline__ void solve2x2(int sxx, int syy, int sxy, int sxt, int syt, const SolveOpts &opt, float &u, float &v)
{
    static_assert(!FAST || MODE == OFX_MODE_LK_FLOAT, "the fast solve is defined for lk_float only");
    if constexpr (FAST) {
        if (solve_fast(sxx, syy, sxy, sxt, syt, u, v) != 0ull) solve_fix_singular(sxx, syy, sxy, sxt, syt, u, v);
    } else {
        double a, b, d, xt, yt, det;
        solve_operands<MODE>(sxx, syy, sxy, sxt, syt, a, b, d, xt, yt, det);
        solve_tail_exact<MODE>(a, b, d, xt, yt, det, u, v);
    }
    if (__builtin_expect(opt.min_det > 0.0f, 0)) { // wave-uniform
        // the empty asm keeps this block a real branch: if-converted (as hipcc does with a plain block), the guard's
        // conversion, compare and two selects ran for every pixel of every launch -- 5 % of the level kernel -- while off
        asm volatile("" : "+v"(u), "+v"(v));
        solve_guard<MODE>(sxx, syy, sxy, opt.min_det, u, v);
    }
}

// The 4 pixels of a lane of the level kernel.
template <int MODE, bool FAST>
__device__ __forceinline__ void solve_lane(const int (&sxx)[4], const int (&syy)[4], const int (&sxy)[4], const int (&sxt)[4],
                                           const int (&syt)[4], const SolveOpts &opt, float (&uv)[8])
{
    if constexpr (FAST) {
        // (pixel by pixel: a pixel's five sums die with its solve -- one branch per row step for all four pixels kept all
        // twenty sums alive for the singular path and spilled)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (__builtin_expect(solve_fast(sxx[j], syy[j], sxy[j], sxt[j], syt[j], uv[2 * j], uv[2 * j + 1]) != 0ull, 0)) {
                asm volatile("" : "+v"(uv[2 * j]), "+v"(uv[2 * j + 1]));
                solve_fix_singular(sxx[j], syy[j], sxy[j], sxt[j], syt[j], uv[2 * j], uv[2 * j + 1]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a, b, d, xt, yt, det;
            solve_operands<MODE>(sxx[j], syy[j], sxy[j], sxt[j], syt[j], a, b, d, xt, yt, det);
            solve_tail_exact<MODE>(a, b, d, xt, yt, det, uv[2 * j], uv[2 * j + 1]);
        }
    }
    if (__builtin_expect(opt.min_det > 0.0f, 0)) {
        asm volatile("" : "+v"(uv[0]), "+v"(uv[1]), "+v"(uv[2]), "+v"(uv[3]), "+v"(uv[4]), "+v"(uv[5]), "+v"(uv[6]), "+v"(uv[7]));
#pragma unroll
        for (int j = 0; j < 4; ++j) solve_guard<MODE>(sxx[j], syy[j], sxy[j], opt.min_det, uv[2 * j], uv[2 * j + 1]);
    }
}
