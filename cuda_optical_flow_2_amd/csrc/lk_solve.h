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

template <int MODE>
__device__ __forceinline__ void solve2x2(int sxx, int syy, int sxy, int sxt, int syt, float &u, float &v)
{
    double a, b, c, d, xt, yt;
    if constexpr (MODE == OFX_MODE_LK_FLOAT) {
        a = (double)(float)sxx;
        b = c = (double)(float)sxy;
        d = (double)(float)syy;
        xt = (double)(float)sxt;
        yt = (double)(float)syt;
    } else {
        a = (double)sxx;
        b = c = (double)sxy;
        d = (double)syy;
        xt = (double)sxt;
        yt = (double)syt;
    }
    double det;
    if constexpr (MODE == OFX_MODE_LK_FLOAT) {
        // a, b, d carry 24 significant bits, so a*d and b*c are exact in double and (a*d) - (b*c) rounds once: the fused
        // form rounds the same exact difference once -- identical bits, one instruction less
        det = __builtin_fma(a, d, -(b * c));
    } else {
        det = a * d - b * c; // 31-bit factors: the products themselves round, keep the reference's three operations
    }
    const double pre = recip_f64(det);
    a *= pre;
    b *= pre;
    if constexpr (MODE == OFX_MODE_LK_FLOAT) c *= pre;
    d *= pre;
    u = (float)(-d * xt + b * yt);
    v = (float)(c * xt - a * yt);
}

