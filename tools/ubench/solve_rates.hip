// Micro-benchmark: cycles per pixel of the 2x2 solve variants of lk_solve.h at 4 / 5 waves per SIMD, four pixels per lane as in
// the level kernel.  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I ../../include -I ../../cuda_optical_flow_2_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "lk_solve.h"

// variants: 0 exact replay, 1 FAST with its per-pixel branch, 2 FAST straight line (no det == 0 handling), 3 exact without the
// reciprocal chain (p = 2^-20: what the reciprocal costs), 4 exact with v_cvt_f64_i32 only (what the float rounding costs),
// 5 conversions only
template <int V>
__device__ __forceinline__ void solve4(const int (&sxx)[4], const int (&syy)[4], const int (&sxy)[4], const int (&sxt)[4], const int (&syt)[4], float (&uv)[8])
{
    const SolveOpts opt{0.0f};
    if constexpr (V == 0) solve_lane<OFX_MODE_LK_FLOAT, false>(sxx, syy, sxy, sxt, syt, opt, uv);
    if constexpr (V == 1) solve_lane<OFX_MODE_LK_FLOAT, true>(sxx, syy, sxy, sxt, syt, opt, uv);
    if constexpr (V == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) (void)solve_fast(sxx[j], syy[j], sxy[j], sxt[j], syt[j], uv[2 * j], uv[2 * j + 1]);
    }
    if constexpr (V == 3 || V == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a, b, d, xt, yt, det;
            if constexpr (V == 3) {
                solve_operands<OFX_MODE_LK_FLOAT>(sxx[j], syy[j], sxy[j], sxt[j], syt[j], a, b, d, xt, yt, det);
                const double pre = det * 9.5367431640625e-07;
                double c = b;
                a *= pre; b *= pre; c *= pre; d *= pre;
                uv[2 * j] = (float)(-d * xt + b * yt);
                uv[2 * j + 1] = (float)(c * xt - a * yt);
            } else {
                a = (double)sxx[j]; b = (double)sxy[j]; d = (double)syy[j]; xt = (double)sxt[j]; yt = (double)syt[j];
                det = __builtin_fma(a, d, -(b * b));
                solve_tail_exact<OFX_MODE_LK_FLOAT>(a, b, d, xt, yt, det, uv[2 * j], uv[2 * j + 1]);
            }
        }
    }
    if constexpr (V == 5) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double a = (double)(float)sxx[j], b = (double)(float)sxy[j], d = (double)(float)syy[j], xt = (double)(float)sxt[j], yt = (double)(float)syt[j];
            uv[2 * j] = (float)(a + b + d);
            uv[2 * j + 1] = (float)(xt + yt);
        }
    }
}

template <int V>
__global__ __launch_bounds__(256, 5) void k(float *out, int n, int seed)
{
    int sxx[4], syy[4], sxy[4], sxt[4], syt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sxx[j] = 900000 + 7 * (int)threadIdx.x + j + seed;
        syy[j] = 800000 + 5 * (int)threadIdx.x + 3 * j;
        sxy[j] = 1000 * j - 13 * (int)threadIdx.x;
        sxt[j] = 12345 * (j + 1) - seed;
        syt[j] = -54321 + 17 * (int)threadIdx.x;
    }
    float acc = 0.0f;
    for (int it = 0; it < n; ++it) {
        float uv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sxx[j]), "+v"(syy[j]), "+v"(sxy[j]), "+v"(sxt[j]), "+v"(syt[j]));
        solve4<V>(sxx, syy, sxy, sxt, syt, uv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += uv[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) sxt[j] += 3, syt[j] -= 5;
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int V>
double run(int waves_per_simd, int n)
{
    float *out;
    (void)hipMalloc(&out, 4096);
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V><<<blocks, 256>>>(out, 64, 1);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        k<V><<<blocks, 256>>>(out, n, r);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipFree(out);
    return best * 1e-3 / ((double)waves_per_simd * n * 4.0); // seconds per (wave, pixel-of-a-lane) on one SIMD
}

int main()
{
    const char *names[] = {"exact replay", "FAST (branch per pixel)", "FAST straight line", "exact, no reciprocal", "exact, cvt_f64_i32 only", "conversions only"};
    const int n = 20000;
    printf("%-28s %10s %10s %10s   ns per wave-pixel per SIMD (x2.4 = cycles at 2.4 GHz)\n", "solve variant", "W=1", "W=4", "W=5");
#define ROW(V) { double a = run<V>(1, n), b = run<V>(4, n), c = run<V>(5, n); \
    printf("%-28s %10.2f %10.2f %10.2f   cyc@2.4: %6.1f %6.1f %6.1f\n", names[V], a*1e9, b*1e9, c*1e9, a*2.4e9, b*2.4e9, c*2.4e9); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
    return 0;
}
