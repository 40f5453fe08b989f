// Micro-benchmark: sustained cycles per wave-instruction on one SIMD for the VALU ops the LK kernel uses.
// 8 independent chains per lane, N iterations, grid sized to give every SIMD `W` waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

typedef short short2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(int *out, int n, int seed)
{
    int a[8], b = seed | 1, c = seed + 3;
    unsigned long long m64 = 0x5555555555555555ull + (unsigned)seed;
    asm volatile("" : "+s"(m64));
    if constexpr (OP == 27) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(b), "v"(c) : "vcc");
    double d[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i + seed; d[i] = a[i] * 0.5 + 1.0; f[i] = a[i] * 0.25f + 1.0f; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 1) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 2) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 4) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
                if constexpr (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
                if constexpr (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if constexpr (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if constexpr (OP == 8) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 9) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 10) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 11) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 12) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 13) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 14) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                if constexpr (OP == 15) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 16) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 18) asm volatile("v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 19) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
                if constexpr (OP == 20) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if constexpr (OP == 21) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if constexpr (OP == 22) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if constexpr (OP == 23) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 24) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(b) : "vcc");
                if constexpr (OP == 25) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 26) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 27) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 28) asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 29) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 30) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
                if constexpr (OP == 31) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 32) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m64));
                if constexpr (OP == 33) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 34) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == 35) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if constexpr (OP == 36) { int sr; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr) : "v"(a[i])); }
                if constexpr (OP == 37) asm volatile("v_sub_u32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 38) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                if constexpr (OP == 39) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == 40) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if constexpr (OP == 41) asm volatile("v_div_fixup_f64 %0, %0, %1, 1.0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if constexpr (OP == 42) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                if constexpr (OP == 43) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            }
        }
    }
    int r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + (int)d[i] + (int)f[i];
    if (r == 0x7fffffff) out[threadIdx.x] = r;
}

template <int OP>
double run(int waves_per_simd, int n)
{
    int *out;
    hipMalloc(&out, 4096);
    const int blocks = 256 * waves_per_simd; // 256 CUs x (4 waves per block = 1 per SIMD) x W
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 64, 1);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(out, n, r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFree(out);
    // wave-instructions per SIMD = W * n * 32
    return best * 1e-3 / ((double)waves_per_simd * n * 32.0); // seconds per wave-instruction per SIMD
}

int main()
{
    const char *names[] = {"v_add_u32", "v_mad_i32_i24", "v_mul_i32_i24", "v_and_b32", "v_bfe_u32", "v_mov_dpp wave_shr", "v_add_f32",
                           "v_fma_f32", "v_pk_add_u16", "v_pk_mad_u16", "v_dot2_i32_i16", "v_fma_f64", "v_mul_f64", "v_add_f64",
                           "v_cvt_f64_i32", "v_lshl_add_u32", "v_add3_u32", "v_perm_b32", "v_mul_i24_sdwa", "v_rcp_f64", "v_cvt_f32_i32",
                           "v_cvt_f64_f32", "v_cvt_f32_f64", "v_pk_fma_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_pk_mul_lo_u16", "v_cndmask",
                           "v_add_u32_dpp", "v_pk_sub_i16", "v_mov_dpp row_shr", "v_sad_u16", "v_cndmask_e64 sgpr", "v_or3_b32",
                           "v_max3_i32", "v_cmp_lt_i32 vcc", "v_readfirstlane", "v_sub_u32_dpp shl", "v_lshlrev_b32", "v_sub_u32", "v_cvt_f32_u32",
                           "v_div_fixup_f64", "v_mov_b32", "v_rcp_f32"};
    const int n = 4000;
    printf("%-22s %10s %10s %10s   (ns per wave-instr per SIMD; x2.4 = cycles @2.4GHz)\n", "op", "W=1", "W=2", "W=4");
#define ROW(OP) { double a = run<OP>(1, n), b = run<OP>(2, n), c = run<OP>(4, n); \
    printf("%-22s %10.3f %10.3f %10.3f   cyc@2.4: %5.2f %5.2f %5.2f\n", names[OP], a*1e9, b*1e9, c*1e9, a*2.4e9, b*2.4e9, c*2.4e9); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16)
    ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31)
    ROW(32) ROW(33) ROW(34) ROW(35) ROW(36) ROW(37) ROW(38) ROW(39) ROW(40) ROW(41) ROW(42) ROW(43)
    return 0;
}
