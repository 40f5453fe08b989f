// Micro-benchmark: sustained ns per wave-instruction (per SIMD, all four SIMDs of every CU busy) of the LDS-side instructions a
// table lookup across the lanes could use, alone and mixed with the VALU work of one bilateral tap.
//   hipcc -O3 --offload-arch=gfx950 -o lds_rates lds_rates.hip && ./lds_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(int *out, int n, int seed)
{
    __shared__ float tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = (float)(i + seed);
    __syncthreads();
    int a[8];
    float f[8], g[8], acc[8];
    const unsigned lim = 0x4B0000FCu;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = ((threadIdx.x * 7 + i * 5 + seed) & 63) << 2; f[i] = a[i] * 0.25f + 1.0f; g[i] = i + 0.5f; acc[i] = 0.0f; }
    const float g0 = (float)seed;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (OP == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if constexpr (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
                if constexpr (OP == 3) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(lim));
                if constexpr (OP == 4) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(f[i]) : "v"(g[i]));
                if constexpr (OP == 5) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)" : "+v"(a[i]));
                if constexpr (OP == 6) { // one tap of the lane-table filter: sub, |.|+2^23, min, permute, add, fma
                    float d, t, wgt;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(f[i]), "v"(g0));
                    asm volatile("v_add_f32_e64 %0, |%1|, %2" : "=v"(t) : "v"(d), "v"(g[0]));
                    asm volatile("v_min_u32 %0, %0, %1" : "+v"(t) : "v"(lim));
                    asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(wgt) : "v"(t), "v"(g[1]));
                    asm volatile("s_waitcnt lgkmcnt(4)");
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(wgt));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(g[i]) : "v"(d), "v"(wgt));
                }
                if constexpr (OP == 7) { // one tap of the exponential filter: sub, mul, fma, exp, add, fma
                    float d, t;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(f[i]), "v"(g0));
                    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(t) : "v"(d));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(g[0]), "v"(g[1]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(t));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(t));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(g[i]) : "v"(d), "v"(t));
                }
                if constexpr (OP == 8) { // the lookup from an LDS table with the lane's own address (bank conflicts as they fall)
                    float d, t, wgt;
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(f[i]), "v"(g0));
                    asm volatile("v_add_f32_e64 %0, |%1|, %2" : "=v"(t) : "v"(d), "v"(g[0]));
                    asm volatile("v_min_u32 %0, %0, %1" : "+v"(t) : "v"(lim));
                    asm volatile("v_and_b32 %0, 0xfc, %0" : "+v"(t));
                    asm volatile("ds_read_b32 %0, %1" : "=v"(wgt) : "v"(t));
                    asm volatile("s_waitcnt lgkmcnt(4)");
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(wgt));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(g[i]) : "v"(d), "v"(wgt));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float r = tab[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + f[i] + g[i] + acc[i];
    if (r == 12345.678f) out[threadIdx.x] = (int)r;
}

template <int OP>
double run(int waves_per_simd, int n)
{
    int *out;
    hipMalloc(&out, 4096);
    const int blocks = 256 * waves_per_simd;
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
    return best * 1e-3 / ((double)waves_per_simd * n * 32.0); // seconds per (wave-instruction or tap) per SIMD
}

int main()
{
    const char *names[] = {"ds_bpermute_b32", "ds_read_b32", "v_exp_f32", "v_min_u32", "v_add_f32 |a|", "ds_swizzle_b32",
                           "tap: lane table", "tap: exponential", "tap: LDS table"};
    const int n = 2000;
    printf("%-22s %10s %10s %10s   (ns per wave-instr / tap per SIMD)\n", "op", "W=1", "W=2", "W=4");
#define ROW(OP) { double a = run<OP>(1, n), b = run<OP>(2, n), c = run<OP>(4, n); printf("%-22s %10.3f %10.3f %10.3f\n", names[OP], a*1e9, b*1e9, c*1e9); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8)
    return 0;
}
