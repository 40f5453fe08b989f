// Micro-benchmark: ns per ds_read_b32 (per SIMD, all four SIMDs of every CU busy, four waves each) by the pattern of the lanes'
// addresses: how a table lookup's cost depends on how many distinct entries a wave touches and where they sit.
//   hipcc -O3 --offload-arch=gfx950 -o lds_conflicts lds_conflicts.hip && ./lds_conflicts
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k(int *out, int n, int pattern, int span)
{
    __shared__ float tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    int a[8];
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        int e;
        if (pattern == 0) e = 0;                       // one address for the whole wave
        else if (pattern == 1) e = lane % span;        // span distinct entries, neighbours differ
        else if (pattern == 2) e = (lane * span) >> 6; // span distinct entries, in runs of equal neighbours
        else if (pattern == 3) e = (h >> 8) % span;    // random among span entries
        else e = lane * span;                          // stride of span entries
        a[i] = (e * 4 + i * 2048) & 32767;
    }
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += f[i];
    if (r == 12345.678f) out[threadIdx.x] = (int)r;
}

double run(int pattern, int span, int n)
{
    int *out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<1024, 256>>>(out, 64, pattern, span);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        k<<<1024, 256>>>(out, n, pattern, span);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipFree(out);
    return best * 1e-3 / (4.0 * n * 32.0) * 1e9;
}

int main()
{
    const int n = 2000;
    printf("%-44s %8s\n", "lanes' addresses", "ns/read");
    printf("%-44s %8.3f\n", "one address", run(0, 1, n));
    for (int s : {2, 4, 8, 16, 32, 64}) { char b[64]; snprintf(b, 64, "lane %% %d", s); printf("%-44s %8.3f\n", b, run(1, s, n)); }
    for (int s : {2, 4, 8, 16, 32}) { char b[64]; snprintf(b, 64, "%d runs of equal neighbours", s); printf("%-44s %8.3f\n", b, run(2, s, n)); }
    for (int s : {2, 4, 8, 16, 32, 64, 128, 256}) { char b[64]; snprintf(b, 64, "random among %d entries", s); printf("%-44s %8.3f\n", b, run(3, s, n)); }
    for (int s : {1, 2, 3, 4, 8, 16, 32}) { char b[64]; snprintf(b, 64, "stride of %d entries", s); printf("%-44s %8.3f\n", b, run(4, s, n)); }
    return 0;
}
