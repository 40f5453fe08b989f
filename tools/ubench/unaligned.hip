// Is a misaligned global_load_dword slower than an aligned one on gfx950?  (decides how the fused shift reads rows)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void rd(const uint8_t *p, uint32_t *o, size_t n, int off)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (size_t k = i * 4; k + 8 < n; k += (size_t)gridDim.x * blockDim.x * 4) {
        uint32_t v;
        __builtin_memcpy(&v, p + k + off, 4);
        acc ^= v;
    }
    if (acc == 0x12345678) o[0] = acc;
}
int main()
{
    const size_t n = (size_t)256 << 20;
    uint8_t *p; uint32_t *o;
    hipMalloc(&p, n + 64); hipMalloc(&o, 64); hipMemset(p, 1, n + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int off = 0; off < 4; ++off) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            rd<<<4096, 256>>>(p, o, n, off);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("offset %d: %.3f ms  %.0f GB/s\n", off, best, n / best / 1e6);
    }
    return 0;
}
