// What MI355X's memory system sustains for the read / write mixes of this project's launches, measured with the simplest possible
// streaming kernels (16 bytes per lane, grid-stride, buffers far larger than the 256 MiB Infinity Cache):
//   read only | write only (nt) | copy A -> B (nt stores) | in-place update of A (read, nt store to the same lines: what a refinement
//   iteration does to the flow) | update of 8/9 of the bytes + 1/9 read-only (the accumulating launch's mix with its images)
// Prints TB/s of total traffic (bytes read + bytes written).    hipcc -O3 --offload-arch=gfx950 rw_mix.hip -o rw_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(u4 *a, u4 *b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (MODE == 0) { // read
            const u4 v = a[i];
            acc ^= v;
        } else if constexpr (MODE == 1) { // write (nt)
            __builtin_nontemporal_store(u4{(uint32_t)i, 1, 2, 3}, &a[i]);
        } else if constexpr (MODE == 2) { // copy
            __builtin_nontemporal_store(a[i], &b[i]);
        } else if constexpr (MODE == 3) { // in-place update
            u4 v = a[i];
            v += 1;
            __builtin_nontemporal_store(v, &a[i]);
        } else if constexpr (MODE == 5) { // write (plain, cached)
            a[i] = u4{(uint32_t)i, 1, 2, 3};
        } else if constexpr (MODE == 6) { // in-place update, plain stores
            u4 v = a[i];
            v += 1;
            a[i] = v;
        } else { // in-place update of a + a read of b at 1/8 of the rate
            u4 v = a[i];
            if ((i & 7) == 0) v ^= b[i >> 3];
            v += 1;
            __builtin_nontemporal_store(v, &a[i]);
        }
    }
    if (MODE == 0 && acc.x == 0x12345678u) a[0] = acc;
}

template <int MODE>
double run(u4 *a, u4 *b, size_t n, double bytes_per_elem)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<256 * 16, 256>>>(a, b, n);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        k<MODE><<<256 * 16, 256>>>(a, b, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return (double)n * bytes_per_elem / (best * 1e-3) / 1e12;
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 2048) << 20, n = bytes / 16;
    u4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    printf("%zu MiB buffers, 16 B per lane, best of 5, TB/s of bytes read + written\n", bytes >> 20);
    printf("read only                 %6.2f\n", run<0>(a, b, n, 16));
    printf("write only (nt)           %6.2f\n", run<1>(a, b, n, 16));
    printf("copy a -> b (nt stores)   %6.2f\n", run<2>(a, b, n, 32));
    printf("in-place update (nt)      %6.2f\n", run<3>(a, b, n, 32));
    printf("update + 1/8 read-only    %6.2f\n", run<4>(a, b, n, 34));
    printf("write only (plain)        %6.2f\n", run<5>(a, b, n, 16));
    printf("in-place update (plain)   %6.2f\n", run<6>(a, b, n, 32));
    return 0;
}
