// Calibrates rocprofv3's FETCH_SIZE on gfx950 for the access widths this project uses: the guide's "x2" correction was
// measured for 16 B/lane streaming reads; the LK march reads one dword per lane (256 B per wave instruction).
// Run under `rocprofv3 --pmc FETCH_SIZE`: each kernel reads the same 256 MiB buffer exactly once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <typename T>
__global__ void rd(const T *p, uint32_t *o, size_t n_elems)
{
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t k = i0; k < n_elems; k += stride) {
        T v = p[k];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (unsigned j = 0; j < sizeof(T) / 4; ++j) acc ^= w[j];
    }
    if (acc == 0x12345678) o[0] = acc;
}
int main()
{
    const size_t n = (size_t)256 << 20;
    uint8_t *p;
    uint32_t *o;
    hipMalloc(&p, n);
    hipMalloc(&o, 64);
    hipMemset(p, 1, n);
    hipDeviceSynchronize();
    for (int r = 0; r < 3; ++r) {
        rd<uint32_t><<<8192, 256>>>(reinterpret_cast<const uint32_t *>(p), o, n / 4);
        rd<uint2><<<8192, 256>>>(reinterpret_cast<const uint2 *>(p), o, n / 8);
        rd<uint4><<<8192, 256>>>(reinterpret_cast<const uint4 *>(p), o, n / 16);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes once\n", n);
    return 0;
}
