// Checks buffer_load_dword ... lds (LDS-DMA) as the LK march would use it: lane-linear placement in LDS, an unaligned byte
// offset per lane, a marker row (0x80000000) reading zeros, and a counted s_waitcnt that leaves younger operations in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

__global__ void k(const uint8_t *src, int n_src, uint32_t *out, int soff_unaligned)
{
    __shared__ __attribute__((aligned(16))) uint32_t buf[4 * 64];
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n_src, 0x00027000);
    auto lds = [&](int row) { return (__attribute__((address_space(3))) void *)(buf + 64 * row); };
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds(0), 4, 4 * l, 1024, 0, 0);            // aligned
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds(1), 4, 4 * l, soff_unaligned, 0, 0);  // unaligned row offset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds(2), 4, 4 * l + (l & 3), 2048, 0, 0);  // unaligned lane offsets
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds(3), 4, 4 * l, (int)0x80000000, 0, 0); // marker row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int row = 0; row < 4; ++row) out[64 * row + l] = buf[64 * row + l];
}

int main()
{
    const int n = 8192;
    std::vector<uint8_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = (uint8_t)(i * 13 + 5);
    uint8_t *src; uint32_t *out;
    (void)hipMalloc(&src, n); (void)hipMalloc(&out, 256 * 4);
    (void)hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice);
    (void)hipMemset(out, 0xEE, 256 * 4);
    k<<<1, 64>>>(src, n, out, 3077);
    std::vector<uint32_t> o(256);
    if (hipMemcpy(o.data(), out, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: kernel fault\n"); return 1; }
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        uint32_t w0, w1, w2;
        memcpy(&w0, &h[1024 + 4 * l], 4); memcpy(&w1, &h[3077 + 4 * l], 4); memcpy(&w2, &h[2048 + 4 * l + (l & 3)], 4);
        const uint32_t want[4] = {w0, w1, w2, 0u};
        for (int row = 0; row < 4; ++row)
            if (o[64 * row + l] != want[row]) { if (bad < 8) printf("row %d lane %d: %08x want %08x\n", row, l, o[64 * row + l], want[row]); ++bad; }
    }
    printf(bad ? "FAIL: %d mismatches\n" : "LDS-DMA behaves as needed (lane-linear, unaligned offsets, marker row)\n", bad);
    return bad ? 1 : 0;
}
