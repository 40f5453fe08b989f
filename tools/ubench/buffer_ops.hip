// Checks, on the device, the raw-buffer behaviour the LK march's scalar diet relies on (lk_body.h, "buffer path"):
//   1. buffer_load_dword at an UNALIGNED byte offset returns the four bytes at that offset (the shifted rows);
//   2. an soffset of 0x80000000 (the marker of a row that does not exist) reads 0, whatever the lane offset;
//   3. a store whose lane offset is out of range is dropped, its neighbours' are not;
//   4. offsets are range-checked against num_records as voffset + soffset (no wrap below 2^32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t *src, int n_src, uint32_t *out, uint8_t *dst, int n_dst, int soff_unaligned)
{
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n_src, 0x00027000);
    out[l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, soff_unaligned, 0);               // 1: unaligned
    out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, (int)0x80000000, 0);         // 2: marker row
    out[128 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, n_src - 128, 0);            // 4: the tail: lanes >= 32 are out of range
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, n_dst, 0x00027000);
    const uint32_t voff = (l % 3 == 1) ? 0x80000000u : 16u * l;                               // 3: every third lane dropped
    u32x4 q = {(uint32_t)l, 1u, 2u, 3u};
    __builtin_amdgcn_raw_buffer_store_b128(q, w, voff, 0, 2);
}

int main()
{
    const int n = 4096;
    std::vector<uint8_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *src, *dst; uint32_t *out;
    (void)hipMalloc(&src, n); (void)hipMalloc(&dst, 2048); (void)hipMalloc(&out, 192 * 4);
    (void)hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice);
    (void)hipMemset(dst, 0xEE, 2048);
    k<<<1, 64>>>(src, n, out, dst, 1024, 1029);
    std::vector<uint32_t> o(192); std::vector<uint8_t> d(2048);
    if (hipMemcpy(o.data(), out, 192 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: kernel fault\n"); return 1; }
    (void)hipMemcpy(d.data(), dst, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        uint32_t want; memcpy(&want, &h[1029 + 4 * l], 4);
        if (o[l] != want) { if (bad < 5) printf("unaligned lane %d: %08x want %08x\n", l, o[l], want); ++bad; }
        if (o[64 + l] != 0) { if (bad < 5) printf("marker lane %d: %08x want 0\n", l, o[64 + l]); ++bad; }
        uint32_t tail = 0; if (l < 32) memcpy(&tail, &h[n - 128 + 4 * l], 4);
        if (o[128 + l] != tail) { if (bad < 5) printf("tail lane %d: %08x want %08x\n", l, o[128 + l], tail); ++bad; }
        for (int b = 0; b < 16; ++b) {
            uint8_t want8 = 0xEE;
            if (l % 3 != 1 && 16 * l + 16 <= 1024) { uint32_t q[4] = {(uint32_t)l, 1, 2, 3}; want8 = ((uint8_t *)q)[b]; }
            if (d[16 * l + b] != want8) { if (bad < 5) printf("store lane %d byte %d: %02x want %02x\n", l, b, d[16 * l + b], want8); ++bad; }
        }
    }
    for (int i = 1024; i < 2048; ++i) if (d[i] != 0xEE) { if (bad < 5) printf("store beyond num_records at %d\n", i); ++bad; }
    printf(bad ? "FAIL: %d mismatches\n" : "buffer ops behave as the LK march assumes (unaligned dword, marker row, dropped stores, range check)\n", bad);
    return bad ? 1 : 0;
}
