// How fast can gfx950 WRITE, and does the shape of a wave's store matter?  The level kernel writes 8 of its 10 B/px (the
// flow), 32 contiguous bytes per lane and row; this measures pure streaming writes of the same volume in several shapes:
//   A  one dwordx4 per lane, lanes contiguous (16 B lane stride): a wave instruction covers 1 KB without gaps
//   B  two dwordx4 per lane at a 32 B lane stride (the level kernel's shape): each instruction covers every other 16 B
//   C  four dwordx2 per lane at a 32 B lane stride
//   D  as B, but rows of 2 KB per wave written one row per iteration by waves that march down (stride = row pitch)
// each with plain and with non-temporal stores; plus a read-only and a copy kernel for scale.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define GLOBAL __attribute__((address_space(1)))

template <bool NT, typename T>
__device__ __forceinline__ void st(T *p, T v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void wr_a(float *dst, size_t n_f4, float seed)
{
    f4 *d = (f4 *)dst;
    const f4 v = {seed, seed + 1, seed + 2, seed + 3};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * blockDim.x) st<NT>(d + i, v);
}
template <bool NT>
__global__ __launch_bounds__(256) void wr_b(float *dst, size_t n_f4, float seed)
{
    f4 *d = (f4 *)dst;
    const f4 v = {seed, seed + 1, seed + 2, seed + 3};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n_f4; i += (size_t)gridDim.x * blockDim.x) {
        st<NT>(d + 2 * i, v);
        st<NT>(d + 2 * i + 1, v);
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void wr_c(float *dst, size_t n_f4, float seed)
{
    f2 *d = (f2 *)dst;
    const f2 v = {seed, seed + 1};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n_f4; i += (size_t)gridDim.x * blockDim.x) {
        st<NT>(d + 4 * i, v);
        st<NT>(d + 4 * i + 1, v);
        st<NT>(d + 4 * i + 2, v);
        st<NT>(d + 4 * i + 3, v);
    }
}
// marching waves: wave w owns a 2 KB-wide column of the "image" (row pitch = waves_x * 2 KB) and walks `rows` rows down
template <bool NT>
__global__ __launch_bounds__(256) void wr_d(float *dst, int waves_x, int rows_per_wave, int rows, float seed)
{
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    const int wx = wave % waves_x, strip = wave / waves_x;
    const size_t pitch_f4 = (size_t)waves_x * 128;
    f4 *d = (f4 *)dst + (size_t)wx * 128 + 2 * lane;
    const f4 v = {seed, seed + 1, seed + 2, seed + 3};
    for (int y = strip * rows_per_wave; y < (strip + 1) * rows_per_wave && y < rows; ++y) {
        st<NT>(d + (size_t)y * pitch_f4, v);
        st<NT>(d + (size_t)y * pitch_f4 + 1, v);
    }
}
// G lanes together write G*16 contiguous bytes per instruction, two instructions cover 2*G*16 bytes: G = 1 is B's shape,
// G = 64 is A's.  (What a lane-to-lane exchange before the level kernel's stores would have to achieve.)
template <bool NT, int G>
__global__ __launch_bounds__(256) void wr_g(float *dst, size_t n_f4, float seed)
{
    f4 *d = (f4 *)dst;
    const f4 v = {seed, seed + 1, seed + 2, seed + 3};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n_f4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t grp = i / G, in = i % G;
        st<NT>(d + 2 * grp * G + in, v);
        st<NT>(d + 2 * grp * G + G + in, v);
    }
}
__global__ __launch_bounds__(256) void rd(const float *src, float *out, size_t n_f4)
{
    const f4 *s = (const f4 *)src;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * blockDim.x) acc += s[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.0f) out[0] = 1;
}
__global__ __launch_bounds__(256) void cp(const float *src, float *dst, size_t n_f4)
{
    const f4 *s = (const f4 *)src;
    f4 *d = (f4 *)dst;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

template <typename F>
static float best_ms(F &&launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 8; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    // four 4K flow fields (354 MB), twice, so that nothing written is still in the 256 MB Infinity Cache when it is rewritten
    const size_t bytes = (size_t)4 * 3840 * 2160 * 8, n_f4 = bytes / 16;
    float *a, *b, *o;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&o, 64);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    const int grid = 256 * 8;
    float *bufs[2] = {a, b};
    int flip = 0;
    auto report = [&](const char *name, float ms, size_t moved) { printf("%-46s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, moved / ms / 1e6); };
    for (int warm = 0; warm < 2; ++warm) {
        report("A  dwordx4 contiguous", best_ms([&] { wr_a<false><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("A  dwordx4 contiguous, nt", best_ms([&] { wr_a<true><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("B  2 x dwordx4, 32 B lane stride", best_ms([&] { wr_b<false><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("B  2 x dwordx4, 32 B lane stride, nt", best_ms([&] { wr_b<true><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("C  4 x dwordx2, 32 B lane stride", best_ms([&] { wr_c<false><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("C  4 x dwordx2, 32 B lane stride, nt", best_ms([&] { wr_c<true><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        // D: a 4 x 4K-wide "image": pitch 15 waves x 2 KB = 30720 B (3840 px x 8 B), 8640 rows, 1020 x 4 waves
        const int waves_x = 15, rows = 4 * 2160, waves = 1020 * 4, strips = waves / waves_x, rpw = (rows + strips - 1) / strips;
        report("D  marching waves (B's shape, row per step)", best_ms([&] { wr_d<false><<<(waves + 3) / 4, 256>>>(bufs[flip ^= 1], waves_x, rpw, rows, 1.f); }), (size_t)waves_x * 2048 * rows);
        report("D  marching waves, nt", best_ms([&] { wr_d<true><<<(waves + 3) / 4, 256>>>(bufs[flip ^= 1], waves_x, rpw, rows, 1.f); }), (size_t)waves_x * 2048 * rows);
        report("G=2  (32 B runs per instruction), nt", best_ms([&] { wr_g<true, 2><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("G=4  (64 B runs), nt", best_ms([&] { wr_g<true, 4><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("G=8  (128 B runs), nt", best_ms([&] { wr_g<true, 8><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("G=16 (256 B runs), nt", best_ms([&] { wr_g<true, 16><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("G=64 (1 KB runs), nt", best_ms([&] { wr_g<true, 64><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("G=4  (64 B runs), plain", best_ms([&] { wr_g<false, 4><<<grid, 256>>>(bufs[flip ^= 1], n_f4, 1.f); }), bytes);
        report("read only (dwordx4)", best_ms([&] { rd<<<grid, 256>>>(bufs[flip ^= 1], o, n_f4); }), bytes);
        report("copy (dwordx4), read + written bytes", best_ms([&] { cp<<<grid, 256>>>(a, b, n_f4); }), 2 * bytes);
        printf("\n");
    }
    return 0;
}
