// Corner kernel: the flow of pixel 0 at every pyramid level, coarse to fine, in one single-wave launch.
//
// The reference shifts `next` at level k by (u,v) = sum_{j>k} 2^(j-k) * flow_j[pixel 0] (OptFlowCPU.cpp:255-266:
// `i * (1 >> offset)` is 0 for every offset >= 1, so only element 0 of each coarser flow level is ever read).  Hence
// the only dependency between the levels of a frame pair is a chain through PIXEL 0, whose flow depends on the
// (radius+1)^2 clipped window at the top-left corner of its level.  This kernel walks that chain once -- per level:
// shift vector, derivatives of the shifted corner, five window sums, the 2x2 solve (lk_solve.h, the same code the
// fused level kernel uses, so the values are bit-identical) -- and publishes every level's shift vector.  Afterwards
// all levels can be shifted and solved concurrently (ofx_shift_levels, ofx_lk_levels).
#include "lk_solve.h"
#include "ofx_internal.h"

namespace {

struct CornerLevel {
    const uint8_t *prev;
    const uint8_t *next; // unshifted
    float *flow;         // pixel 0 is written when flow_row0 == 0
    int w, h, pitch, row_end, flow_row0;
};

struct CornerArgs {
    CornerLevel lv[OFX_MAX_LEVELS];
    float *uv; // 2 floats per level
    int levels, radius;
};

__device__ __forceinline__ int pix(const uint8_t *img, const CornerLevel &L, int x, int y)
{
    return (x >= 0 && x < L.w && y >= 0 && y < L.h && y < L.row_end) ? (int)img[(size_t)y * (size_t)L.pitch + x] : 0;
}

// cpu::shift_back_pyramid on channel 0 for one pixel (same rule as shift_1ch_kernel in pyramid.hip)
__device__ __forceinline__ int shifted_next(const CornerLevel &L, int x, int y, bool shifted, float u, float v)
{
    if (x < 0 || x >= L.w || y < 0 || y >= L.h) return 0;
    if (!shifted) return pix(L.next, L, x, y);
    const float ty = (float)y + v, tx = (float)x + u;
    const bool yin = ty > -1.0f && ty < (float)L.h;
    const int ny = yin ? (int)ty : 0;
    if (yin && ny < L.row_end && tx > -1.0f && tx < (float)L.w) return (int)L.next[(size_t)ny * (size_t)L.pitch + (int)tx];
    return (3ll * ((long long)y * L.w + x) < (long long)L.w * (long long)L.h) ? pix(L.next, L, x, y) : 0;
}

template <int MODE>
__global__ __launch_bounds__(64) void corner_kernel(const CornerArgs A)
{
    __shared__ float f0[OFX_MAX_LEVELS][2];
    const int lane = threadIdx.x;
    for (int k = A.levels - 1; k >= 0; --k) {
        const CornerLevel &L = A.lv[k];
        // shift vector: float accumulation, coarsest level first (OptFlowCPU.cpp:257-266)
        float u = 0.0f, v = 0.0f;
        for (int j = A.levels - 1; j > k; --j) {
            const float mult = (float)(1 << (j - k));
            u += mult * f0[j][0];
            v += mult * f0[j][1];
        }
        const bool shifted = k != A.levels - 1;
        if (shifted && lane == 0) {
            A.uv[2 * k] = u;
            A.uv[2 * k + 1] = v;
        }
        // window of pixel 0, clipped to the image: taps [0..R] x [0..R]
        const int tw = min(A.radius + 1, L.w), th = min(A.radius + 1, L.h);
        int sxx = 0, syy = 0, sxy = 0, sxt = 0, syt = 0;
        for (int t = lane; t < tw * th; t += 64) {
            const int x = t % tw, y = t / tw;
            int p[3][3], q[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    p[i][j] = pix(L.prev, L, x - 1 + j, y - 1 + i);
                    q[i][j] = shifted_next(L, x - 1 + j, y - 1 + i, shifted, u, v);
                }
            int ix = (p[0][2] + 2 * p[1][2] + p[2][2]) - (p[0][0] + 2 * p[1][0] + p[2][0]); // Dx_3x3, kernels.cpp:6-10
            int iy = (p[2][0] + 2 * p[2][1] + p[2][2]) - (p[0][0] + 2 * p[0][1] + p[0][2]); // Dy_3x3, kernels.cpp:15-19
            int it;
            if constexpr (MODE == OFX_MODE_LK_FLOAT) {
                // Dt_3x3 (kernels.cpp:20-24) on next - prev
                int d[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) d[i][j] = q[i][j] - p[i][j];
                it = (d[0][0] + d[0][2] + d[2][0] + d[2][2]) + 2 * (d[0][1] + d[1][0] + d[1][2] + d[2][1]) + 3 * d[1][1];
            } else {
                // per-tap truncated Gaussian (OptFlowCPU.cpp:102 with GAUS_KERNEL_3x3), u8 wrap (:106, :15)
                const int gp = (p[0][0] >> 4) + (p[0][2] >> 4) + (p[2][0] >> 4) + (p[2][2] >> 4) + (p[0][1] >> 3) + (p[1][0] >> 3) +
                               (p[1][2] >> 3) + (p[2][1] >> 3) + (p[1][1] >> 2);
                const int gq = (q[0][0] >> 4) + (q[0][2] >> 4) + (q[2][0] >> 4) + (q[2][2] >> 4) + (q[0][1] >> 3) + (q[1][0] >> 3) +
                               (q[1][2] >> 3) + (q[2][1] >> 3) + (q[1][1] >> 2);
                ix &= 0xff;
                iy &= 0xff;
                it = (gq - gp) & 0xff;
            }
            sxx += ix * ix;
            syy += iy * iy;
            sxy += ix * iy;
            sxt += ix * it;
            syt += iy * it;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            sxx += __shfl_xor(sxx, m);
            syy += __shfl_xor(syy, m);
            sxy += __shfl_xor(sxy, m);
            sxt += __shfl_xor(sxt, m);
            syt += __shfl_xor(syt, m);
        }
        float fu, fv;
        solve2x2<MODE>(sxx, syy, sxy, sxt, syt, fu, fv);
        if (lane == 0) {
            f0[k][0] = fu;
            f0[k][1] = fv;
            if (L.flow != nullptr && L.flow_row0 == 0) {
                L.flow[0] = fu;
                L.flow[1] = fv;
            }
        }
        __syncthreads();
    }
}

} // namespace

extern "C" int ofx_corner_flows(const ofx_lk_desc *levels, int n_levels, int window, int mode, float *d_uv, void *stream)
{
    OFX_REQUIRE(levels && d_uv && n_levels >= 1 && n_levels <= OFX_MAX_LEVELS, "ofx_corner_flows: bad arguments");
    OFX_REQUIRE(window >= 3 && (window & 1), "ofx_corner_flows: window must be odd and >= 3 (got %d)", window);
    OFX_REQUIRE(mode == OFX_MODE_COMPAT_CPU || mode == OFX_MODE_LK_FLOAT, "ofx_corner_flows: bad mode %d", mode);
    CornerArgs a{};
    a.levels = n_levels;
    a.radius = window >> 1;
    a.uv = d_uv;
    for (int k = 0; k < n_levels; ++k) {
        const ofx_geom *g = &levels[k].geom;
        OFX_TRY(ofx_check_geom(g, "ofx_corner_flows"));
        OFX_REQUIRE(levels[k].d_prev && levels[k].d_next, "ofx_corner_flows: null plane at level %d", k);
        OFX_REQUIRE(g->row0 == 0, "ofx_corner_flows: level %d buffer must start at row 0 (it holds rows from %d)", k, g->row0);
        const int need = a.radius + 2 < g->h ? a.radius + 2 : g->h;
        OFX_REQUIRE(g->rows >= need, "ofx_corner_flows: level %d holds %d rows, the corner needs %d", k, g->rows, need);
        a.lv[k] = CornerLevel{levels[k].d_prev, levels[k].d_next, levels[k].d_flow, g->w, g->h, g->pitch, g->rows, levels[k].flow_row0};
    }
    if (mode == OFX_MODE_LK_FLOAT)
        hipLaunchKernelGGL(corner_kernel<OFX_MODE_LK_FLOAT>, dim3(1), dim3(64), 0, ofx_stream(stream), a);
    else
        hipLaunchKernelGGL(corner_kernel<OFX_MODE_COMPAT_CPU>, dim3(1), dim3(64), 0, ofx_stream(stream), a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
