// Device side of the corner kernel (see corner.hip).  Shared with the pipelined stream kernel.
#pragma once

#include "lk_solve.h"
#include "ofx_internal.h"

namespace ofx_dev {

struct CornerLevel {
    const uint8_t *prev;
    const uint8_t *next; // unshifted
    float *flow;         // pixel 0 is written when flow_row0 == 0
    int w, h, pitch, row_end, flow_row0;
    int col_end; // the planes hold columns [0, col_end) and rows [0, row_end) of the w x h level (a top-left patch, or all of it)
};

struct CornerArgs {
    CornerLevel lv[OFX_MAX_LEVELS];
    float *uv;   // 2 floats per level
    int *status; // optional: bit k is set when level k needed a pixel inside the image but outside its planes
    int levels, radius;
};

// Pixel fetches are branch-free: the address is clamped into the buffer and the value masked afterwards, so all the
// loads of a tap are issued back to back and cost one memory round trip (conditional loads made hipcc wait per load).
__device__ __forceinline__ int pix(const uint8_t *img, const CornerLevel &L, int x, int y, int &miss)
{
    const bool inside = x >= 0 && x < L.w && y >= 0 && y < L.h;
    const bool in = inside && y < L.row_end && x < L.col_end;
    miss |= (inside && !in) ? 1 : 0;
    const int cx = min(max(x, 0), min(L.w, L.col_end) - 1), cy = min(max(y, 0), min(L.h, L.row_end) - 1);
    const int v = (int)img[(size_t)cy * (size_t)L.pitch + cx];
    return in ? v : 0;
}

// cpu::shift_back_pyramid on channel 0 for one pixel (same rule as shift_1ch_kernel in pyramid.hip)
__device__ __forceinline__ int shifted_next(const CornerLevel &L, int x, int y, bool shifted, float u, float v, int &miss)
{
    const bool inside = x >= 0 && x < L.w && y >= 0 && y < L.h;
    const int own = pix(L.next, L, x, y, miss);
    const float ty = (float)y + v, tx = (float)x + u;
    const bool yin = ty > -1.0f && ty < (float)L.h;
    const int ny = yin ? (int)ty : 0;
    const bool target = shifted && yin && tx > -1.0f && tx < (float)L.w; // the target pixel exists in the image
    const int tnx = target ? (int)tx : 0;
    const bool hit = target && ny < L.row_end && tnx < L.col_end;
    miss |= (inside && target && !hit) ? 1 : 0;
    const int nx = hit ? tnx : 0;
    const int moved = (int)L.next[(size_t)(hit ? ny : 0) * (size_t)L.pitch + nx];
    const bool keep = 3ll * ((long long)y * L.w + x) < (long long)L.w * (long long)L.h;
    const int val = !shifted ? own : (hit ? moved : (keep ? own : 0));
    return inside ? val : 0;
}

// One wave walks the pyramid coarse to fine.  f0 = 2*OFX_MAX_LEVELS floats of LDS private to this wave (the corner
// flows found so far); only wave-level ordering is needed, so the function can run inside a larger workgroup.
template <int MODE>
__device__ __forceinline__ void corner_wave(const CornerArgs &A, int lane, float *f0)
{
    for (int k = A.levels - 1; k >= 0; --k) {
        const CornerLevel &L = A.lv[k];
        // shift vector: float accumulation, coarsest level first (OptFlowCPU.cpp:257-266)
        float u = 0.0f, v = 0.0f;
        for (int j = A.levels - 1; j > k; --j) {
            const float mult = (float)(1 << (j - k));
            u += mult * f0[2 * j];
            v += mult * f0[2 * j + 1];
        }
        const bool shifted = k != A.levels - 1;
        if (shifted && lane == 0) {
            A.uv[2 * k] = u;
            A.uv[2 * k + 1] = v;
        }
        // window of pixel 0, clipped to the image: taps [0..R] x [0..R]
        const int tw = min(A.radius + 1, L.w), th = min(A.radius + 1, L.h);
        int sxx = 0, syy = 0, sxy = 0, sxt = 0, syt = 0, miss = 0;
        for (int t = lane; t < tw * th; t += 64) {
            const int x = t % tw, y = t / tw;
            int p[3][3], q[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    p[i][j] = pix(L.prev, L, x - 1 + j, y - 1 + i, miss);
                    q[i][j] = shifted_next(L, x - 1 + j, y - 1 + i, shifted, u, v, miss);
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
        if (A.status != nullptr && __any(miss != 0) && lane == 0) atomicOr(A.status, 1 << k);
        if (lane == 0) {
            f0[2 * k] = fu;
            f0[2 * k + 1] = fv;
            if (L.flow != nullptr && L.flow_row0 == 0) {
                L.flow[0] = fu;
                L.flow[1] = fv;
            }
        }
        // LDS operations of one wave execute in order; the fence only stops the compiler from moving the next
        // level's reads of f0 above the store
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace ofx_dev
