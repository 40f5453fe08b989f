// Fused dense Lucas-Kanade level kernel for gfx950 (MI355X).
//
// One kernel does what the reference does in ten launches plus two host loops per level
// (OptFlowGpu.cu:1930-1964 / OptFlowCPU.cpp:329-384): 3x3 derivative stencils, the five windowed sums of
// products and the 2x2 solve.  Algorithmic HBM traffic: 2 bytes read + 8 bytes written per pixel.
//
// Structure (DESIGN.md "lk_level"):
//   * one 64-lane wave per workgroup; a lane owns 4 adjacent columns, so a wave spans 256 image columns and
//     every global load is one aligned dword per lane (256 B per wave instruction), every flow store 32 B per lane.
//   * the wave marches DOWN a strip of rows.  Per step it loads one new row of each image, forms the three
//     derivative values of its 4 columns (left/right neighbour columns come from the adjacent lanes through
//     DPP wave shifts, no LDS), and updates 5 x 4 vertical running sums:  V += P(row entering) - P(row leaving).
//     The leaving row's derivatives come back from a lane-private LDS ring (no barriers: a lane only ever
//     reads what it wrote).
//   * the horizontal half of the box sum is done in registers: in-lane prefix/suffix sums plus whole-lane totals
//     of the neighbouring lanes, again through DPP.
//   * all sums are exact int32, so results do not depend on strip/tile/shard boundaries.
//   * window radius R and mode are template parameters; the host dispatches.
//
// Border semantics follow the reference exactly: image taps outside the image contribute nothing
// (OptFlowCPU.cpp:98, OptFlowGpu.cu:1066-1075) and window taps outside the image are skipped
// (OptFlowCPU.cpp:182-191) -- i.e. the image and the derivative planes are zero-extended.
#include "ofx_internal.h"

namespace {

struct LkArgs {
    const uint8_t *prev;
    const uint8_t *next;
    float *flow;   // interleaved (u,v), 2*w floats per row, row (y - flow_row0)
    int32_t *sums; // optional: 5 planes of w ints per row (test/inspection variant)
    size_t sums_plane;
    int w, h, pitch, row0, row_end; // buffer holds global rows [row0,row_end)
    int out_y0, out_y1, flow_row0;
    int strip_h, tiles_x;
};

// value of x held by lane (lane + D); 0 where that lane does not exist.  gfx9 DPP whole-wave shifts.
template <int D>
__device__ __forceinline__ int lane_from(int x)
{
    if constexpr (D == 0) {
        return x;
    } else if constexpr (D > 0) {
        return lane_from<D - 1>(__builtin_amdgcn_update_dpp(0, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
    } else {
        return lane_from<D + 1>(__builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
    }
}

// ---- horizontal box sum over columns [c-R, c+R] for the 4 columns of a lane -------------------------------
// q[k] = a0+..+ak, s[k] = ak+..+a3 (q[3] == s[0] == lane total).
template <int R, int I, int D>
__device__ __forceinline__ int hbox_right(const int (&q)[4])
{
    constexpr int hi = I + R; // last relative column of the window; lane +D holds relative columns 4D..4D+3
    if constexpr (hi < 4 * D) {
        return 0;
    } else if constexpr (hi >= 4 * D + 3) {
        return lane_from<D>(q[3]) + hbox_right<R, I, D + 1>(q);
    } else {
        return lane_from<D>(q[hi - 4 * D]);
    }
}

template <int R, int I, int D>
__device__ __forceinline__ int hbox_left(const int (&s)[4])
{
    constexpr int lo = I - R; // first relative column; lane -D holds relative columns -4D..-4D+3
    if constexpr (lo > -4 * D + 3) {
        return 0;
    } else if constexpr (lo <= -4 * D) {
        return lane_from<-D>(s[0]) + hbox_left<R, I, D + 1>(s);
    } else {
        return lane_from<-D>(s[lo + 4 * D]);
    }
}

template <int R, int I>
__device__ __forceinline__ int hbox_one(const int (&q)[4], const int (&s)[4])
{
    constexpr int lo = I - R, hi = I + R;
    constexpr int olo = lo > 0 ? lo : 0, ohi = hi < 3 ? hi : 3;
    int own;
    if constexpr (olo == 0) {
        own = q[ohi];
    } else if constexpr (ohi == 3) {
        own = s[olo];
    } else {
        own = q[ohi] - q[olo - 1];
    }
    return own + hbox_right<R, I, 1>(q) + hbox_left<R, I, 1>(s);
}

template <int R>
__device__ __forceinline__ void hbox4(const int (&a)[4], int (&out)[4])
{
    int q[4], s[4];
    q[0] = a[0];
    q[1] = q[0] + a[1];
    q[2] = q[1] + a[2];
    q[3] = q[2] + a[3];
    s[3] = a[3];
    s[2] = s[3] + a[2];
    s[1] = s[2] + a[1];
    s[0] = q[3];
    out[0] = hbox_one<R, 0>(q, s);
    out[1] = hbox_one<R, 1>(q, s);
    out[2] = hbox_one<R, 2>(q, s);
    out[3] = hbox_one<R, 3>(q, s);
}

// ---- 2x2 solve ---------------------------------------------------------------------------------------------
// MODE 1: gpu::inverse_matrix_float, OptFlowGpu.cu:1833-1845 -- the sums are float planes there, so each exact
//         integer sum is rounded once to float first.
// MODE 0: inline loop of cpu::calc_optical_flow, OptFlowCPU.cpp:369-382 -- int sums, `c` left unscaled.
// Same operation order as the reference, in double, with IEEE division; this file is built with
// -ffp-contract=off so no product/sum pair is fused.
template <int MODE>
__device__ __forceinline__ void solve2x2(int sxx, int syy, int sxy, int sxt, int syt, float &u, float &v)
{
    double a, b, c, d, xt, yt;
    if constexpr (MODE == OFX_MODE_LK_FLOAT) {
        a = (double)(float)sxx;
        b = c = (double)(float)sxy;
        d = (double)(float)syy;
        xt = (double)(float)sxt;
        yt = (double)(float)syt;
    } else {
        a = (double)sxx;
        b = c = (double)sxy;
        d = (double)syy;
        xt = (double)sxt;
        yt = (double)syt;
    }
    const double pre = 1.0 / (a * d - b * c);
    a *= pre;
    b *= pre;
    if constexpr (MODE == OFX_MODE_LK_FLOAT) c *= pre;
    d *= pre;
    u = (float)(-d * xt + b * yt);
    v = (float)(c * xt - a * yt);
}

__device__ __forceinline__ void unpack4(uint32_t d, int (&o)[4])
{
    o[0] = d & 0xff;
    o[1] = (d >> 8) & 0xff;
    o[2] = (d >> 16) & 0xff;
    o[3] = d >> 24;
}

// geometry of a wave tile for radius R (also used by the host)
template <int R>
struct TileGeom {
    static constexpr int LO_LANE = (R + 1 + 3) / 4; // first lane whose 4 outputs have all their taps inside the wave
    static constexpr int HI_LANE = (251 - R) / 4;   // last such lane (derivatives are valid for wave columns 1..254)
    static constexpr int OUT_W = (HI_LANE - LO_LANE + 1) * 4;
};

template <int R, int MODE, bool SUMS>
__global__ __launch_bounds__(64) void lk_level_kernel(const LkArgs A)
{
    using G = TileGeom<R>;
    constexpr int NS = 2 * R + 1;

    // lane-private ring of the last NS derivative rows of this lane's 4 columns
    __shared__ uint4 ring_a[NS * 64];
    __shared__ uint2 ring_b[MODE == OFX_MODE_LK_FLOAT ? NS * 64 : 1];

    const int lane = threadIdx.x;
    const int tile = blockIdx.x % A.tiles_x;
    const int strip = blockIdx.x / A.tiles_x;
    const int cb = tile * G::OUT_W - G::LO_LANE * 4 + 4 * lane; // first of this lane's 4 image columns
    const int ys = A.out_y0 + strip * A.strip_h;
    const int ye = min(ys + A.strip_h, A.out_y1);

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        ring_a[s * 64 + lane] = make_uint4(0, 0, 0, 0);
        if constexpr (MODE == OFX_MODE_LK_FLOAT) ring_b[s * 64 + lane] = make_uint2(0, 0);
    }

    // column validity: bytes outside [0,w) read as zero, derivatives there are zero
    const bool ld_ok = cb >= 0 && cb < A.w;
    uint32_t bmask = 0;
    int cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = (cb + j) >= 0 && (cb + j) < A.w;
        cv[j] = in ? -1 : 0;
        bmask |= in ? (0xffu << (8 * j)) : 0u;
    }
    const size_t col_off = ld_ok ? (size_t)cb : 0;

    // rows outside the image are the zero border; rows past the last one this strip needs (the loop prefetches one
    // row ahead) or outside the buffer are never dereferenced
    const int y_lim = min(min(ye + R + 1, A.h), A.row_end);
    const int y_min = max(0, A.row0);
    auto load_row = [&](const uint8_t *img, int y) -> uint32_t {
        if (y < y_min || y >= y_lim || !ld_ok) return 0u;
        return *reinterpret_cast<const uint32_t *>(img + (size_t)(y - A.row0) * (size_t)A.pitch + col_off) & bmask;
    };

    // rolling 3-row windows (top, mid, bot) of both images, unpacked
    int pt[4], pm[4], pb[4], nt[4], nm[4], nb[4];
    const int y_first = ys - R; // first derivative row this strip needs
    unpack4(load_row(A.prev, y_first - 1), pm);
    unpack4(load_row(A.next, y_first - 1), nm);
    unpack4(load_row(A.prev, y_first), pb);
    unpack4(load_row(A.next, y_first), nb);
    uint32_t pf_p = load_row(A.prev, y_first + 1);
    uint32_t pf_n = load_row(A.next, y_first + 1);

    int vxx[4] = {0, 0, 0, 0}, vyy[4] = {0, 0, 0, 0}, vxy[4] = {0, 0, 0, 0}, vxt[4] = {0, 0, 0, 0}, vyt[4] = {0, 0, 0, 0};
    int slot = 0;

    for (int yy = y_first; yy < ye + R; ++yy) {
        // rotate the row windows and take the prefetched row; prefetch the next one
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pt[j] = pm[j];
            pm[j] = pb[j];
            nt[j] = nm[j];
            nm[j] = nb[j];
        }
        unpack4(pf_p, pb);
        unpack4(pf_n, nb);
        pf_p = load_row(A.prev, yy + 2);
        pf_n = load_row(A.next, yy + 2);

        const int rv = (yy >= 0 && yy < A.h) ? -1 : 0;

        // ---- derivatives of row yy at this lane's 4 columns ------------------------------------------------
        int ix[4], iy[4], it[4];
        {
            // vertical parts of the separable Sobel pair (kernels.cpp:6-19): sm = [1 2 1]^T, df = [-1 0 1]^T
            int sm[6], df[6];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sm[j + 1] = pt[j] + 2 * pm[j] + pb[j];
                df[j + 1] = pb[j] - pt[j];
            }
            sm[0] = lane_from<-1>(sm[4]);
            sm[5] = lane_from<1>(sm[1]);
            df[0] = lane_from<-1>(df[4]);
            df[5] = lane_from<1>(df[1]);
            if constexpr (MODE == OFX_MODE_LK_FLOAT) {
                // It = Dt_3x3 (*) next - Dt_3x3 (*) prev (OptFlowGpu.cu:1936-1940) = Dt_3x3 (*) (next - prev), all
                // exact integers.  Dt_3x3 = [1 2 1]^T[1 2 1] - centre (kernels.cpp:20-24).
                int g[6], dm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dm[j] = nm[j] - pm[j];
                    g[j + 1] = (nt[j] - pt[j]) + 2 * dm[j] + (nb[j] - pb[j]);
                }
                g[0] = lane_from<-1>(g[4]);
                g[5] = lane_from<1>(g[1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = cv[j] & rv;
                    ix[j] = (sm[j + 2] - sm[j]) & m;
                    iy[j] = (df[j] + 2 * df[j + 1] + df[j + 2]) & m;
                    it[j] = (g[j] + 2 * g[j + 1] + g[j + 2] - dm[j]) & m;
                }
            } else {
                // cpu path: int accumulator truncated after every tap (OptFlowCPU.cpp:102) => each Gaussian tap
                // contributes floor(px * w): corner px>>4, edge px>>3, centre px>>2 (GAUS_KERNEL_3x3, kernels.cpp:61-64).
                // side[] = column contribution when the column is left/right of the centre, mid[] when it is the centre.
                int side[6], mid[4]; // bits 0..15: prev, bits 16..31: next
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sp = (pt[j] >> 4) + (pm[j] >> 3) + (pb[j] >> 4);
                    const int sn = (nt[j] >> 4) + (nm[j] >> 3) + (nb[j] >> 4);
                    const int mp = (pt[j] >> 3) + (pm[j] >> 2) + (pb[j] >> 3);
                    const int mn = (nt[j] >> 3) + (nm[j] >> 2) + (nb[j] >> 3);
                    side[j + 1] = sp | (sn << 16);
                    mid[j] = mp | (mn << 16);
                }
                side[0] = lane_from<-1>(side[4]);
                side[5] = lane_from<1>(side[1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = cv[j] & rv & 0xff; // (unsigned char) wrap, OptFlowCPU.cpp:106
                    const int gsum = side[j] + mid[j] + side[j + 2]; // both halves < 256: no carry between them
                    const int gp = gsum & 0xffff, gn = gsum >> 16;
                    ix[j] = (sm[j + 2] - sm[j]) & m;
                    iy[j] = (df[j] + 2 * df[j + 1] + df[j + 2]) & m;
                    it[j] = (gn - gp) & m; // It2 - It1 as unsigned char, OptFlowCPU.cpp:15,340
                }
            }
        }

        // ---- ring: fetch the row leaving the window (yy - NS), store the entering one ------------------------
        int ox[4], oy[4], ot[4];
        if constexpr (MODE == OFX_MODE_LK_FLOAT) {
            const uint4 ra = ring_a[slot * 64 + lane];
            const uint2 rb = ring_b[slot * 64 + lane];
            const uint32_t w4[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ox[j] = (int)(short)(w4[j] & 0xffff);
                oy[j] = (int)w4[j] >> 16;
            }
            ot[0] = (int)(short)(rb.x & 0xffff);
            ot[1] = (int)rb.x >> 16;
            ot[2] = (int)(short)(rb.y & 0xffff);
            ot[3] = (int)rb.y >> 16;
            ring_a[slot * 64 + lane] = make_uint4(((uint32_t)ix[0] & 0xffff) | ((uint32_t)iy[0] << 16),
                                                  ((uint32_t)ix[1] & 0xffff) | ((uint32_t)iy[1] << 16),
                                                  ((uint32_t)ix[2] & 0xffff) | ((uint32_t)iy[2] << 16),
                                                  ((uint32_t)ix[3] & 0xffff) | ((uint32_t)iy[3] << 16));
            ring_b[slot * 64 + lane] = make_uint2(((uint32_t)it[0] & 0xffff) | ((uint32_t)it[1] << 16),
                                                  ((uint32_t)it[2] & 0xffff) | ((uint32_t)it[3] << 16));
        } else {
            const uint4 ra = ring_a[slot * 64 + lane];
            const uint32_t w4[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ox[j] = w4[j] & 0xff;
                oy[j] = (w4[j] >> 8) & 0xff;
                ot[j] = (w4[j] >> 16) & 0xff;
            }
            ring_a[slot * 64 + lane] = make_uint4((uint32_t)ix[0] | ((uint32_t)iy[0] << 8) | ((uint32_t)it[0] << 16),
                                                  (uint32_t)ix[1] | ((uint32_t)iy[1] << 8) | ((uint32_t)it[1] << 16),
                                                  (uint32_t)ix[2] | ((uint32_t)iy[2] << 8) | ((uint32_t)it[2] << 16),
                                                  (uint32_t)ix[3] | ((uint32_t)iy[3] << 8) | ((uint32_t)it[3] << 16));
        }
        slot = (slot + 1 == NS) ? 0 : slot + 1;

        // ---- vertical running sums of the five products (OptFlowCPU.cpp:347-358 order: xx, yy, xy, xt, yt) ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vxx[j] += ix[j] * ix[j] - ox[j] * ox[j];
            vyy[j] += iy[j] * iy[j] - oy[j] * oy[j];
            vxy[j] += ix[j] * iy[j] - ox[j] * oy[j];
            vxt[j] += ix[j] * it[j] - ox[j] * ot[j];
            vyt[j] += iy[j] * it[j] - oy[j] * ot[j];
        }

        // ---- emit output row y = yy - R ----------------------------------------------------------------------
        const int y = yy - R;
        if (y >= ys) {
            int hxx[4], hyy[4], hxy[4], hxt[4], hyt[4];
            hbox4<R>(vxx, hxx);
            hbox4<R>(vyy, hyy);
            hbox4<R>(vxy, hxy);
            hbox4<R>(vxt, hxt);
            hbox4<R>(vyt, hyt);
            const bool out_lane = lane >= G::LO_LANE && lane <= G::HI_LANE && cb < A.w;
            if (out_lane) {
                const size_t pix = (size_t)(y - A.flow_row0) * (size_t)A.w + (size_t)cb;
                if constexpr (SUMS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (cb + j < A.w) {
                            A.sums[pix + j] = hxx[j];
                            A.sums[A.sums_plane + pix + j] = hyy[j];
                            A.sums[2 * A.sums_plane + pix + j] = hxy[j];
                            A.sums[3 * A.sums_plane + pix + j] = hxt[j];
                            A.sums[4 * A.sums_plane + pix + j] = hyt[j];
                        }
                    }
                } else {
                    float uv[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) solve2x2<MODE>(hxx[j], hyy[j], hxy[j], hxt[j], hyt[j], uv[2 * j], uv[2 * j + 1]);
                    float *dst = A.flow + 2 * pix;
                    if (cb + 3 < A.w) {
                        // 32 contiguous bytes per lane; the address is only 8-byte aligned in general (odd w*y)
                        float2 *d2 = reinterpret_cast<float2 *>(dst);
                        d2[0] = make_float2(uv[0], uv[1]);
                        d2[1] = make_float2(uv[2], uv[3]);
                        d2[2] = make_float2(uv[4], uv[5]);
                        d2[3] = make_float2(uv[6], uv[7]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cb + j < A.w) {
                                dst[2 * j] = uv[2 * j];
                                dst[2 * j + 1] = uv[2 * j + 1];
                            }
                    }
                }
            }
        }
    }
}

template <int R, int MODE, bool SUMS>
int launch_r(const LkArgs &base, int rows_out, hipStream_t st)
{
    using G = TileGeom<R>;
    LkArgs a = base;
    a.tiles_x = ofx_div_up(a.w, G::OUT_W);
    // enough single-wave workgroups to give every SIMD a few waves, but strips tall enough that the 2R priming
    // rows (derivatives + vertical sums only, no solve/store) stay a small fraction
    const int target_waves = 4096;
    int strips = ofx_div_up(target_waves, a.tiles_x);
    int strip_h = ofx_div_up(rows_out, strips);
    const int min_h = 2 * R > 8 ? 2 * R : 8;
    if (strip_h < min_h) strip_h = min_h;
    if (strip_h > rows_out) strip_h = rows_out;
    strips = ofx_div_up(rows_out, strip_h);
    a.strip_h = strip_h;
    const unsigned grid = (unsigned)(a.tiles_x * strips);
    hipLaunchKernelGGL((lk_level_kernel<R, MODE, SUMS>), dim3(grid), dim3(64), 0, st, a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}

template <int MODE, bool SUMS>
int launch_mode(int radius, const LkArgs &a, int rows_out, hipStream_t st)
{
    switch (radius) {
    case 1: return launch_r<1, MODE, SUMS>(a, rows_out, st);
    case 2: return launch_r<2, MODE, SUMS>(a, rows_out, st);
    case 3: return launch_r<3, MODE, SUMS>(a, rows_out, st);
    case 4: return launch_r<4, MODE, SUMS>(a, rows_out, st);
    case 5: return launch_r<5, MODE, SUMS>(a, rows_out, st);
    case 6: return launch_r<6, MODE, SUMS>(a, rows_out, st);
    case 7: return launch_r<7, MODE, SUMS>(a, rows_out, st);
    case 8: return launch_r<8, MODE, SUMS>(a, rows_out, st);
    case 9: return launch_r<9, MODE, SUMS>(a, rows_out, st);
    case 10: return launch_r<10, MODE, SUMS>(a, rows_out, st);
    case 11: return launch_r<11, MODE, SUMS>(a, rows_out, st);
    default: break;
    }
    if constexpr (MODE == OFX_MODE_COMPAT_CPU) {
        if (radius == 12) return launch_r<12, MODE, SUMS>(a, rows_out, st);
    }
    ofx_set_error("ofx_lk_level: window %d not supported in mode %d", 2 * radius + 1, MODE);
    return OFX_E_UNSUPPORTED;
}

int lk_dispatch(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode, float *d_flow,
                int32_t *d_sums, int flow_row0, void *stream)
{
    OFX_TRY(ofx_check_geom(g, "ofx_lk_level"));
    OFX_REQUIRE(d_prev && d_next && (d_flow || d_sums), "ofx_lk_level: null pointer");
    OFX_REQUIRE(window >= 3 && (window & 1), "ofx_lk_level: window must be odd and >= 3 (got %d)", window);
    OFX_REQUIRE(mode == OFX_MODE_COMPAT_CPU || mode == OFX_MODE_LK_FLOAT, "ofx_lk_level: bad mode %d", mode);
    OFX_REQUIRE(((uintptr_t)d_prev & 3) == 0 && ((uintptr_t)d_next & 3) == 0, "ofx_lk_level: planes must be 4-byte aligned");
    OFX_REQUIRE(flow_row0 <= g->out_y0, "ofx_lk_level: flow_row0 %d > out_y0 %d", flow_row0, g->out_y0);
    const int radius = window >> 1;
    OFX_TRY(ofx_check_halo(g, radius + 1, "ofx_lk_level"));
    const int rows_out = g->out_y1 - g->out_y0;
    if (rows_out <= 0) return OFX_OK;
    LkArgs a{};
    a.prev = d_prev;
    a.next = d_next;
    a.flow = d_flow;
    a.sums = d_sums;
    a.w = g->w;
    a.h = g->h;
    a.pitch = g->pitch;
    a.row0 = g->row0;
    a.row_end = g->row0 + g->rows;
    a.out_y0 = g->out_y0;
    a.out_y1 = g->out_y1;
    a.flow_row0 = flow_row0;
    hipStream_t st = ofx_stream(stream);
    if (d_sums) {
        // plane stride of the inspection output = rows from flow_row0 to out_y1
        a.sums_plane = (size_t)(g->out_y1 - flow_row0) * (size_t)g->w;
        return mode == OFX_MODE_LK_FLOAT ? launch_mode<OFX_MODE_LK_FLOAT, true>(radius, a, rows_out, st)
                                         : launch_mode<OFX_MODE_COMPAT_CPU, true>(radius, a, rows_out, st);
    }
    return mode == OFX_MODE_LK_FLOAT ? launch_mode<OFX_MODE_LK_FLOAT, false>(radius, a, rows_out, st)
                                     : launch_mode<OFX_MODE_COMPAT_CPU, false>(radius, a, rows_out, st);
}

} // namespace

extern "C" int ofx_lk_level(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode,
                            float *d_flow, int flow_row0, void *stream)
{
    OFX_REQUIRE(d_flow, "ofx_lk_level: d_flow is null");
    return lk_dispatch(d_prev, d_next, g, window, mode, d_flow, nullptr, flow_row0, stream);
}

extern "C" int ofx_lk_level_sums(const uint8_t *d_prev, const uint8_t *d_next, const ofx_geom *g, int window, int mode,
                                 int32_t *d_sums5, int flow_row0, void *stream)
{
    OFX_REQUIRE(d_sums5, "ofx_lk_level_sums: d_sums5 is null");
    return lk_dispatch(d_prev, d_next, g, window, mode, nullptr, d_sums5, flow_row0, stream);
}
