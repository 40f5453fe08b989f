// The LK march with EIGHT columns per lane (a wave row = 512 image columns): round 4.  Included by lk_body.h after lk_body_buf.h.
//
// Why.  The stream launch is bound by the vector instructions its SIMDs issue, not by memory (profiles/r03_pmc_stream.txt: the VALU
// pipes 86 % busy, traffic 1.006 x algorithmic).  tools/ubench/valu_rates.hip splits the opcodes into two classes: plain 32-bit
// add / sub / and / mov run at ~1.15 ns per wave instruction, everything else the march uses -- DPP, packed 16-bit, v_dot2, fp64,
// conversions, v_perm -- at ~1.85 ns.  Of a 4-column lane's 55 box-sum instructions per row step 46 are DPP adds (each output column
// reaches into a neighbouring lane twice: 2R + 1 = 9 columns do not fit 4); with 8 columns per lane every output column needs exactly
// ONE neighbour term -- out[j] = suffix(left lane)[j + 4] + prefix[j + 4] for j < 4, suffix[j - 4] + prefix(right lane)[j - 4] for
// j >= 4 -- and the prefix / suffix sums are plain adds: 13 cheap + 8 DPP per quantity and 8 columns instead of 2 x (3 cheap + 8 DPP).
// What else is per lane-row rather than per pixel halves per pixel as well: the six neighbour moves of the derivative stage, the row
// offsets, masks, selectors and wait states (~45 scalar instructions per 256 columns), the LDS exchange addresses.
//
// What it costs: registers.  A lane carries 2 x the row state (3 rows x 2 planes x 8 columns = 48 packed registers) and 2 x the running
// sums (40): ~170 VGPRs, three waves per SIMD instead of five -- which a kernel that is bound by its VALU does not miss (every wave
// has eight independent columns in flight where it had four).
//
// Same arithmetic, same operands, same order of the (exact, integer) sums as lk_wave_buf: the results are bit-identical
// (OFX_LK_COLS=4 / 8 selects per process; the parity tests run under both).  This form has no deep fetch (the LDS-direct loads of
// lk_body_buf.h are dword-wide): launches that choose it (levels of 16 Mpx and more) keep four columns.
#pragma once

#include <utility>

namespace ofx_dev {

template <int R, int NC>
struct TileGeomW {
    static constexpr int W = 64 * NC;                        // image columns a wave row spans
    static constexpr int LO_LANE = (R + 1 + NC - 1) / NC;    // first lane whose outputs have all their taps inside the wave
    static constexpr int HI_LANE = (W - 1 - R - NC) / NC;    // last such lane (derivatives are valid for wave columns 1 .. W - 2)
    static constexpr int OUT_W = (HI_LANE - LO_LANE + 1) * NC;
};
static_assert(TileGeomW<4, 4>::LO_LANE == TileGeom<4>::LO_LANE && TileGeomW<4, 4>::HI_LANE == TileGeom<4>::HI_LANE, "TileGeomW<R, 4> is TileGeom<R>");
static_assert(TileGeomW<11, 4>::OUT_W == TileGeom<11>::OUT_W, "TileGeomW<R, 4> is TileGeom<R>");

constexpr int kLkWaveLdsW = 64 * 64 + 256; // one row of a wide wave (64 lanes x 8 pixels x 8 bytes) + the reads past its valid end

// one image row of a lane's NC columns for both marching windows (lo halves: entering, hi: leaving): p = prev, q = next - prev
// (lk_float) or next (compat_cpu)
template <int NC>
struct RowW {
    s2 p[NC];
    s2 q[NC];
};

template <int NC>
__device__ __forceinline__ void pin_row_w(RowW<NC> &r)
{
#pragma unroll
    for (int g = 0; g < NC; g += 4) {
        asm volatile("" : "+v"(r.p[g]), "+v"(r.p[g + 1]), "+v"(r.p[g + 2]), "+v"(r.p[g + 3]), "+v"(r.q[g]), "+v"(r.q[g + 1]), "+v"(r.q[g + 2]), "+v"(r.q[g + 3]));
    }
}

// pi / ni: the entering row's dwords of prev / next (4 columns each), po / no: the leaving row's
template <int MODE, int NC>
__device__ __forceinline__ void unpack_w(const uint32_t (&pi)[NC / 4], const uint32_t (&ni)[NC / 4], const uint32_t (&po)[NC / 4],
                                         const uint32_t (&no)[NC / 4], RowW<NC> &r)
{
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        r.p[4 * g + 0] = pair_bytes<0>(pi[g], po[g]);
        r.p[4 * g + 1] = pair_bytes<1>(pi[g], po[g]);
        r.p[4 * g + 2] = pair_bytes<2>(pi[g], po[g]);
        r.p[4 * g + 3] = pair_bytes<3>(pi[g], po[g]);
        r.q[4 * g + 0] = pair_bytes<0>(ni[g], no[g]);
        r.q[4 * g + 1] = pair_bytes<1>(ni[g], no[g]);
        r.q[4 * g + 2] = pair_bytes<2>(ni[g], no[g]);
        r.q[4 * g + 3] = pair_bytes<3>(ni[g], no[g]);
    }
    if constexpr (MODE != OFX_MODE_COMPAT_CPU) {
#pragma unroll
        for (int j = 0; j < NC; ++j) r.q[j] = r.q[j] - r.p[j];
    }
}

// derivs_pk for NC columns (same operations per column; ONE set of six neighbour moves per lane-row)
template <int MODE, int NC>
__device__ __forceinline__ void derivs_w(const RowW<NC> &t, const RowW<NC> &m, const RowW<NC> &b, const s2 two, s2 (&ix)[NC], s2 (&iy)[NC],
                                         s2 (&it)[NC])
{
    if constexpr (MODE != OFX_MODE_COMPAT_CPU) {
        s2 sm[NC + 2], df[NC + 2], g[NC + 2];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            sm[j + 1] = m.p[j] * two + (t.p[j] + b.p[j]); // [1 2 1]^T (kernels.cpp:6-19)
            df[j + 1] = b.p[j] - t.p[j];                  // [-1 0 1]^T
            g[j + 1] = m.q[j] * two + (t.q[j] + b.q[j]);  // Dt_3x3 = [1 2 1]^T[1 2 1] - centre (kernels.cpp:20-24) on next - prev
        }
        sm[0] = lane_shift_s2(sm[NC], true);
        sm[NC + 1] = lane_shift_s2(sm[1], false);
        df[0] = lane_shift_s2(df[NC], true);
        df[NC + 1] = lane_shift_s2(df[1], false);
        g[0] = lane_shift_s2(g[NC], true);
        g[NC + 1] = lane_shift_s2(g[1], false);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            ix[j] = sm[j + 2] - sm[j];
            iy[j] = df[j + 1] * two + (df[j] + df[j + 2]);
            it[j] = g[j + 1] * two + (g[j] + g[j + 2]) - m.q[j];
        }
    } else {
        // cpu path: int accumulator truncated after every tap (OptFlowCPU.cpp:102) => each Gaussian tap contributes floor(px * w):
        // corner px >> 4, edge px >> 3, centre px >> 2 (GAUS_KERNEL_3x3, kernels.cpp:61-64); u8 wrap (:106, :15)
        s2 sm[NC + 2], df[NC + 2], sp[NC + 2], sn[NC + 2], mp[NC], mn[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            sm[j + 1] = (m.p[j] + m.p[j]) + t.p[j] + b.p[j];
            df[j + 1] = b.p[j] - t.p[j];
            sp[j + 1] = (t.p[j] >> 4) + (m.p[j] >> 3) + (b.p[j] >> 4);
            sn[j + 1] = (t.q[j] >> 4) + (m.q[j] >> 3) + (b.q[j] >> 4);
            mp[j] = (t.p[j] >> 3) + (m.p[j] >> 2) + (b.p[j] >> 3);
            mn[j] = (t.q[j] >> 3) + (m.q[j] >> 2) + (b.q[j] >> 3);
        }
        sm[0] = lane_shift_s2(sm[NC], true);
        sm[NC + 1] = lane_shift_s2(sm[1], false);
        df[0] = lane_shift_s2(df[NC], true);
        df[NC + 1] = lane_shift_s2(df[1], false);
        sp[0] = lane_shift_s2(sp[NC], true);
        sp[NC + 1] = lane_shift_s2(sp[1], false);
        sn[0] = lane_shift_s2(sn[NC], true);
        sn[NC + 1] = lane_shift_s2(sn[1], false);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            constexpr uint32_t m8 = 0x00ff00ffu; // the (unsigned char) wrap of OptFlowCPU.cpp:106
            ix[j] = as_s2(as_u32(sm[j + 2] - sm[j]) & m8);
            iy[j] = as_s2(as_u32((df[j + 1] + df[j + 1]) + df[j] + df[j + 2]) & m8);
            const s2 gp = sp[j] + mp[j] + sp[j + 2], gn = sn[j] + mn[j] + sn[j + 2];
            it[j] = as_s2(as_u32(gn - gp) & m8);
        }
    }
}

// accumulate_pk for NC columns: v[n][j] += P_n(entering) - P_n(leaving), n = xx, yy, xy, xt, yt (OptFlowCPU.cpp:347-358)
template <int NC>
__device__ __forceinline__ void accumulate_w(const s2 (&ix)[NC], const s2 (&iy)[NC], const s2 (&it)[NC], const uint32_t (&mm)[NC], int (&v)[5][NC])
{
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const s2 nx = ix[j] * as_s2(mm[j]), ny = iy[j] * as_s2(mm[j]);
        v[0][j] = __builtin_amdgcn_sdot2(ix[j], nx, v[0][j], false);
        v[1][j] = __builtin_amdgcn_sdot2(iy[j], ny, v[1][j], false);
        v[2][j] = __builtin_amdgcn_sdot2(ix[j], ny, v[2][j], false);
        v[3][j] = __builtin_amdgcn_sdot2(nx, it[j], v[3][j], false);
        v[4][j] = __builtin_amdgcn_sdot2(ny, it[j], v[4][j], false);
    }
}

// ---- the five horizontal box sums of a lane's NC columns, in lockstep (see hbox4x5) ---------------------------------------------
// q[n][k] = a[n][0] + .. + a[n][k] (prefix), s[n][k] = a[n][k] + .. + a[n][NC-1] (suffix).  Column I's window [I - R, I + R] is its
// part inside this lane (a prefix, a suffix, or a difference of prefixes) + a prefix of every lane it reaches to the right + a suffix of
// every lane it reaches to the left; a lane further away than the last one contributes its total (q[NC-1]).
template <int C, int NC>
__device__ __forceinline__ void column_w(const int (&a)[5][NC], int (&x)[5])
{
#pragma unroll
    for (int n = 0; n < 5; ++n) x[n] = a[n][C];
}

template <int R, int NC, int I, int D>
__device__ __forceinline__ void hboxw_right(const int (&q)[5][NC], int (&acc)[5])
{
    constexpr int hi = I + R; // last relative column of the window; lane +D holds relative columns NC*D .. NC*D + NC - 1
    if constexpr (hi < NC * D) {
        return;
    } else if constexpr (hi >= NC * D + NC - 1) {
        int x[5];
        column_w<NC - 1, NC>(q, x);
        add_from5<D>(acc, x);
        hboxw_right<R, NC, I, D + 1>(q, acc);
    } else {
        int x[5];
        column_w<hi - NC * D, NC>(q, x);
        add_from5<D>(acc, x);
    }
}

template <int R, int NC, int I, int D>
__device__ __forceinline__ void hboxw_left(const int (&s)[5][NC], int (&acc)[5])
{
    constexpr int lo = I - R; // first relative column; lane -D holds relative columns -NC*D .. -NC*D + NC - 1
    if constexpr (lo > -NC * D + NC - 1) {
        return;
    } else if constexpr (lo <= -NC * D) {
        int x[5];
        column_w<0, NC>(s, x);
        add_from5<-D>(acc, x);
        hboxw_left<R, NC, I, D + 1>(s, acc);
    } else {
        int x[5];
        column_w<lo + NC * D, NC>(s, x);
        add_from5<-D>(acc, x);
    }
}

template <int R, int NC, int I>
__device__ __forceinline__ void hboxw_one(const int (&q)[5][NC], const int (&s)[5][NC], int (&out)[5][NC])
{
    constexpr int lo = I - R, hi = I + R;
    constexpr int olo = lo > 0 ? lo : 0, ohi = hi < NC - 1 ? hi : NC - 1;
    int acc[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        if constexpr (olo == 0) acc[n] = q[n][ohi];
        else if constexpr (ohi == NC - 1) acc[n] = s[n][olo];
        else acc[n] = q[n][ohi] - q[n][olo - 1];
    }
    hboxw_right<R, NC, I, 1>(q, acc);
    hboxw_left<R, NC, I, 1>(s, acc);
#pragma unroll
    for (int n = 0; n < 5; ++n) out[n][I] = acc[n];
}

template <int R, int NC, int... I>
__device__ __forceinline__ void hboxw_all(const int (&q)[5][NC], const int (&s)[5][NC], int (&out)[5][NC], std::integer_sequence<int, I...>)
{
    (hboxw_one<R, NC, I>(q, s, out), ...);
}

template <int R, int NC>
__device__ __forceinline__ void hbox_w(const int (&a)[5][NC], int (&out)[5][NC])
{
    int q[5][NC], s[5][NC];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        q[n][0] = a[n][0];
#pragma unroll
        for (int k = 1; k < NC; ++k) q[n][k] = q[n][k - 1] + a[n][k];
        s[n][NC - 1] = a[n][NC - 1];
#pragma unroll
        for (int k = NC - 2; k >= 1; --k) s[n][k] = s[n][k + 1] + a[n][k];
        s[n][0] = q[n][NC - 1];
    }
    hboxw_all<R, NC>(q, s, out, std::make_integer_sequence<int, NC>{});
}

// One quantity at a time (the form the march uses: q and s of ONE quantity are alive at a time -- sixteen registers instead of
// eighty; the eight outputs of a quantity are independent of each other, so no DPP operation follows its producer directly).
template <int R, int NC, int I, int D>
__device__ __forceinline__ int hboxq_right(const int (&q)[NC], int acc)
{
    constexpr int hi = I + R;
    if constexpr (hi < NC * D) return acc;
    else if constexpr (hi >= NC * D + NC - 1) return hboxq_right<R, NC, I, D + 1>(q, add_from<D>(acc, q[NC - 1]));
    else return add_from<D>(acc, q[hi - NC * D]);
}

template <int R, int NC, int I, int D>
__device__ __forceinline__ int hboxq_left(const int (&s)[NC], int acc)
{
    constexpr int lo = I - R;
    if constexpr (lo > -NC * D + NC - 1) return acc;
    else if constexpr (lo <= -NC * D) return hboxq_left<R, NC, I, D + 1>(s, add_from<-D>(acc, s[0]));
    else return add_from<-D>(acc, s[lo + NC * D]);
}

template <int R, int NC, int I>
__device__ __forceinline__ int hboxq_one(const int (&q)[NC], const int (&s)[NC])
{
    constexpr int lo = I - R, hi = I + R;
    constexpr int olo = lo > 0 ? lo : 0, ohi = hi < NC - 1 ? hi : NC - 1;
    int own;
    if constexpr (olo == 0) own = q[ohi];
    else if constexpr (ohi == NC - 1) own = s[olo];
    else own = q[ohi] - q[olo - 1];
    return hboxq_left<R, NC, I, 1>(s, hboxq_right<R, NC, I, 1>(q, own));
}

template <int R, int NC, int... I>
__device__ __forceinline__ void hboxq_all(const int (&q)[NC], const int (&s)[NC], int (&out)[NC], std::integer_sequence<int, I...>)
{
    ((out[I] = hboxq_one<R, NC, I>(q, s)), ...);
}

template <int R, int NC>
__device__ __forceinline__ void hbox_q(const int (&a)[NC], int (&out)[NC])
{
    int q[NC], s[NC];
    q[0] = a[0];
#pragma unroll
    for (int k = 1; k < NC; ++k) q[k] = q[k - 1] + a[k];
    s[NC - 1] = a[NC - 1];
#pragma unroll
    for (int k = NC - 2; k >= 1; --k) s[k] = s[k + 1] + a[k];
    s[0] = q[NC - 1];
    hboxq_all<R, NC>(q, s, out, std::make_integer_sequence<int, NC>{});
}

// one pixel of the level kernel (solve_lane's body): the guard is applied by the caller's loop
template <int MODE, bool FAST>
__device__ __forceinline__ void solve_px(int sxx, int syy, int sxy, int sxt, int syt, float &u, float &v)
{
    if constexpr (FAST) {
        if (__builtin_expect(solve_fast(sxx, syy, sxy, sxt, syt, u, v) != 0ull, 0)) {
            asm volatile("" : "+v"(u), "+v"(v));
            solve_fix_singular(sxx, syy, sxy, sxt, syt, u, v);
        }
    } else {
        double a, b, d, xt, yt, det;
        solve_operands<MODE>(sxx, syy, sxy, sxt, syt, a, b, d, xt, yt, det);
        solve_tail_exact<MODE>(a, b, d, xt, yt, det, u, v);
    }
}

// the NC pixels of a lane (solve_lane)
template <int MODE, bool FAST, int NC>
__device__ __forceinline__ void solve_lane_w(const int (&h)[5][NC], const SolveOpts &opt, float (&uv)[2 * NC])
{
    if constexpr (FAST) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (__builtin_expect(solve_fast(h[0][j], h[1][j], h[2][j], h[3][j], h[4][j], uv[2 * j], uv[2 * j + 1]) != 0ull, 0)) {
                asm volatile("" : "+v"(uv[2 * j]), "+v"(uv[2 * j + 1]));
                solve_fix_singular(h[0][j], h[1][j], h[2][j], h[3][j], h[4][j], uv[2 * j], uv[2 * j + 1]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            double a, b, d, xt, yt, det;
            solve_operands<MODE>(h[0][j], h[1][j], h[2][j], h[3][j], h[4][j], a, b, d, xt, yt, det);
            solve_tail_exact<MODE>(a, b, d, xt, yt, det, uv[2 * j], uv[2 * j + 1]);
        }
    }
    if (__builtin_expect(opt.min_det > 0.0f, 0)) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            asm volatile("" : "+v"(uv[2 * j]), "+v"(uv[2 * j + 1]));
            solve_guard<MODE>(h[0][j], h[1][j], h[2][j], opt.min_det, uv[2 * j], uv[2 * j + 1]);
        }
    }
}

// One wave of the fused level kernel, NC = 8 columns per lane.  ITER as in lk_wave_buf (0: flow = result; 1: flow += result; 2: the same
// and the warped image of the next iteration; 3: flow = result and the warped image of iteration 2; 4 / 5: 2 / 3 on a shard's row window).
template <int R, int MODE, bool FAST, bool INTERIOR, int ITER = 0, int NC = 8>
__device__ __forceinline__ void lk_wave_wide(const LkTable &T, int wave, int lane, uint8_t *xlds)
{
    static_assert(NC == 8, "the loads, the exchange and the stores below are written for eight columns per lane");
    using G = TileGeomW<R, NC>;
    constexpr int NS = 2 * R + 1, NG = NC / 4;
    constexpr bool ACC = ITER == 1 || ITER == 2 || ITER == 4, WOUT = ITER >= 2, ROWWIN = ITER >= 4;

    if (wave >= T.first_block[T.n]) return;
    int level = 0, hi = T.n;
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (wave >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    LkArgs A = T.lv[level];
    pin_scalar(A.w);
    pin_scalar(A.h);
    pin_scalar(A.pitch);
    pin_scalar(A.row0);
    pin_scalar(A.row_end);
    pin_scalar(A.flow_row0);
    pin_scalar(A.min_det);
    const SolveOpts sopt{A.min_det};
    const int guard_on = __builtin_amdgcn_readfirstlane(A.min_det > 0.0f ? 1 : 0); // (a scalar flag: the test per pixel is one s_cmp, not a vector compare)
    const int block = wave - T.first_block[level];
    const int tile = block % A.tiles_x;
    const int strip = block / A.tiles_x;
    const int cb = tile * G::OUT_W - G::LO_LANE * NC + NC * lane; // first of this lane's NC image columns (a multiple of 8)
    const int ys = A.out_y0 + strip * A.strip_h;
    const int ye = min(ys + A.strip_h, A.out_y1);

    const int plane_bytes = (A.row_end - A.row0) * A.pitch;
    const __amdgpu_buffer_rsrc_t rs_prev = make_rsrc(A.prev, plane_bytes), rs_next = make_rsrc(A.next, plane_bytes);
    const __amdgpu_buffer_rsrc_t rs_flow = make_rsrc(A.flow, (A.out_y1 - A.flow_row0) * A.w * 8);

    // column validity: bytes outside [0,w) read as zero, derivatives there are zero.  (The pitch is a multiple of 64 and cb of 8: a
    // lane whose first column lies inside the image can load all 8 bytes; the ones past w are masked.)
    const bool ld_ok = INTERIOR || (cb >= 0 && cb < A.w);
    uint32_t bmask[NG];
    int cm[NC];
#pragma unroll
    for (int g = 0; g < NG; ++g) bmask[g] = INTERIOR ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < NC; ++j) cm[j] = -1;
    if constexpr (!INTERIOR) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const bool in = (cb + j) >= 0 && (cb + j) < A.w;
            cm[j] = in ? -1 : 0;
            bmask[j / 4] |= in ? (0xffu << (8 * (j % 4))) : 0u;
        }
    }
    uint32_t col_off = ld_ok ? (uint32_t)cb : 0u;
    // the exchanged layout of an output row: lane l stores the 16-byte chunks l, l + 64, l + 128, l + 192 of the tile's row
    const int x0 = tile * G::OUT_W;
    const int nv = min(x0 + G::OUT_W, A.w) - x0;
    uint32_t l16 = 16u * (uint32_t)lane;
    const int lim = 8 * nv - 16, c16 = 16 * lane;
    uint32_t vo[4];
    bool st2[4];
    bool ragged_l = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool st4 = c16 + 1024 * i <= lim;
        st2[i] = c16 + 1024 * i == lim + 8; // a level of odd width ends inside a chunk: that lane stores one pixel
        vo[i] = st4 ? l16 + 1024u * (uint32_t)i : (uint32_t)kOob;
        ragged_l = ragged_l || st2[i];
    }
    const bool ragged = __any(ragged_l) != 0;
    const lds_ptr xl_w = (lds_ptr)xlds + 64 * lane;
    const lds_ptr xl_base = (lds_ptr)xlds + 64 * G::LO_LANE;
    [[maybe_unused]] uint32_t nat_off = (uint32_t)cb * 8u; // ACC: byte offset of this lane's first pixel in a flow row
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_wsrc = rs_prev, rs_wout = rs_prev;
    [[maybe_unused]] uint32_t wvo[NG] = {(uint32_t)kOob, (uint32_t)kOob}, wmiss = 0u;
    [[maybe_unused]] int wnpx[NG] = {0, 0};
    [[maybe_unused]] WarpRowState WM[NG];
    if constexpr (WOUT) {
        pin_scalar(A.warp_scale);
        rs_wsrc = make_rsrc(A.warp_src, plane_bytes);
        rs_wout = make_rsrc(A.warp_out, plane_bytes);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const bool out_lane = lane >= G::LO_LANE && lane <= G::HI_LANE && cb + 4 * g < A.w;
            wvo[g] = out_lane ? (uint32_t)(cb + 4 * g) : (uint32_t)kOob;
            wnpx[g] = out_lane ? min(4, A.w - cb - 4 * g) : 0;
            warp_row_clear(WM[g]);
        }
    }

    const int y_lim = min(min(ye + R + 1, A.h), A.row_end);
    const int y_min = max(0, A.row0);
    const int y_first = ys - R; // first derivative row this strip needs
    const int span_in = max(y_lim - y_min, 0);
    const int y_min_out = max(y_min, y_first - 1), span_out = max(y_lim - y_min_out, 0);
    auto row_off = [&](int y) -> int { return (uint32_t)(y - y_min) < (uint32_t)span_in ? (y - A.row0) * A.pitch : kOob; };
    auto row_off_out = [&](int y) -> int { return (uint32_t)(y - y_min_out) < (uint32_t)span_out ? (y - A.row0) * A.pitch : kOob; };

    // ---- fused shift (lk_wave_impl, "fused shift"): the column map (int)((float)x + u) is monotone with steps of 0 or 1, so the
    // in-image targets of a lane's 8 columns lie in 8 consecutive source bytes.  Per lane, once: the base column nb of those bytes and,
    // per output dword, a v_perm_b32 selector into the 8-byte window (sel), the selector that repairs the pixels whose column target is
    // outside the image with their own byte (fix), and the one that takes every pixel's own byte (own_sel: row target outside).
    const float su = A.uv ? A.uv[0] : 0.0f, sv = A.uv ? A.uv[1] : 0.0f;
    uint32_t nb_off = 0u, sel[NG], fix[NG], own_sel[NG];
    bool all_in;
    {
        int n[NC], nb = 0x7fffffff;
        bool in[NC], lane_all_in = true;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int x = cb + j;
            const float tx = (float)x + su;
            in[j] = x >= 0 && x < A.w && tx > -1.0f && tx < (float)A.w;
            n[j] = in[j] ? (int)tx : 0;
            if (in[j]) nb = min(nb, n[j]);
        }
        nb = max(0, min(nb == 0x7fffffff ? 0 : nb, A.pitch - NC)); // the window stays inside the row pitch
#pragma unroll
        for (int g = 0; g < NG; ++g) sel[g] = fix[g] = own_sel[g] = 0u;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int x = cb + j, g = j / 4, sh8 = 8 * (j % 4);
            const bool inside = x >= 0 && x < A.w;
            // sel: 0..7 = byte of the window (v_perm(hi, lo, sel)), 0x0c = 0x00
            sel[g] |= (!inside ? 0x0cu : (in[j] ? (uint32_t)(n[j] - nb) : 0x0cu)) << sh8;
            // fix / own_sel on v_perm(own dword, shifted result, .): 0..3 = byte of the shifted result, 4..7 = byte of the own dword
            fix[g] |= (!inside ? 0x0cu : (in[j] ? (uint32_t)(j % 4) : (uint32_t)(4 + j % 4))) << sh8;
            own_sel[g] |= (!inside ? 0x0cu : (uint32_t)(4 + j % 4)) << sh8;
            if (inside && !in[j]) lane_all_in = false;
        }
        nb_off = (uint32_t)nb;
        all_in = __all(lane_all_in) != 0;
    }
    const int y_none = (A.h + 2) / 3, y_part = (A.h % 3) ? A.h / 3 : -1;
    int map_base = 0, row_tab = kOob;
    auto refresh_map = [&](int y0) {
        map_base = y0;
        const int y = y0 + lane;
        const float ty = (float)y + sv;
        const bool yin = ty > -1.0f && ty < (float)A.h;
        const int ny = yin ? (int)ty : 0;
        const bool ok = y >= y_min && y < y_lim && yin && ny >= A.row0 && ny < A.row_end;
        row_tab = ok ? (ny - A.row0) * A.pitch : kOob;
    };
    struct NextRaw {
        u32x2 own, sh; // the row's own 8 bytes / the 8 bytes at the shifted position
        int miss;      // wave-uniform: all ones when the shifted row does not exist
    };
    auto fetch_next = [&](int y, int po) -> NextRaw {
        NextRaw r;
        const int e = __builtin_amdgcn_readlane(row_tab, y - map_base);
        r.sh = __builtin_amdgcn_raw_buffer_load_b64(rs_next, nb_off, e, 0);
        r.miss = e >> 31;
        asm("" : "=v"(r.own)); // (never used while every byte comes from the shifted window)
        if (__builtin_expect(!all_in || r.miss != 0, 0)) {
            r.own = __builtin_amdgcn_raw_buffer_load_b64(rs_next, col_off, y < y_none ? po : kOob, 0);
            if (y == y_part) { // pixels past w * h / 3 of the row in between keep nothing (OptFlowCPU.cpp:247)
                uint32_t km[NG] = {0u, 0u};
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    if (3ll * (cb + j) < (long long)A.w * (A.h % 3)) km[j / 4] |= 0xffu << (8 * (j % 4));
                r.own.x &= km[0];
                r.own.y &= km[1];
            }
        }
        return r;
    };
    auto finish_next = [&](const NextRaw &r, uint32_t (&out)[NG]) {
        out[0] = __builtin_amdgcn_perm(r.sh.y, r.sh.x, sel[0]);
        out[1] = __builtin_amdgcn_perm(r.sh.y, r.sh.x, sel[1]);
        if (__builtin_expect(!all_in || r.miss != 0, 0)) { // (wave-uniform) some pixels keep their own byte
            uint32_t s0, s1;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(s0) : "s"(r.miss), "v"(own_sel[0]), "v"(fix[0]));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(s1) : "s"(r.miss), "v"(own_sel[1]), "v"(fix[1]));
            out[0] = __builtin_amdgcn_perm(r.own.x, out[0], s0);
            out[1] = __builtin_amdgcn_perm(r.own.y, out[1], s1);
        }
    };
    auto fetch_prev = [&](int po) -> u32x2 { return __builtin_amdgcn_raw_buffer_load_b64(rs_prev, col_off, po, 0); };
    auto finish_row = [&](const u32x2 raw, uint32_t (&out)[NG]) {
        out[0] = INTERIOR ? raw.x : (raw.x & bmask[0]);
        out[1] = INTERIOR ? raw.y : (raw.y & bmask[1]);
    };
    auto load_pair = [&](int y, bool out, uint32_t (&p)[NG], uint32_t (&n)[NG]) {
        const int po = out ? row_off_out(y) : row_off(y);
        finish_row(fetch_prev(po), p);
        finish_next(fetch_next(y, po), n);
    };

    constexpr int H = OFX_LK_FOLD_PRIMING ? R - 1 : 0;
    constexpr int PR = 2 * R - H;
    const int y_lo0 = y_first + H;
    const int nsteps = (ye - ys) + PR;
    RowW<NC> wp[3];
    const s2 two = pk_two();
    refresh_map(y_first - 1);
    {
        uint32_t pi[NG], ni[NG], po[NG] = {0u, 0u}, no[NG] = {0u, 0u};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            load_pair(y_lo0 - 1 + t, false, pi, ni);
            if constexpr (H > 0) load_pair(y_first - 1 + t, true, po, no);
            unpack_w<MODE, NC>(pi, ni, po, no, wp[t]);
        }
    }
    int v[5][NC];
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int j = 0; j < NC; ++j) v[n][j] = 0;
    int fso0 = __builtin_amdgcn_readfirstlane(((ys - A.flow_row0) * A.w + x0) * 8);
    int fstep = A.w * 8;
    pin_scalar(fso0);
    pin_scalar(fstep);

    auto body = [&](auto K, int s) {
        constexpr int k = decltype(K)::value; // s mod 3
        const int yy = y_lo0 + s;             // derivative row entering the window (low halves)
        const int yo = yy - NS;               // derivative row leaving it (high halves, once the folded priming is over)
        const bool folded = H > 0 && s < H;   // high halves: the entering row y_first + s
        const int yh = folded ? y_first + s : yo;
        const int ro = (H > 1 && s + 1 < H) ? y_first + s + 2 : yo + 2; // b row of the high stream's next step

        // the loads of the rows the next step adds, finished at the end of this step, before its stores
        if (yy + 2 - map_base >= 64) refresh_map(yo + 2);
        const int po_in = row_off(yy + 2), po_out = row_off_out(ro);
        const u32x2 pf_ip = fetch_prev(po_in), pf_op = fetch_prev(po_out);
        const NextRaw pf_in = fetch_next(yy + 2, po_in);
        NextRaw pf_on = {u32x2{0u, 0u}, u32x2{0u, 0u}, -1};
        if (ro >= y_first - 1) pf_on = fetch_next(ro, po_out);
        const bool emit = s >= PR;
        [[maybe_unused]] f32x4 old[4];
        if constexpr (ACC) {
            asm("" : "=v"(old[0]), "=v"(old[1]), "=v"(old[2]), "=v"(old[3]));
            if (emit) {
                const int fnat = __builtin_amdgcn_readfirstlane(fso0 + (s - PR) * fstep - x0 * 8); // offset of the row's pixel 0
                if constexpr (INTERIOR) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) old[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_flow, nat_off + 16u * (uint32_t)i, fnat, 0));
                } else { // (per pixel: a level of odd width ends inside a 16-byte piece)
                    auto off = [&](int j) { return cm[j] ? nat_off + 8u * (uint32_t)j : (uint32_t)kOob; };
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs_flow, off(2 * i), fnat, 0), b = __builtin_amdgcn_raw_buffer_load_b64(rs_flow, off(2 * i + 1), fnat, 0);
                        old[i] = __builtin_bit_cast(f32x4, u32x4{a.x, a.y, b.x, b.y});
                    }
                }
            }
        }

        const uint32_t him = folded ? 0x00010000u : (yo >= y_first ? 0xffff0000u : 0u);
        uint32_t rowm = ((uint32_t)yy < (uint32_t)A.h ? 0x00000001u : 0u) | ((uint32_t)yh < (uint32_t)A.h ? him : 0u);
        if constexpr (INTERIOR) rowm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowm);
        uint32_t mm[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) mm[j] = (uint32_t)cm[j] & rowm;
        {
            s2 ix[NC], iy[NC], it[NC];
            derivs_w<MODE, NC>(wp[k], wp[(k + 1) % 3], wp[(k + 2) % 3], two, ix, iy, it);
            accumulate_w<NC>(ix, iy, it, mm, v);
        }
        auto take_rows = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            uint32_t a_p[NG], a_n[NG], a_po[NG], a_no[NG];
            finish_row(pf_ip, a_p);
            finish_next(pf_in, a_n);
            finish_row(pf_op, a_po);
            finish_next(pf_on, a_no);
            unpack_w<MODE, NC>(a_p, a_n, a_po, a_no, wp[k]);
            pin_row_w<NC>(wp[k]);
        };

        if (emit) {
            // the box sums quantity by quantity, then the solve pixel by pixel, every pixel's (u, v) into the exchange row as soon as
            // it exists: what is alive at a time is one quantity's prefix / suffix sums, the forty sums and one pixel's doubles
            int hb[5][NC];
#pragma unroll
            for (int n = 0; n < 5; ++n) hbox_q<R, NC>(v[n], hb[n]);
            [[maybe_unused]] float fu[NC], fv[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                float pu, pv;
                solve_px<MODE, FAST>(hb[0][j], hb[1][j], hb[2][j], hb[3][j], hb[4][j], pu, pv);
                if (__builtin_expect(guard_on != 0, 0)) { // (wave-uniform; a real branch: see solve2x2)
                    asm volatile("" : "+v"(pu), "+v"(pv));
                    solve_guard<MODE>(hb[0][j], hb[1][j], hb[2][j], sopt.min_det, pu, pv);
                }
                if constexpr (ACC) { // (the old flow is zero in the columns outside the image, which are never stored)
                    const f32x4 o = old[j / 2];
                    pu = ((j & 1) ? o.z : o.x) + pu;
                    pv = ((j & 1) ? o.w : o.y) + pv;
                }
                if constexpr (WOUT) fu[j] = pu, fv[j] = pv;
                *(__attribute__((address_space(3))) f32x2 *)(xl_w + 8 * j) = f32x2{pu, pv};
            }
            if constexpr (WOUT) {
                // the warped row of the step before: second stage and store; then this row's first stage from the flow just formed
                const int yw = yy - R; // this step's output row
                const int wso = __builtin_amdgcn_readfirstlane(s > PR ? (yw - 1 - A.row0) * A.pitch : kOob);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const uint32_t wn = warp_row_finish(WM[g]);
                    __builtin_amdgcn_raw_buffer_store_b32(wn, rs_wout, wvo[g], wso, 0);
                    const float gu[4] = {fu[4 * g], fu[4 * g + 1], fu[4 * g + 2], fu[4 * g + 3]}, gv[4] = {fv[4 * g], fv[4 * g + 1], fv[4 * g + 2], fv[4 * g + 3]};
                    uint32_t miss_g = 0u;
                    warp_row_prepare<ROWWIN>(rs_wsrc, A.warp_scale, A.w, A.h, A.pitch, A.row0, A.row_end, cb + 4 * g, yw, wnpx[g], gu, gv, WM[g], miss_g);
                    wmiss |= miss_g;
                }
            }
        }
        take_rows();
        if (emit) {
            // the row comes back out of LDS in the exchanged layout only now, after the rows have been taken: sixteen registers that
            // are not alive across the unpacking (a wave's LDS operations execute in order: no barrier between its writes and reads)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const lds_ptr xl_r = xl_base + lane_off_var(l16);
            f32x4 xo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xo[i] = *(__attribute__((address_space(3))) f32x4 *)(xl_r + 1024 * i);
            const int fso = __builtin_amdgcn_readfirstlane(fso0 + (s - PR) * fstep); // this row's offset in the flow
            // four gap-free streaming stores of 1 KB; the lanes past the tile's end are dropped by the resource's range check
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xo[i]), rs_flow, vo[i], fso, OFX_LK_STORE_AUX);
            if (__builtin_expect(ragged, 0)) { // the one lane whose chunk holds a single pixel
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 qv = __builtin_bit_cast(u32x4, xo[i]);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{qv.x, qv.y}, rs_flow, st2[i] ? l16 + 1024u * (uint32_t)i : (uint32_t)kOob, fso, OFX_LK_STORE_AUX);
                }
            }
        }
    };

#if OFX_LK_PROGRESS_PRIORITY
    const int q1 = nsteps / 4, q2 = nsteps / 2, q3 = nsteps - nsteps / 4;
    __builtin_amdgcn_s_setprio(3);
#define OFX_LK_PRIO_STEP()                               \
    do {                                                 \
        if (s >= q3) __builtin_amdgcn_s_setprio(0);      \
        else if (s >= q2) __builtin_amdgcn_s_setprio(1); \
        else if (s >= q1) __builtin_amdgcn_s_setprio(2); \
    } while (0)
#else
#define OFX_LK_PRIO_STEP() ((void)0)
#endif
    int s = 0;
    while (true) {
        body(std::integral_constant<int, 0>{}, s);
        if (++s >= nsteps) break;
        body(std::integral_constant<int, 1>{}, s);
        if (++s >= nsteps) break;
        body(std::integral_constant<int, 2>{}, s);
        if (++s >= nsteps) break;
        OFX_LK_PRIO_STEP();
    }
#undef OFX_LK_PRIO_STEP
    if constexpr (WOUT) { // the warped row of the last step
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t wn = warp_row_finish(WM[g]);
            __builtin_amdgcn_raw_buffer_store_b32(wn, rs_wout, wvo[g], (ye - 1 - A.row0) * A.pitch, 0);
        }
        if constexpr (ROWWIN) {
            if (__any(wmiss != 0u) && A.warp_status != nullptr && lane == 0) atomicOr(A.warp_status, 1 << A.warp_status_bit);
        }
    }
}

// the wave's variant for its tile (wave-uniform): interior tiles have compile-time column masks
template <int R, int MODE, bool FAST, int ITER = 0, int NC = 8>
__device__ __forceinline__ void lk_wave_w(const LkTable &T, int wave, int lane, uint8_t *xlds)
{
    if (wave >= T.first_block[T.n]) return;
    int level = 0, hi = T.n;
    while (hi - level > 1) {
        const int mid = (level + hi) >> 1;
        if (wave >= T.first_block[mid]) level = mid;
        else hi = mid;
    }
    const int tile = (wave - T.first_block[level]) % T.lv[level].tiles_x;
    const int cb0 = tile * TileGeomW<R, NC>::OUT_W - TileGeomW<R, NC>::LO_LANE * NC;
    if (cb0 >= 0 && cb0 + TileGeomW<R, NC>::W <= T.lv[level].w) lk_wave_wide<R, MODE, FAST, true, ITER, NC>(T, wave, lane, xlds);
    else lk_wave_wide<R, MODE, FAST, false, ITER, NC>(T, wave, lane, xlds);
}

} // namespace ofx_dev
