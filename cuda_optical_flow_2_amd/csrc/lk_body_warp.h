// The bilinear warp of lk_iter in the form the accumulating march uses (lk_body_buf.h, ITER == 2): one
// lane's 4 adjacent pixels of one row, in two stages with the loads of the taps in flight between them.  Included by lk_body.h.
//
// A refinement iteration used to be two launches: warp_u8_kernel (flow -> warped image: a flow load, then the tap loads that
// depend on it, per short wave -- latency-bound at 0.37 of the HBM roofline) and the accumulating level kernel.  The march of
// iteration j now also writes the warped image iteration j + 1 reads: the flow of an output row is in registers right after
// its solve, so the row's warp needs no flow load at all, and its tap loads have a whole step of the march to arrive.  Only the
// first refinement iteration of a pair still needs warp_u8_kernel.
//
// Per pixel the four taps are two (generally unaligned) dwords, one from each of the rows yi and y1, at byte xi of the row --
// pulled back to the row's last dword where xi lies beyond it -- loaded through a buffer resource, so that no coordinate, however
// wild, reads outside the level.  Unlike warp_u8_kernel's 3 x 8-byte window per lane this form has no condition to qualify for
// and therefore no second, general form: eight loads per lane instead of six, fewer instructions, no call.
// The arithmetic per pixel is the oracle's (orc_warp_bilinear_u8), in the operation order of warp4_general (stages_body.h):
// the bytes are those of warp_u8_kernel.  A pixel whose flow is not finite is not warped (stages_body.h): here, a pixel with
// zero flow -- both fractions are 0 then, and p + 0 * (q - p) is p for all bytes p, q.
#pragma once

namespace ofx_dev {

// OFX_WARP_PACK_SEL: the four pixels' byte selectors in ONE register between the stages (4 bits per pixel: xi - xb and x1 - xb, each
// 0 .. 3) instead of four -- three VGPRs for a kernel that sits on the 128-register line (the LDS ring of lk_body_buf.h needs them),
// ~16 more vector instructions per row step to pack and unpack.
#ifndef OFX_WARP_PACK_SEL
#define OFX_WARP_PACK_SEL 0 // (on only together with OFX_LK_OUT_RING=1)
#endif
struct WarpRowState {      // a row of a lane between the two stages
    float fx[4], fy[4];    // the fractions of the source coordinates
#if OFX_WARP_LEAN
#elif OFX_WARP_PACK_SEL
    uint32_t selp;         // bits 4k .. 4k+1: xi - xb of pixel k, bits 4k+2 .. 4k+3: x1 - xb
#else
    uint32_t sel[4];       // per pixel the byte selector (xi - xb, x1 - xb, zero, zero) into its two dwords (general rows only)
#endif
    uint32_t ra[4], rb[4]; // per pixel the dwords of rows yi and y1 (loads in flight between the stages)
    int general;           // wave-uniform: some pixel of the wave's row has its taps at the right end of a source row (see below)
};

// Round 4: the common row.  The selector only differs from (byte 0, byte 1) where a tap column reaches the last three bytes of the
// row pitch or the image's last column -- for every other pixel the dword fetched AT byte xi holds p(xi) in byte 0 and p(xi + 1) in
// byte 1.  One test per lane-row (the largest of the four xi against min(pitch - 4, w - 2)) and a wave-uniform branch replace the
// seven selector instructions per pixel in stage 1 and the two v_perm per pixel in stage 2; v_cvt_f32_ubyte0 / 1 read the bytes where
// they lie.  Same bytes (the iteration tests pass with either form).  MEASURED SLOWER and therefore OFF: 4K, 5 iterations, the
// accumulating launch that also warps takes 498-517 us with it against 457-461 us without (profiles/r04_ablation.txt, batch 4) --
// 38 fewer vector instructions per row step, but the tap loads now sit in two arms of a branch, and at the join hipcc's wait-count
// insertion no longer lets them stay in flight across the step.  OFX_WARP_FAST_ROWS=1 builds it.
#ifndef OFX_WARP_FAST_ROWS
#define OFX_WARP_FAST_ROWS 0
#endif
// Round 4, second session: NO row is general.  The dword fetched AT byte xi of a tap row holds p(xi) in byte 0 and p(xi + 1) in byte 1
// whenever p(xi + 1) is looked at: (a) the right tap is only replaced by the pixel itself (replicate border) for xi = w - 1, and there
// the source column was clamped to w - 1, its fraction is 0 and p + 0 * (q - p) is p for every byte q -- whatever byte 1 holds;
// (b) a dword that starts in the last three bytes of the row pitch runs into the next row, but its bytes 0 and 1 are still this
// row's (xi <= w - 1 < pitch, and xi + 1 <= w - 1 when it counts).  What is left is the dword that starts in the last three bytes of the
// LAST row of the plane: the resource is declared three bytes longer than the rows (lk_body_buf.h), which is why d_warp_src must be
// followed by three readable bytes (include/ofx.h; every plane of a session is followed by 64).  No selector, no pull-back, no
// permute: ~40 vector instructions per row step less than the general form (of ~500) and four registers (sel[]) -- and, unlike
// OFX_WARP_FAST_ROWS, no branch.  Same bytes (the iteration tests compare against the warp launch and the oracle).  0 = the general form.
#ifndef OFX_WARP_LEAN
#define OFX_WARP_LEAN 1
#endif

__device__ __forceinline__ void warp_row_clear(WarpRowState &M)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) M.fx[k] = M.fy[k] = 0.0f, M.ra[k] = M.rb[k] = 0u;
#if OFX_WARP_LEAN
#elif OFX_WARP_PACK_SEL
    M.selp = 0u; // (both taps byte 0 of a zero dword: the pending row of the first emitting step is stored nowhere)
#else
    for (int k = 0; k < 4; ++k) M.sel[k] = 0x0c0c0c0cu;
#endif
    M.general = 1;
}

// Stage 1: source coordinates and selectors; issues the eight tap loads through `rs` (the rows [row0, row_end) of the warp source,
// `pitch` bytes apart, pitch >= 4: the whole level, or -- ROWWIN -- the row window a shard holds).
// ROWWIN: a tap row outside the window is replaced by the window's nearest row, and `miss` gets bit k set for a wanted pixel k
// (k < npx) with a finite flow whose taps needed such a row: the caller reports it (ofx_session_corner_status, bits 16 + level).
template <bool ROWWIN>
__device__ __forceinline__ void warp_row_prepare(const __amdgpu_buffer_rsrc_t &rs, float scale, int w, int h, int pitch, int row0, int row_end, int x0,
                                                 int y, int npx, const float (&fu)[4], const float (&fv)[4], WarpRowState &M, uint32_t &miss)
{
    const float xf0 = (float)x0, yf = (float)y, wmaxf = (float)(w - 1), hmaxf = (float)(h - 1);
    const int wmax = w - 1, hmax = h - 1;
    int xi[4], ya[4], yb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xr = xf0 + (float)k;
        const float sxr = xr + scale * fu[k], syr = yf + scale * fv[k];
        const bool ok = __builtin_fabsf(sxr) <= 1e9f && __builtin_fabsf(syr) <= 1e9f; // (NaN fails)
        const float sx = __builtin_amdgcn_fmed3f(ok ? sxr : xr, 0.0f, wmaxf);
        const float sy = __builtin_amdgcn_fmed3f(ok ? syr : yf, 0.0f, hmaxf);
        xi[k] = (int)sx;
        const int yi = (int)sy;
        M.fx[k] = __builtin_amdgcn_fractf(sx); // == sx - (float)xi: sx >= 0, the difference is exact
        M.fy[k] = __builtin_amdgcn_fractf(sy);
        ya[k] = yi, yb[k] = min(yi + 1, hmax);
        if constexpr (ROWWIN) {
            if (k < npx && ok && (ya[k] < row0 || yb[k] >= row_end)) miss |= 1u << k;
            ya[k] = min(max(ya[k], row0), row_end - 1) - row0;
            yb[k] = min(max(yb[k], row0), row_end - 1) - row0;
        }
    }
#if OFX_WARP_LEAN
    M.general = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#ifdef OFX_X_NO_TAPS // (diagnostic build: no tap loads)
        M.ra[k] = (uint32_t)(ya[k] * pitch + xi[k]), M.rb[k] = (uint32_t)(yb[k] * pitch + xi[k]);
#else
        M.ra[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(ya[k] * pitch + xi[k]), 0, 0);
        M.rb[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(yb[k] * pitch + xi[k]), 0, 0);
#endif
    }
    (void)wmax;
    return;
#else
    // the common row: every tap dword can be fetched at byte xi itself and holds p(xi), p(xi + 1) in its bytes 0 and 1
    const int xlim = min(pitch - 4, w - 2);
    M.general = !OFX_WARP_FAST_ROWS || __any(max(max(xi[0], xi[1]), max(xi[2], xi[3])) > xlim) != 0;
    if (__builtin_expect(M.general, 0)) {
#if OFX_WARP_PACK_SEL
        M.selp = 0u;
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xb = min(xi[k], pitch - 4); // the dword stays inside the row pitch
#ifdef OFX_X_NO_TAPS // (diagnostic build: no tap loads)
            M.ra[k] = (uint32_t)(ya[k] * pitch + xb), M.rb[k] = (uint32_t)(yb[k] * pitch + xb);
#else
            M.ra[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(ya[k] * pitch + xb), 0, 0);
            M.rb[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(yb[k] * pitch + xb), 0, 0);
#endif
#if OFX_WARP_PACK_SEL
            M.selp |= ((uint32_t)(xi[k] - xb) | ((uint32_t)(min(xi[k] + 1, wmax) - xb) << 2)) << (4 * k);
#else
            M.sel[k] = (uint32_t)(xi[k] - xb) | ((uint32_t)(min(xi[k] + 1, wmax) - xb) << 8) | 0x0c0c0000u;
#endif
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            M.ra[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(ya[k] * pitch + xi[k]), 0, 0);
            M.rb[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(yb[k] * pitch + xi[k]), 0, 0);
        }
    }
#endif
}

// Stage 2: the taps have arrived; the row's four bytes.
__device__ __forceinline__ uint32_t warp_row_finish(const WarpRowState &M)
{
    uint32_t out = 0;
    auto blend = [&](int k, uint32_t pa, uint32_t pb) {
        const float p00 = (float)(pa & 0xffu), p01 = (float)((pa >> 8) & 0xffu); // (v_cvt_f32_ubyte0 / 1)
        const float p10 = (float)(pb & 0xffu), p11 = (float)((pb >> 8) & 0xffu);
        const float a = p00 + M.fx[k] * (p01 - p00);
        const float b = p10 + M.fx[k] * (p11 - p10);
        const float v = a + M.fy[k] * (b - a);
        out |= ((uint32_t)(int)(v + 0.5f) & 0xffu) << (8 * k);
    };
#if OFX_WARP_LEAN
#pragma unroll
    for (int k = 0; k < 4; ++k) blend(k, M.ra[k], M.rb[k]);
    return out;
#else
    if (__builtin_expect(M.general, 0)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#if OFX_WARP_PACK_SEL
            const uint32_t sk = ((M.selp >> (4 * k)) & 3u) | (((M.selp >> (4 * k + 2)) & 3u) << 8) | 0x0c0c0000u;
#else
            const uint32_t sk = M.sel[k];
#endif
            blend(k, __builtin_amdgcn_perm(0u, M.ra[k], sk), __builtin_amdgcn_perm(0u, M.rb[k], sk));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) blend(k, M.ra[k], M.rb[k]);
    }
    return out;
#endif
}

} // namespace ofx_dev
