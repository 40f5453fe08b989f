// The pair-at-a-time path's pyramid launch with the corner chain of the pair aboard (round 4).
//
// ofx_session_build_pyramid -> ofx_session_corner_flows -> the LK launch were three launches per pair, the middle one a single
// wave walking a dependent chain for 12 us while the chip idles.  The chain needs levels >= 1 of the NEXT frame's pyramid only
// around the top-left corner, and the pyramid of a top-left patch is the top-left part of the frame's pyramid (corner_body.h,
// PatchBuild) -- so one more block of the pyramid launch builds that small pyramid for itself from level 0 (which is complete
// before the launch) and walks the chain on it, beside the tiles of the real pyramid and without any hand-over between blocks.
// The previous frame's patch planes are the ones the previous pair's block built as ITS next frame (first = 1), unless the
// session says they are not (first = 0: both).  A shift that leaves the patch is repaired inside the block from the whole
// level 0 (corner_block, patch_build_reloc), so the vectors are the reference's for every input, as with the stand-alone
// corner kernel that reads whole planes.
#include "corner_body.h"
#include "stages_body.h"

using namespace ofx_dev;

namespace {

struct PyrCornerArgs {
    PyrArgs pyr;
    int gx; // tiles per row of the pyramid's grid; block 0 is the chain, block 1 + i tile i
    CornerHead hd;
    CornerLevel lv[OFX_MAX_LEVELS];
    PatchBuild patch;
    PatchBuildSlot slot;
};

template <int MODE, bool FAST>
__global__ __launch_bounds__(kPyrThreads) void pyramid_corner_kernel(const PyrCornerArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = (int)threadIdx.x;
    if (blockIdx.x == 0) {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        __builtin_amdgcn_s_setprio(3); // the latency-bound chain goes first wherever it shares a SIMD
        patch_build_block(A.patch, A.slot, tid); // (all 256 threads; ends with a barrier)
        corner_block<MODE, FAST>(A.hd, A.lv, A.patch, tid, wv, reinterpret_cast<float *>(lds), lds + kCornerScratch,
                                 reinterpret_cast<int *>(lds + kCornerScratch - 32));
        return;
    }
    const int b = (int)blockIdx.x - 1;
    pyramid_block(A.pyr, b % A.gx, b / A.gx + A.pyr.by0, tid, lds);
}

} // namespace

// C: a corner stage with build_patch set (ofx.h: both frames' level 0, the two sets of patch planes, the relocated set or NULL);
// first: PatchBuild::first.  The pyramid arguments are ofx_pyramid_1ch's.
int ofx_pyramid_corner_1ch(const uint8_t *d_level0, int pitch0, int w, int h, uint8_t *const *d_levels, const int *pitches, int levels,
                           const ofx_corner_stage *C, int first, int window, int mode, void *stream)
{
    static_assert(kPyrThreads == 256, "the chain's block is four waves");
    OFX_REQUIRE(C && C->build_patch && C->levels == levels && levels >= 2, "ofx_pyramid_corner_1ch: bad corner stage");
    PyrCornerArgs a{};
    size_t lds_bytes = 0;
    int bx = 0, by = 0;
    OFX_TRY(ofx_pyramid_args(d_level0, pitch0, w, h, d_levels, pitches, levels, nullptr, 0, nullptr, nullptr, &a.pyr, &lds_bytes, &bx, &by));
    a.gx = bx;
    OFX_TRY(ofx_corner_args(C->level, C->levels, window, mode, C->d_uv, C->cols, C->d_status, nullptr, &a.hd, a.lv));
    a.hd.pair_status = C->d_pair_status;
    OFX_REQUIRE(C->patch_w > 0 && C->patch_h > 0 && C->d_patch_src[0] && C->d_patch_src[1] && C->d_patch[0][1] && C->d_patch[1][1],
                "ofx_pyramid_corner_1ch: incomplete patch description");
    const bool reloc = C->d_patch_reloc[1] != nullptr;
    PatchBuild &pb = a.patch;
    pb.n = levels - 1;
    pb.first = first ? 1 : 0;
    const long long stride = C->d_patch[1][1] - C->d_patch[0][1];
    OFX_REQUIRE(stride > -(1ll << 31) && stride < (1ll << 31), "ofx_pyramid_corner_1ch: the two sets of patch planes are too far apart");
    pb.frame_stride = (int)stride; // (may be negative: the sets swap roles from pair to pair; the device adds it as a 64-bit wrap-around)
    for (int k = 0; k < levels; ++k) {
        pb.pw[k] = C->patch_w >> k;
        pb.ph[k] = C->patch_h >> k;
        pb.pitch[k] = k ? C->patch_pitch[k] : 0;
        pb.off[k] = k ? (int)(C->d_patch[0][k] - C->d_patch[0][1]) : 0;
        OFX_REQUIRE(pb.pw[k] > 0 && pb.ph[k] > 0, "ofx_pyramid_corner_1ch: the patch is too small for %d levels", levels);
        if (k) {
            OFX_REQUIRE(C->d_patch[0][k] && (C->patch_pitch[k] & 3) == 0 && C->patch_pitch[k] >= ((pb.pw[k] + 3) & ~3) && ((uintptr_t)C->d_patch[0][k] & 3) == 0 &&
                            C->d_patch[1][k] == C->d_patch[0][k] + stride,
                        "ofx_pyramid_corner_1ch: bad patch plane at level %d", k);
            OFX_REQUIRE(!reloc || (C->d_patch_reloc[k] && C->d_patch_reloc[k] - C->d_patch_reloc[1] == pb.off[k] && ((uintptr_t)C->d_patch_reloc[k] & 3) == 0),
                        "ofx_pyramid_corner_1ch: the relocated patch planes must be laid out like the patch planes (level %d)", k);
            OFX_REQUIRE(((C->patch_w >> (k - 1)) & 1) == 0 && ((C->patch_h >> (k - 1)) & 1) == 0, "ofx_pyramid_corner_1ch: the patch must have even dimensions below its top level");
        }
    }
    for (int f = 0; f < 2; ++f)
        OFX_REQUIRE((C->patch_src_pitch[f] & 3) == 0 && C->patch_src_pitch[f] >= C->patch_w && ((uintptr_t)C->d_patch_src[f] & 3) == 0,
                    "ofx_pyramid_corner_1ch: bad patch source");
    if (reloc) { // (the conditions of ofx_stream_launch's repair)
        const ofx_geom &g0 = C->level[0].geom;
        OFX_REQUIRE(g0.row0 == 0 && g0.rows == g0.h && (C->cols[0] == 0 || C->cols[0] >= g0.w), "ofx_pyramid_corner_1ch: the repair needs level 0 to be the whole frames");
        OFX_REQUIRE(C->patch_w <= g0.w && C->patch_h <= g0.h, "ofx_pyramid_corner_1ch: the patch must lie inside the frame");
        const int lc = levels - 1, need = (window >> 1) + 5;
        OFX_REQUIRE((pb.pw[lc] >= need || pb.pw[lc] >= (g0.w >> lc)) && (pb.ph[lc] >= need || pb.ph[lc] >= (g0.h >> lc)),
                    "ofx_pyramid_corner_1ch: a %dx%d patch leaves %dx%d at the coarsest level, the repair needs %d", C->patch_w, C->patch_h, pb.pw[lc],
                    pb.ph[lc], need);
        a.hd.reloc = C->d_patch_reloc[1];
    }
    a.slot = PatchBuildSlot{{C->d_patch_src[0], C->d_patch_src[1]}, {C->patch_src_pitch[0], C->patch_src_pitch[1]}, C->d_patch[0][1]};
    const size_t corner_lds = (size_t)kCornerScratch + kCornerTileBytes + (size_t)levels * kCornerCacheBytes;
    const size_t lds = lds_bytes > corner_lds ? lds_bytes : corner_lds;
    // (experiment, wrong results: OFX_X_CHAIN_ONLY=1 launches the chain's block alone -- what it costs on an idle chip)
    static const bool chain_only = [] { const char *e = getenv("OFX_X_CHAIN_ONLY"); return e && atoi(e) != 0; }();
    const dim3 grid((unsigned)(chain_only ? 1 : bx * by + 1));
    hipStream_t st = ofx_stream(stream);
    if (mode == OFX_MODE_LK_FLOAT) hipLaunchKernelGGL((pyramid_corner_kernel<OFX_MODE_LK_FLOAT, false>), grid, dim3(kPyrThreads), lds, st, a);
    else if (mode == OFX_MODE_LK_FLOAT_FAST) hipLaunchKernelGGL((pyramid_corner_kernel<OFX_MODE_LK_FLOAT, true>), grid, dim3(kPyrThreads), lds, st, a);
    else hipLaunchKernelGGL((pyramid_corner_kernel<OFX_MODE_COMPAT_CPU, false>), grid, dim3(kPyrThreads), lds, st, a);
    OFX_HIP(hipGetLastError());
    return OFX_OK;
}
