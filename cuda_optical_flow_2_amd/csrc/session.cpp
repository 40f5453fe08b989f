// Device-resident session: the frame loop of main.cu:192-272 with every buffer living in HBM.
//
// One hipMalloc arena holds, per pyramid level: the previous and the next frame's 1-channel planes, a scratch
// plane for the shifted next frame, the flow field, and the 2-float shift vector.  Nothing is allocated or freed
// while frames flow (the reference does 58 cudaMalloc/cudaFree calls per level).
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "ofx_internal.h"

namespace {
constexpr size_t kAlign = 256;
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
} // namespace

struct ofx_session {
    ofx_params p{};
    int w[OFX_MAX_LEVELS]{}, h[OFX_MAX_LEVELS]{}, pitch[OFX_MAX_LEVELS]{};
    int own0[OFX_MAX_LEVELS]{}, own1[OFX_MAX_LEVELS]{}; // rows this rank computes
    int buf0[OFX_MAX_LEVELS]{}, buf1[OFX_MAX_LEVELS]{}; // rows the plane buffers hold
    int cmp0[OFX_MAX_LEVELS]{}, cmp1[OFX_MAX_LEVELS]{}; // rows this rank downsamples itself
    uint8_t *plane[3][OFX_MAX_LEVELS]{};                // 0 prev, 1 next, 2 shifted scratch
    float *flow[OFX_MAX_LEVELS]{};
    float *uv = nullptr;        // 2 floats per level
    uint8_t *staging = nullptr; // one tightly packed 3ch level-0 frame for host uploads
    void *arena = nullptr;
    size_t arena_bytes = 0;
    bool have_next = false, have_prev = false;
    // optional timing of the level-0 fused LK launch: event pairs recorded on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
};

static ofx_geom level_geom(const ofx_session *s, int k, int out0, int out1)
{
    ofx_geom g;
    g.w = s->w[k];
    g.h = s->h[k];
    g.pitch = s->pitch[k];
    g.row0 = s->buf0[k];
    g.rows = s->buf1[k] - s->buf0[k];
    g.out_y0 = out0;
    g.out_y1 = out1;
    return g;
}

extern "C" int ofx_session_create(const ofx_params *p, ofx_session **out)
{
    OFX_REQUIRE(p && out, "ofx_session_create: null argument");
    OFX_REQUIRE(p->width > 0 && p->height > 0, "ofx_session_create: bad size %dx%d", p->width, p->height);
    OFX_REQUIRE(p->levels >= 1 && p->levels <= OFX_MAX_LEVELS, "ofx_session_create: levels %d out of range", p->levels);
    OFX_REQUIRE(p->window >= 3 && (p->window & 1), "ofx_session_create: window must be odd and >= 3");
    OFX_REQUIRE(p->mode == OFX_MODE_COMPAT_CPU || p->mode == OFX_MODE_LK_FLOAT, "ofx_session_create: bad mode %d", p->mode);
    OFX_REQUIRE((p->width >> (p->levels - 1)) > 0 && (p->height >> (p->levels - 1)) > 0,
                "ofx_session_create: %d levels is too many for %dx%d", p->levels, p->width, p->height);
    for (int k = 0; k + 1 < p->levels; ++k)
        OFX_REQUIRE(((p->width >> k) & 1) == 0 && ((p->height >> k) & 1) == 0,
                    "ofx_session_create: level %d is %dx%d; every level that is downsampled must have even dimensions "
                    "(the reference assumes a source stride of exactly 2*w, OptFlowCPU.cpp:117)",
                    k, p->width >> k, p->height >> k);
    OFX_HIP(hipSetDevice(p->device));

    ofx_session *s = new (std::nothrow) ofx_session();
    OFX_REQUIRE(s != nullptr, "ofx_session_create: out of host memory");
    s->p = *p;
    size_t total = 0;
    std::vector<size_t> off_plane[3], off_flow;
    for (int k = 0; k < p->levels; ++k) {
        s->w[k] = p->width >> k;
        s->h[k] = p->height >> k;
        s->pitch[k] = (int)align_up((size_t)s->w[k], 64);
        if (p->sharded) {
            s->own0[k] = p->own_y0[k];
            s->own1[k] = p->own_y1[k];
            s->buf0[k] = p->buf_y0[k];
            s->buf1[k] = p->buf_y1[k];
            const bool has_comp = p->comp_y1[k] > 0;
            s->cmp0[k] = has_comp ? p->comp_y0[k] : s->own0[k];
            s->cmp1[k] = has_comp ? p->comp_y1[k] : s->own1[k];
            const bool ok = 0 <= s->buf0[k] && s->buf0[k] <= s->own0[k] && s->own0[k] <= s->own1[k] &&
                            s->own1[k] <= s->buf1[k] && s->buf1[k] <= s->h[k] && s->buf0[k] < s->buf1[k] &&
                            s->buf0[k] <= s->cmp0[k] && s->cmp0[k] <= s->own0[k] && s->own1[k] <= s->cmp1[k] &&
                            s->cmp1[k] <= s->buf1[k];
            if (!ok) {
                ofx_set_error("ofx_session_create: level %d shard rows own [%d,%d) buf [%d,%d) invalid for height %d", k,
                              s->own0[k], s->own1[k], s->buf0[k], s->buf1[k], s->h[k]);
                delete s;
                return OFX_E_INVALID;
            }
        } else {
            s->own0[k] = s->buf0[k] = s->cmp0[k] = 0;
            s->own1[k] = s->buf1[k] = s->cmp1[k] = s->h[k];
        }
        const size_t plane_bytes = align_up((size_t)s->pitch[k] * (size_t)(s->buf1[k] - s->buf0[k]) + 64, kAlign);
        for (int t = 0; t < 3; ++t) {
            off_plane[t].push_back(total);
            total += plane_bytes;
        }
        off_flow.push_back(total);
        const size_t own_rows = (size_t)(s->own1[k] - s->own0[k]);
        total += align_up((own_rows ? own_rows : 1) * (size_t)s->w[k] * 2 * sizeof(float), kAlign);
    }
    const size_t off_uv = total;
    total += align_up((size_t)OFX_MAX_LEVELS * 2 * sizeof(float), kAlign);
    const size_t off_staging = total;
    total += align_up((size_t)p->width * (size_t)p->height * 3, kAlign);

    hipError_t e = hipMalloc(&s->arena, total);
    if (e != hipSuccess) {
        ofx_set_error("ofx_session_create: hipMalloc(%zu bytes): %s", total, hipGetErrorString(e));
        delete s;
        return OFX_E_HIP;
    }
    s->arena_bytes = total;
    e = hipMemset(s->arena, 0, total);
    if (e != hipSuccess) {
        ofx_set_error("ofx_session_create: hipMemset: %s", hipGetErrorString(e));
        (void)hipFree(s->arena);
        delete s;
        return OFX_E_HIP;
    }
    uint8_t *base = static_cast<uint8_t *>(s->arena);
    for (int k = 0; k < p->levels; ++k) {
        for (int t = 0; t < 3; ++t) s->plane[t][k] = base + off_plane[t][k];
        s->flow[k] = reinterpret_cast<float *>(base + off_flow[k]);
    }
    s->uv = reinterpret_cast<float *>(base + off_uv);
    s->staging = base + off_staging;
    *out = s;
    return OFX_OK;
}

extern "C" int ofx_session_destroy(ofx_session *s)
{
    if (!s) return OFX_OK;
    hipError_t e = hipSuccess;
    for (hipEvent_t ev : s->ev) (void)hipEventDestroy(ev);
    if (s->arena) {
        (void)hipSetDevice(s->p.device);
        e = hipFree(s->arena);
    }
    delete s;
    if (e != hipSuccess) {
        ofx_set_error("ofx_session_destroy: hipFree: %s", hipGetErrorString(e));
        return OFX_E_HIP;
    }
    return OFX_OK;
}

// copy rows [buf0,buf1) of a tightly packed w-bytes-per-row frame into the level-0 `next` plane
static int load_level0(ofx_session *s, const uint8_t *src, bool src_is_host, int src_pitch, hipStream_t st)
{
    const int rows = s->buf1[0] - s->buf0[0];
    OFX_HIP(hipMemcpy2DAsync(s->plane[1][0], (size_t)s->pitch[0], src + (size_t)s->buf0[0] * (size_t)src_pitch, (size_t)src_pitch,
                             (size_t)s->w[0], (size_t)rows, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    s->have_next = true;
    return OFX_OK;
}

extern "C" int ofx_session_set_frame_host(ofx_session *s, const uint8_t *h_gray1, void *stream)
{
    OFX_REQUIRE(s && h_gray1, "ofx_session_set_frame_host: null argument");
    return load_level0(s, h_gray1, true, s->w[0], ofx_stream(stream));
}

extern "C" int ofx_session_set_frame_device(ofx_session *s, const uint8_t *d_gray1, int pitch, void *stream)
{
    OFX_REQUIRE(s && d_gray1, "ofx_session_set_frame_device: null argument");
    OFX_REQUIRE(pitch >= s->w[0], "ofx_session_set_frame_device: pitch %d < width %d", pitch, s->w[0]);
    return load_level0(s, d_gray1, false, pitch, ofx_stream(stream));
}

extern "C" int ofx_session_set_frame_host_3ch(ofx_session *s, const uint8_t *h_img3, void *stream)
{
    OFX_REQUIRE(s && h_img3, "ofx_session_set_frame_host_3ch: null argument");
    OFX_REQUIRE(!s->p.sharded, "ofx_session_set_frame_host_3ch: not available on a sharded session");
    hipStream_t st = ofx_stream(stream);
    OFX_HIP(hipMemcpyAsync(s->staging, h_img3, (size_t)s->w[0] * (size_t)s->h[0] * 3, hipMemcpyHostToDevice, st));
    // the reference reads channel 0 only (OptFlowCPU.cpp:102, OptFlowGpu.cu:1079)
    OFX_TRY(ofx_extract_ch0(s->staging, s->plane[1][0], s->w[0], s->h[0], s->pitch[0], stream));
    s->have_next = true;
    return OFX_OK;
}

extern "C" int ofx_session_downsample_level(ofx_session *s, int k, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_downsample_level: null session");
    OFX_REQUIRE(k >= 1 && k < s->p.levels, "ofx_session_downsample_level: level %d out of range", k);
    OFX_REQUIRE(s->have_next, "ofx_session_downsample_level: no frame loaded");
    const ofx_geom g = level_geom(s, k, s->cmp0[k], s->cmp1[k]);
    return ofx_downsample_1ch(s->plane[1][k - 1], s->pitch[k - 1], s->buf0[k - 1], s->buf1[k - 1] - s->buf0[k - 1],
                              s->plane[1][k], &g, stream);
}

extern "C" int ofx_session_build_pyramid(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_build_pyramid: null session");
    if (!s->have_next) {
        ofx_set_error("ofx_session_build_pyramid: no frame loaded");
        return OFX_E_STATE;
    }
    for (int k = 1; k < s->p.levels; ++k) OFX_TRY(ofx_session_downsample_level(s, k, stream));
    return OFX_OK;
}

extern "C" int ofx_session_compute_uv(ofx_session *s, int level, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_compute_uv: null session");
    OFX_REQUIRE(level >= 0 && level < s->p.levels, "ofx_session_compute_uv: level %d out of range", level);
    if (level == s->p.levels - 1) return OFX_OK; // top level is not shifted (OptFlowCPU.cpp:321)
    const float *lv[OFX_MAX_LEVELS] = {};
    for (int k = 0; k < s->p.levels; ++k) lv[k] = s->flow[k];
    return ofx_shift_vector(lv, level, s->p.levels, s->uv + 2 * level, stream);
}

extern "C" int ofx_session_run_level(ofx_session *s, int level, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_level: null session");
    OFX_REQUIRE(level >= 0 && level < s->p.levels, "ofx_session_run_level: level %d out of range", level);
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_run_level: need a previous and a next frame (load, build, swap, load, build)");
        return OFX_E_STATE;
    }
    const uint8_t *next = s->plane[1][level];
    if (level != s->p.levels - 1) {
        // shift every row the LK stencil will read: own rows +- (radius + 1), clipped to the buffer
        const int halo = (s->p.window >> 1) + 1;
        int y0 = s->own0[level] - halo, y1 = s->own1[level] + halo;
        if (y0 < s->buf0[level]) y0 = s->buf0[level];
        if (y1 > s->buf1[level]) y1 = s->buf1[level];
        const ofx_geom gs = level_geom(s, level, y0, y1);
        OFX_TRY(ofx_shift_1ch(s->plane[1][level], s->plane[2][level], &gs, s->uv + 2 * level, stream));
        next = s->plane[2][level];
    }
    const ofx_geom g = level_geom(s, level, s->own0[level], s->own1[level]);
    const bool timed = s->timing && level == 0 && s->ev_used + 2 <= s->ev.size();
    if (timed) OFX_HIP(hipEventRecord(s->ev[s->ev_used], ofx_stream(stream)));
    OFX_TRY(ofx_lk_level(s->plane[0][level], next, &g, s->p.window, s->p.mode, s->flow[level], s->own0[level], stream));
    if (timed) {
        OFX_HIP(hipEventRecord(s->ev[s->ev_used + 1], ofx_stream(stream)));
        s->ev_used += 2;
    }
    return OFX_OK;
}

extern "C" int ofx_session_timing(ofx_session *s, int max_launches)
{
    OFX_REQUIRE(s && max_launches >= 0, "ofx_session_timing: bad arguments");
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    s->ev.clear();
    s->ev_used = 0;
    s->timing = max_launches > 0;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t e;
        OFX_HIP(hipEventCreate(&e));
        s->ev.push_back(e);
    }
    return OFX_OK;
}

extern "C" int ofx_session_timing_read(ofx_session *s, double *avg_us, double *min_us, int *launches)
{
    OFX_REQUIRE(s && avg_us && launches, "ofx_session_timing_read: bad arguments");
    double sum = 0, mn = 1e30;
    const int n = (int)(s->ev_used / 2);
    for (int i = 0; i < n; ++i) {
        OFX_HIP(hipEventSynchronize(s->ev[2 * i + 1]));
        float ms = 0;
        OFX_HIP(hipEventElapsedTime(&ms, s->ev[2 * i], s->ev[2 * i + 1]));
        sum += ms * 1e3;
        if (ms * 1e3 < mn) mn = ms * 1e3;
    }
    *avg_us = n ? sum / n : 0.0;
    if (min_us) *min_us = n ? mn : 0.0;
    *launches = n;
    s->ev_used = 0;
    return OFX_OK;
}

// Shift vectors of every level at once (ofx_corner_flows); meaningful on the rank whose buffers start at row 0.
extern "C" int ofx_session_corner_flows(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_corner_flows: null session");
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_corner_flows: need a previous and a next frame");
        return OFX_E_STATE;
    }
    ofx_lk_desc d[OFX_MAX_LEVELS];
    for (int k = 0; k < s->p.levels; ++k)
        d[k] = ofx_lk_desc{s->plane[0][k], s->plane[1][k], level_geom(s, k, s->own0[k], s->own1[k]), s->flow[k], s->own0[k]};
    return ofx_corner_flows(d, s->p.levels, s->p.window, s->p.mode, s->uv, stream);
}

// Every level's shift in one launch, then every level's fused LK in one launch, using the uv slots as they stand.
extern "C" int ofx_session_run_levels(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_levels: null session");
    if (!s->have_prev || !s->have_next) {
        ofx_set_error("ofx_session_run_levels: need a previous and a next frame");
        return OFX_E_STATE;
    }
    const int L = s->p.levels, halo = (s->p.window >> 1) + 1;
    ofx_shift_desc sh[OFX_MAX_LEVELS];
    ofx_lk_desc lk[OFX_MAX_LEVELS];
    int ns = 0, nl = 0;
    for (int k = L - 1; k >= 0; --k) { // coarse levels first: their few waves start at once and finish early
        const uint8_t *next = s->plane[1][k];
        if (k != L - 1) {
            int y0 = s->own0[k] - halo, y1 = s->own1[k] + halo;
            if (y0 < s->buf0[k]) y0 = s->buf0[k];
            if (y1 > s->buf1[k]) y1 = s->buf1[k];
            sh[ns++] = ofx_shift_desc{s->plane[1][k], s->plane[2][k], level_geom(s, k, y0, y1), s->uv + 2 * k};
            next = s->plane[2][k];
        }
        lk[nl++] = ofx_lk_desc{s->plane[0][k], next, level_geom(s, k, s->own0[k], s->own1[k]), s->flow[k], s->own0[k]};
    }
    if (ns) OFX_TRY(ofx_shift_levels(sh, ns, stream));
    const bool timed = s->timing && s->ev_used + 2 <= s->ev.size();
    if (timed) OFX_HIP(hipEventRecord(s->ev[s->ev_used], ofx_stream(stream)));
    OFX_TRY(ofx_lk_levels(lk, nl, s->p.window, s->p.mode, stream));
    if (timed) {
        OFX_HIP(hipEventRecord(s->ev[s->ev_used + 1], ofx_stream(stream)));
        s->ev_used += 2;
    }
    return OFX_OK;
}

extern "C" int ofx_session_run_flow(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_flow: null session");
    OFX_REQUIRE(!s->p.sharded || s->buf0[0] == 0, "ofx_session_run_flow: on a sharded session only the rank holding row 0 can "
                                                   "form the shift vectors; use corner_flows + broadcast + run_levels");
    OFX_TRY(ofx_session_corner_flows(s, stream));
    return ofx_session_run_levels(s, stream);
}

// The reference's literal sequence (one level after the other, main.cu:256-262); kept for comparison and tests.
extern "C" int ofx_session_run_flow_sequential(ofx_session *s, void *stream)
{
    OFX_REQUIRE(s, "ofx_session_run_flow_sequential: null session");
    for (int k = s->p.levels - 1; k >= 0; --k) {
        OFX_TRY(ofx_session_compute_uv(s, k, stream));
        OFX_TRY(ofx_session_run_level(s, k, stream));
    }
    return OFX_OK;
}

extern "C" int ofx_session_swap(ofx_session *s)
{
    OFX_REQUIRE(s, "ofx_session_swap: null session");
    for (int k = 0; k < s->p.levels; ++k) {
        uint8_t *t = s->plane[0][k];
        s->plane[0][k] = s->plane[1][k];
        s->plane[1][k] = t;
    }
    s->have_prev = s->have_next;
    s->have_next = false;
    return OFX_OK;
}

extern "C" int ofx_session_plane(ofx_session *s, int which, int level, uint8_t **d_ptr, ofx_geom *geom)
{
    OFX_REQUIRE(s && which >= 0 && which < 3 && level >= 0 && level < s->p.levels, "ofx_session_plane: bad arguments");
    if (d_ptr) *d_ptr = s->plane[which][level];
    if (geom) *geom = level_geom(s, level, s->own0[level], s->own1[level]);
    return OFX_OK;
}

extern "C" int ofx_session_flow(ofx_session *s, int level, float **d_ptr, int *row0, int *rows)
{
    OFX_REQUIRE(s && level >= 0 && level < s->p.levels, "ofx_session_flow: bad arguments");
    if (d_ptr) *d_ptr = s->flow[level];
    if (row0) *row0 = s->own0[level];
    if (rows) *rows = s->own1[level] - s->own0[level];
    return OFX_OK;
}

extern "C" int ofx_session_shift_uv(ofx_session *s, int level, float **d_uv)
{
    OFX_REQUIRE(s && d_uv && level >= 0 && level < s->p.levels, "ofx_session_shift_uv: bad arguments");
    *d_uv = s->uv + 2 * level;
    return OFX_OK;
}

extern "C" int ofx_session_get_flow_host(ofx_session *s, int level, float *h_dst, void *stream)
{
    OFX_REQUIRE(s && h_dst && level >= 0 && level < s->p.levels, "ofx_session_get_flow_host: bad arguments");
    const size_t bytes = (size_t)(s->own1[level] - s->own0[level]) * (size_t)s->w[level] * 2 * sizeof(float);
    hipStream_t st = ofx_stream(stream);
    OFX_HIP(hipMemcpyAsync(h_dst, s->flow[level], bytes, hipMemcpyDeviceToHost, st));
    OFX_HIP(hipStreamSynchronize(st));
    return OFX_OK;
}

// gpu::calc_opt_flow (OptFlowGpu.cuh:33, OptFlowGpu.cu:1909-1979) with host pointers: upload both images and the
// two floats of every coarser flow level that the shift reads, run one level on the device, download its flow.
extern "C" int ofx_calc_opt_flow_host(const uint8_t *h_prev3, const uint8_t *h_next3, int w, int h, float **h_flow_pyr, int level,
                                      int max_level, int window, int mode)
{
    OFX_REQUIRE(h_prev3 && h_next3 && h_flow_pyr && w > 0 && h > 0, "ofx_calc_opt_flow_host: bad arguments");
    OFX_REQUIRE(max_level >= 1 && max_level <= OFX_MAX_LEVELS && level >= 0 && level < max_level,
                "ofx_calc_opt_flow_host: bad level %d of %d", level, max_level);
    OFX_REQUIRE(h_flow_pyr[level] != nullptr, "ofx_calc_opt_flow_host: flow level %d is null", level);
    const size_t n = (size_t)w * (size_t)h;
    const int pitch = (int)align_up((size_t)w, 64);
    const size_t plane = align_up((size_t)pitch * (size_t)h + 64, kAlign);
    const size_t img3 = align_up(3 * n, kAlign);
    const size_t flow_b = align_up(2 * n * sizeof(float), kAlign);
    const size_t total = 2 * img3 + 3 * plane + flow_b + 2 * kAlign;
    uint8_t *base = nullptr;
    OFX_HIP(hipMalloc(reinterpret_cast<void **>(&base), total));
    int rc = OFX_OK;
    auto fail = [&](int code) {
        (void)hipFree(base);
        return code;
    };
    uint8_t *d_p3 = base, *d_n3 = base + img3, *d_p1 = d_n3 + img3, *d_n1 = d_p1 + plane, *d_s1 = d_n1 + plane;
    float *d_flow = reinterpret_cast<float *>(d_s1 + plane);
    float *d_coarse = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(d_flow) + flow_b); // 2 floats per level
    float *d_uv = d_coarse + 2 * OFX_MAX_LEVELS;
    hipError_t e = hipMemcpy(d_p3, h_prev3, 3 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_n3, h_next3, 3 * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ofx_set_error("ofx_calc_opt_flow_host: upload: %s", hipGetErrorString(e));
        return fail(OFX_E_HIP);
    }
    if ((rc = ofx_extract_ch0(d_p3, d_p1, w, h, pitch, nullptr)) != OFX_OK) return fail(rc);
    if ((rc = ofx_extract_ch0(d_n3, d_n1, w, h, pitch, nullptr)) != OFX_OK) return fail(rc);
    ofx_geom g{w, h, pitch, 0, h, 0, h};
    const uint8_t *d_next = d_n1;
    if (level != max_level - 1) {
        float coarse[2 * OFX_MAX_LEVELS] = {};
        const float *lv[OFX_MAX_LEVELS] = {};
        for (int k = level + 1; k < max_level; ++k) {
            if (!h_flow_pyr[k]) {
                ofx_set_error("ofx_calc_opt_flow_host: flow level %d is null", k);
                return fail(OFX_E_INVALID);
            }
            coarse[2 * k] = h_flow_pyr[k][0];
            coarse[2 * k + 1] = h_flow_pyr[k][1];
            lv[k] = d_coarse + 2 * k;
        }
        e = hipMemcpy(d_coarse, coarse, sizeof coarse, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ofx_set_error("ofx_calc_opt_flow_host: upload: %s", hipGetErrorString(e));
            return fail(OFX_E_HIP);
        }
        if ((rc = ofx_shift_vector(lv, level, max_level, d_uv, nullptr)) != OFX_OK) return fail(rc);
        if ((rc = ofx_shift_1ch(d_n1, d_s1, &g, d_uv, nullptr)) != OFX_OK) return fail(rc);
        d_next = d_s1;
    }
    if ((rc = ofx_lk_level(d_p1, d_next, &g, window, mode, d_flow, 0, nullptr)) != OFX_OK) return fail(rc);
    e = hipMemcpy(h_flow_pyr[level], d_flow, 2 * n * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        ofx_set_error("ofx_calc_opt_flow_host: download: %s", hipGetErrorString(e));
        return fail(OFX_E_HIP);
    }
    OFX_HIP(hipFree(base));
    return OFX_OK;
}
