// Error channel, argument checks and small queries of libofx_hip.so.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "ofx_internal.h"

static thread_local char g_err[512] = "";

void ofx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *ofx_last_error(void) { return g_err; }

extern "C" int ofx_abi_version(void) { return 10; }

extern "C" int ofx_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        ofx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -OFX_E_HIP;
    }
    return n;
}

int ofx_check_geom(const ofx_geom *g, const char *who)
{
    OFX_REQUIRE(g != nullptr, "%s: geometry is null", who);
    OFX_REQUIRE(g->w > 0 && g->h > 0, "%s: bad size %dx%d", who, g->w, g->h);
    OFX_REQUIRE(g->pitch >= g->w && (g->pitch & 3) == 0, "%s: pitch %d must be a multiple of 4 and >= w=%d", who,
                g->pitch, g->w);
    OFX_REQUIRE(g->rows > 0 && g->row0 >= 0 && g->row0 + g->rows <= g->h,
                "%s: buffer rows [%d,%d) not inside the image height %d", who, g->row0, g->row0 + g->rows, g->h);
    OFX_REQUIRE(g->out_y0 >= 0 && g->out_y0 <= g->out_y1 && g->out_y1 <= g->h, "%s: output rows [%d,%d) not inside [0,%d)",
                who, g->out_y0, g->out_y1, g->h);
    return OFX_OK;
}

int ofx_check_halo(const ofx_geom *g, int halo, const char *who)
{
    if (g->out_y1 <= g->out_y0) return OFX_OK;
    const int lo = g->out_y0 - halo > 0 ? g->out_y0 - halo : 0;
    const int hi = g->out_y1 + halo < g->h ? g->out_y1 + halo : g->h;
    OFX_REQUIRE(lo >= g->row0 && hi <= g->row0 + g->rows,
                "%s: rows [%d,%d) are needed (halo %d) but the buffer holds [%d,%d)", who, lo, hi, halo, g->row0,
                g->row0 + g->rows);
    return OFX_OK;
}

// ---- roctx ranges around the session's stages (tracing) ------------------------------------------------------
// `rocprofv3 --marker-trace` shows them next to the kernels, so a timeline of a frame loop needs no custom tracer.  The marker
// library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so) the first time a range is opened with
// OFX_ROCTX=1 in the environment: no link dependency, and nothing at all -- one load of a static flag -- when the variable is
// not set.
namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_push = nullptr;
roctx_pop_fn g_pop = nullptr;
bool roctx_on()
{
    static const bool on = [] {
        const char *e = getenv("OFX_ROCTX");
        if (!e || atoi(e) <= 0) return false;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                g_push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
                g_pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
                if (g_push && g_pop) return true;
            }
        }
        return false;
    }();
    return on;
}
} // namespace

void ofx_range_push(const char *name)
{
    if (roctx_on()) (void)g_push(name);
}
void ofx_range_pop(void)
{
    if (roctx_on()) (void)g_pop();
}
