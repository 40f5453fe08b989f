// Device scratch of the host-pointer wrappers (namespace gpu / namespace cpu, compat_gpu.cpp / compat_cpu.cpp).
//
// The reference allocates and frees device memory inside every wrapper (58 cudaMalloc/cudaFree per pyramid level,
// SURVEY 3.2).  Here a wrapper call carves its buffers out of a per-thread arena that is kept between calls: in the steady
// state of a frame loop (the same sizes every frame) no wrapper allocates at all.  A call that needs more than the arena
// holds takes extra blocks for its own duration; when it ends the arena is re-sized to what that call used, so the next
// one fits.  The copies are synchronous, as the reference's cudaMemcpy is (large ones are staged by several threads: compat_stage.cpp).
#pragma once

#include <vector>

#include "compat_stage.h"
#include "ofx_internal.h"

namespace ofx_compat {

struct Arena {
    void *base = nullptr;
    size_t cap = 0;
    int device = -1;
    int live = 0; // Scratch objects alive on this thread: they all carve from `base`, so there may be one at a time
    ~Arena()
    {
        // thread exit.  On the main thread this can run after the HIP runtime has shut down: then the block is the driver's
        // to reclaim with the process, and no HIP call is made.
        int dev = 0;
        if (base && hipGetDevice(&dev) == hipSuccess) (void)hipFree(base);
    }
};

inline Arena &arena()
{
    static thread_local Arena a;
    return a;
}

class Scratch {
  public:
    Scratch()
    {
        Arena &a = arena();
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && a.base && a.device != dev) { // the thread moved to another device
            (void)hipFree(a.base);
            a.base = nullptr;
            a.cap = 0;
        }
        a.device = dev;
        if (a.live++ > 0) { // a wrapper that opened a second Scratch would be handed the first one's memory again
            ofx_set_error("compat scratch: nested use on one thread (a wrapper called from inside a wrapper)");
            rc_ = OFX_E_STATE;
        }
    }
    ~Scratch()
    {
        for (void *p : extra_) (void)hipFree(p);
        Arena &a = arena();
        --a.live;
        if (need_ > a.cap) { // grow once, to what this call used in total (25 % slack for slightly larger frames)
            if (a.base) (void)hipFree(a.base);
            a.base = nullptr;
            a.cap = 0;
            const size_t want = need_ + need_ / 4;
            if (hipMalloc(&a.base, want) == hipSuccess) a.cap = want;
            else (void)hipGetLastError();
        }
    }
    template <typename T>
    T *alloc(size_t count)
    {
        if (rc_ != OFX_OK) return nullptr;
        const size_t bytes = (count * sizeof(T) + 64 + 255) / 256 * 256;
        Arena &a = arena();
        need_ += bytes;
        if (used_ + bytes <= a.cap) {
            void *p = static_cast<char *>(a.base) + used_;
            used_ += bytes;
            return static_cast<T *>(p);
        }
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            ofx_set_error("hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
            rc_ = OFX_E_HIP;
            return nullptr;
        }
        extra_.push_back(p);
        return static_cast<T *>(p);
    }
    template <typename T>
    T *upload(const T *host, size_t count)
    {
        T *d = alloc<T>(count);
        if (d) copy(d, host, count * sizeof(T), hipMemcpyHostToDevice);
        return d;
    }
    template <typename T>
    void download(T *host, const T *dev, size_t count)
    {
        if (rc_ == OFX_OK) copy(host, dev, count * sizeof(T), hipMemcpyDeviceToHost);
    }
    void run(int rc)
    {
        if (rc_ == OFX_OK) rc_ = rc;
    }
    bool ok() const { return rc_ == OFX_OK; }
    int rc() const { return rc_; }

  private:
    void copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
    {
        // chunked through pinned bounce buffers by a few threads (compat_stage.cpp); synchronous like the hipMemcpy of the reference
        const int rc = kind == hipMemcpyHostToDevice ? stage_h2d(dst, src, bytes) : stage_d2h(dst, src, bytes);
        if (rc != OFX_OK) rc_ = rc;
    }
    std::vector<void *> extra_;
    size_t used_ = 0, need_ = 0;
    int rc_ = OFX_OK;
};

// status of the calling thread's last gpu:: / cpu:: wrapper call (gpu_compat_last_status)
int &status();

inline bool args_ok(bool cond, const char *who)
{
    if (!cond) {
        ofx_set_error("%s: null pointer or non-positive size", who);
        status() = OFX_E_INVALID;
    }
    return cond;
}

} // namespace ofx_compat
