// Host <-> device transfers of the host-pointer wrappers (namespace gpu / namespace cpu, ofx_calc_opt_flow_host).
//
// The reference's call surface hands over PAGEABLE host memory (malloc, main.cu:95-104) and is synchronous, so every byte has
// to pass through the CPU once (pageable -> pinned) before the DMA engine can take it -- a plain hipMemcpy does exactly that,
// on one core, at ~12 GB/s: the drop-in surface at 4K was 26 ms per pair, 300 MB of PCIe traffic, none of it overlapped.
// Here a transfer is cut into chunks that a few threads (the caller + a small pool kept by the library) move concurrently:
// each copies its chunk into a pinned bounce buffer of its own and hands it to the DMA engine on a stream of its own, two
// buffers per thread so that the copy of one chunk overlaps the DMA of the one before.  Down-loads run the other way round.
// The call is still synchronous at return and retains nothing of the caller's (the contract of OptFlowGpu.cu:1909-1979).
//
// ofx_stage_h2d_ch0 additionally picks channel 0 out of a 3-channel image while it copies (the flow path only ever reads
// channel 0, OptFlowGpu.cu:1079 / OptFlowCPU.cpp:102): a third of the PCIe bytes for the images of gpu::calc_opt_flow.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "compat_stage.h"
#include "ofx_internal.h"

namespace ofx_compat {
namespace {

constexpr size_t kChunk = 2u << 20;      // bytes per chunk (and per pinned bounce buffer)
constexpr size_t kStagedMin = 3u << 20;  // smaller transfers: one hipMemcpy (a job's hand-over costs more than it saves)

struct Lane { // one thread's resources on one device
    int device = -1;
    hipStream_t st = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int next = 0;
    int prepare(int dev)
    {
        if (device == dev && st) return OFX_OK;
        release();
        OFX_HIP(hipSetDevice(dev));
        OFX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            OFX_HIP(hipHostMalloc(&pin[i], kChunk, hipHostMallocDefault));
            OFX_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        device = dev;
        return OFX_OK;
    }
    void release()
    {
        for (int i = 0; i < 2; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (pin[i]) (void)hipHostFree(pin[i]);
            ev[i] = nullptr;
            pin[i] = nullptr;
            busy[i] = false;
        }
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        device = -1;
    }
    // a bounce buffer whose previous DMA has completed
    int slot(int *out)
    {
        const int s = next;
        next ^= 1;
        if (busy[s]) {
            OFX_HIP(hipEventSynchronize(ev[s]));
            busy[s] = false;
        }
        *out = s;
        return OFX_OK;
    }
};

// channel 0 of n interleaved 3-channel pixels: out[i] = in[3 i].  The flow path's images pass through here on their way to the
// device, so the loop matters: byte by byte it runs at ~1 GB/s per core -- a 4K frame would take longer to thin out than to
// transfer.  With SSSE3 byte shuffles 16 pixels (48 source bytes) take three shuffles and two ORs.
#if defined(__x86_64__)
__attribute__((target("ssse3"))) void extract_ch0_ssse3(uint8_t *out, const uint8_t *in, size_t n)
{
    typedef char v16 __attribute__((vector_size(16)));
    const v16 s0 = {0, 3, 6, 9, 12, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    const v16 s1 = {-1, -1, -1, -1, -1, -1, 2, 5, 8, 11, 14, -1, -1, -1, -1, -1};
    const v16 s2 = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1, 4, 7, 10, 13};
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        v16 a, b, c;
        memcpy(&a, in + 3 * i, 16);
        memcpy(&b, in + 3 * i + 16, 16);
        memcpy(&c, in + 3 * i + 32, 16);
        const v16 r = __builtin_ia32_pshufb128(a, s0) | __builtin_ia32_pshufb128(b, s1) | __builtin_ia32_pshufb128(c, s2);
        memcpy(out + i, &r, 16);
    }
    for (; i < n; ++i) out[i] = in[3 * i];
}
#endif
void extract_ch0(uint8_t *out, const uint8_t *in, size_t n)
{
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("ssse3");
    if (have) return extract_ch0_ssse3(out, in, n);
#endif
    for (size_t i = 0; i < n; ++i) out[i] = in[3 * i];
}

enum Kind { H2D, D2H, H2D_CH0 };

struct Job {
    Kind kind;
    int device;
    uint8_t *dst;
    const uint8_t *src;
    size_t bytes;       // H2D / D2H
    int w, h, pitch;    // H2D_CH0: 3-channel w x h source, 1-channel destination plane of `pitch` bytes per row
    size_t n_chunks;
    int rows_per_chunk; // H2D_CH0
    std::atomic<size_t> next{0};
    std::atomic<int> rc{OFX_OK};
};

int run_chunks(Job &J, Lane &L)
{
    OFX_TRY(L.prepare(J.device));
    // a down-load in flight: the chunk whose DMA was issued last is copied out after the next one has been issued
    int pend_slot = -1;
    uint8_t *pend_dst = nullptr;
    size_t pend_len = 0;
    auto finish_pending = [&]() -> int {
        if (pend_slot < 0) return OFX_OK;
        OFX_HIP(hipEventSynchronize(L.ev[pend_slot]));
        L.busy[pend_slot] = false;
        memcpy(pend_dst, L.pin[pend_slot], pend_len);
        pend_slot = -1;
        return OFX_OK;
    };
    for (;;) {
        const size_t i = J.next.fetch_add(1);
        if (i >= J.n_chunks || J.rc.load() != OFX_OK) break;
        int s = 0;
        if (J.kind == D2H && pend_slot == L.next) OFX_TRY(finish_pending()); // (the slot about to be reused holds the pending chunk)
        OFX_TRY(L.slot(&s));
        if (J.kind == H2D) {
            const size_t off = i * kChunk, len = J.bytes - off < kChunk ? J.bytes - off : kChunk;
            memcpy(L.pin[s], J.src + off, len);
            OFX_HIP(hipMemcpyAsync(J.dst + off, L.pin[s], len, hipMemcpyHostToDevice, L.st));
            OFX_HIP(hipEventRecord(L.ev[s], L.st));
            L.busy[s] = true;
        } else if (J.kind == H2D_CH0) {
            const int y0 = (int)i * J.rows_per_chunk, y1 = y0 + J.rows_per_chunk < J.h ? y0 + J.rows_per_chunk : J.h;
            uint8_t *p = static_cast<uint8_t *>(L.pin[s]);
            extract_ch0(p, J.src + (size_t)y0 * (size_t)J.w * 3, (size_t)(y1 - y0) * (size_t)J.w); // (rows are tightly packed: one run)
            OFX_HIP(hipMemcpy2DAsync(J.dst + (size_t)y0 * (size_t)J.pitch, (size_t)J.pitch, p, (size_t)J.w, (size_t)J.w, (size_t)(y1 - y0),
                                     hipMemcpyHostToDevice, L.st));
            OFX_HIP(hipEventRecord(L.ev[s], L.st));
            L.busy[s] = true;
        } else {
            const size_t off = i * kChunk, len = J.bytes - off < kChunk ? J.bytes - off : kChunk;
            OFX_HIP(hipMemcpyAsync(L.pin[s], J.src + off, len, hipMemcpyDeviceToHost, L.st));
            OFX_HIP(hipEventRecord(L.ev[s], L.st));
            L.busy[s] = true;
            OFX_TRY(finish_pending());
            pend_slot = s;
            pend_dst = J.dst + off;
            pend_len = len;
        }
    }
    OFX_TRY(finish_pending());
    // synchronous at return: every DMA this thread issued has completed (the bounce buffers are free again)
    for (int s = 0; s < 2; ++s)
        if (L.busy[s]) {
            OFX_HIP(hipEventSynchronize(L.ev[s]));
            L.busy[s] = false;
        }
    return OFX_OK;
}

class Stager {
  public:
    static Stager &get()
    {
        static Stager s;
        return s;
    }
    int threads() const { return (int)pool_.size(); }
    int run(Job &J, bool solo = false)
    {
        static thread_local Lane mine;
        if (solo) return run_chunks(J, mine); // a small transfer: the caller's own lane, nobody is woken
        std::unique_lock<std::mutex> serial(serial_); // one job at a time (wrappers on several host threads take turns)
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &J;
            active_ = (int)pool_.size();
            active_atomic_.store(active_, std::memory_order_release);
            ++gen_;
            gen_atomic_.store(gen_, std::memory_order_release);
        }
        cv_work_.notify_all();
        const int rc = run_chunks(J, mine);
        if (rc != OFX_OK) J.rc.store(rc);
        {
            const auto t_spin = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
            while (active_atomic_.load(std::memory_order_acquire) != 0 && std::chrono::steady_clock::now() < t_spin) __builtin_ia32_pause();
            std::unique_lock<std::mutex> g(m_);
            cv_done_.wait(g, [&] { return active_ == 0; });
            job_ = nullptr;
        }
        return J.rc.load();
    }

  private:
    Stager()
    {
        const char *e = getenv("OFX_STAGE_THREADS");
        int n = e ? atoi(e) : 3;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw - 1) n = hw - 1;
        if (n < 0) n = 0;
        for (int i = 0; i < n; ++i) pool_.emplace_back([this] { loop(); });
    }
    ~Stager()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            stop_atomic_.store(true);
        }
        cv_work_.notify_all();
        for (std::thread &t : pool_) t.join();
    }
    void loop()
    {
        Lane lane;
        unsigned long seen = 0;
        for (;;) {
            Job *J = nullptr;
            {
                // a frame loop hands over a transfer every 100-200 us: spin that long for the next one before going to sleep (a
                // condition-variable wake-up costs 30-50 us, as much as a 1080p level's transfer itself)
                const auto t_spin = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
                while (gen_atomic_.load(std::memory_order_acquire) == seen && !stop_atomic_.load(std::memory_order_relaxed) &&
                       std::chrono::steady_clock::now() < t_spin)
                    __builtin_ia32_pause();
                std::unique_lock<std::mutex> g(m_);
                cv_work_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) break;
                seen = gen_;
                J = job_;
            }
            if (J) {
                const int rc = run_chunks(*J, lane);
                if (rc != OFX_OK) J->rc.store(rc);
            }
            {
                std::lock_guard<std::mutex> g(m_);
                --active_;
                active_atomic_.store(active_, std::memory_order_release);
            }
            cv_done_.notify_all();
        }
        // (the process is going down, possibly after the HIP runtime: the lane's resources are left to the driver)
    }
    std::vector<std::thread> pool_;
    std::mutex m_, serial_;
    std::condition_variable cv_work_, cv_done_;
    Job *job_ = nullptr;
    unsigned long gen_ = 0;
    std::atomic<unsigned long> gen_atomic_{0}; // (copy of gen_ the spinning workers poll without the mutex)
    std::atomic<bool> stop_atomic_{false};
    std::atomic<int> active_atomic_{0};
    int active_ = 0;
    bool stop_ = false;
};

int plain(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    OFX_HIP(hipMemcpy(dst, src, bytes, kind)); // blocking, ordered after the null-stream kernels
    return OFX_OK;
}

bool staging_on()
{
    static const bool on = [] {
        const char *e = getenv("OFX_STAGE_THREADS");
        return !(e && atoi(e) < 0);
    }();
    return on;
}

} // namespace

int stage_h2d(void *d_dst, const void *h_src, size_t bytes)
{
    if (bytes == 0) return OFX_OK;
    if (bytes < kStagedMin || !staging_on()) return plain(d_dst, h_src, bytes, hipMemcpyHostToDevice);
    Job J;
    J.kind = H2D;
    OFX_HIP(hipGetDevice(&J.device));
    J.dst = static_cast<uint8_t *>(d_dst);
    J.src = static_cast<const uint8_t *>(h_src);
    J.bytes = bytes;
    J.n_chunks = (bytes + kChunk - 1) / kChunk;
    // (the destination is scratch no earlier kernel still uses: every wrapper ends with a blocking down-load)
    return Stager::get().run(J);
}

int stage_d2h(void *h_dst, const void *d_src, size_t bytes)
{
    if (bytes == 0) return OFX_OK;
    if (bytes < kStagedMin || !staging_on()) return plain(h_dst, d_src, bytes, hipMemcpyDeviceToHost);
    OFX_HIP(hipStreamSynchronize(nullptr)); // the kernels that produce the data run on the null stream; the lanes' streams do not wait for it
    Job J;
    J.kind = D2H;
    OFX_HIP(hipGetDevice(&J.device));
    J.dst = static_cast<uint8_t *>(h_dst);
    J.src = static_cast<const uint8_t *>(d_src);
    J.bytes = bytes;
    J.n_chunks = (bytes + kChunk - 1) / kChunk;
    return Stager::get().run(J);
}

int stage_h2d_ch0(uint8_t *d_dst1, int pitch, const uint8_t *h_src3, int w, int h)
{
    OFX_REQUIRE(d_dst1 && h_src3 && w > 0 && h > 0 && pitch >= w, "stage_h2d_ch0: bad arguments");
    Job J;
    J.kind = H2D_CH0;
    OFX_HIP(hipGetDevice(&J.device));
    J.dst = d_dst1;
    J.src = h_src3;
    J.w = w;
    J.h = h;
    J.pitch = pitch;
    J.rows_per_chunk = (int)(kChunk / (size_t)w) > 0 ? (int)(kChunk / (size_t)w) : 1;
    if ((size_t)w > kChunk) {
        ofx_set_error("stage_h2d_ch0: rows of %d pixels exceed the bounce buffers", w);
        return OFX_E_UNSUPPORTED;
    }
    J.n_chunks = ((size_t)h + J.rows_per_chunk - 1) / J.rows_per_chunk;
    return Stager::get().run(J, (size_t)w * (size_t)h * 3 < kStagedMin || !staging_on());
}

int stage_threads() { return staging_on() ? Stager::get().threads() + 1 : 1; }

} // namespace ofx_compat

extern "C" int ofx_stage_threads(void) { return ofx_compat::stage_threads(); }
