// ctx.cpp -- context lifetime, error strings, HIP-event kernel timing (include/pantax_hip.h).
#include <exception>
#include <cstdarg>
#include <algorithm>
#include <cstring>
#include <thread>
#include <unistd.h>
#include "common.hpp"

namespace ptx {

// error text of the calling thread's last failed call: messages are per thread, so concurrent callers of one ctx
// never read each other's
static thread_local std::string g_init_err;
static thread_local std::string g_thread_err;
static thread_local const Ctx *g_thread_err_ctx = nullptr;

int fail(Ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        // the per-thread copy is what pantax_hip_last_error returns to the failing thread; the shared one (for other threads)
        // has its own lock: fail() is also reached before an entry point takes the ctx lock
        { std::lock_guard<std::mutex> g(ctx->err_mu); ctx->err = buf; }
        g_thread_err = buf; g_thread_err_ctx = ctx;
    } else g_init_err = buf;
    return code;
}

KTimer::KTimer(Ctx *c, const char *nm) : ctx(c), name(nm) {
    if (!ctx->timing) return;
    if (!ctx->timing_filter.empty()) {   // only the named launches are bracketed ("a" or "a|b|c")
        const std::string &f = ctx->timing_filter;
        const size_t len = std::strlen(nm);
        bool hit = false;
        for (size_t pos = 0; pos <= f.size() && !hit;) {
            const size_t bar = std::min(f.find('|', pos), f.size());
            hit = bar - pos == len && f.compare(pos, len, nm) == 0;
            pos = bar + 1;
        }
        if (!hit) return;
    }
    if (!ctx->free_events.empty()) {
        start = ctx->free_events.back().first;
        stop = ctx->free_events.back().second;
        ctx->free_events.pop_back();
    } else {
        if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) { start = stop = nullptr; return; }
    }
    (void)hipEventRecord(start, ctx->stream);
}
KTimer::~KTimer() {
    if (!start) return;
    (void)hipEventRecord(stop, ctx->stream);
    ctx->pending.push_back({name, start, stop});
}

int collect_timings(Ctx *ctx) {
    if (ctx->pending.empty()) return 0;
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &t : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            auto &a = ctx->acc[t.name];
            a.first += 1;
            a.second += ms;
        }
        ctx->free_events.push_back({t.start, t.stop});
    }
    ctx->pending.clear();
    return 0;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

const char *pantax_hip_version(void) { return "pantax-hip 0.1.0 (gfx950)"; }

const char *pantax_hip_last_error(const pantax_hip_ctx *ctx) {
    if (!ctx) return g_init_err.c_str();
    if (g_thread_err_ctx == ctx) return g_thread_err.c_str();   // this thread's own last failure on this ctx
    std::lock_guard<std::mutex> g(const_cast<pantax_hip_ctx *>(ctx)->err_mu);   // another thread's failure: a copy that outlives the lock
    g_thread_err = ctx->err;
    return g_thread_err.c_str();
}

int pantax_hip_init(pantax_hip_ctx **out, const int *device_ids, int n_devices) {
    if (!out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    if (n_devices != 1) return fail(nullptr, PANTAX_HIP_E_INVALID, "one ctx drives one GPU (one process per GPU); n_devices=%d", n_devices);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, PANTAX_HIP_E_NO_DEVICE, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return fail(nullptr, PANTAX_HIP_E_INVALID, "device id %d out of range (count %d)", dev, count);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return fail(nullptr, PANTAX_HIP_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PANTAX_HIP_E_NO_DEVICE, "device %d is %s; this build carries gfx950 code only", dev, prop.gcnArchName);
    if ((e = hipSetDevice(dev)) != hipSuccess) return fail(nullptr, PANTAX_HIP_E_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    pantax_hip_ctx *ctx = new pantax_hip_ctx();
    ctx->device = dev;
    ctx->n_cu = prop.multiProcessorCount;
    // The main stream at the highest priority, the side stream (index rebuild) at the lowest: in a stream of steps the rebuild for
    // step i+1 runs beside the tail of step i, which is the critical chain -- the rebuild has 2 ms of slack and fills what the
    // chain's narrow kernels (sample ranking, the LP workgroups) leave idle instead of taking wave slots from its wide ones
    // (cfg3: 6.80 -> 6.51 ms per step; PANTAX_STREAM_PRIO=0 = equal priorities, for measurements)
    int prio_lo = 0, prio_hi = 0;
    const char *prio_env = std::getenv("PANTAX_STREAM_PRIO");
    const bool prio = !(prio_env && prio_env[0] == '0') && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_lo != prio_hi;
    if ((e = prio ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PANTAX_HIP_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if ((prio ? hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_seq, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PANTAX_HIP_E_HIP, "hipStreamCreate failed");
    }
    if (ctx->d_scalars.alloc(64) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PANTAX_HIP_E_HIP, "hipMalloc failed");
    }
    *out = ctx;
    return 0;
}

void pantax_hip_destroy(pantax_hip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->pending) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto &p : ctx->free_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    ctx->d_scalars.release();
    ctx->d_scan_ws.release();
    ctx->pin_down.release();
    ctx->pin_up.release();
    for (auto &half : ctx->pin_up_ev) for (hipEvent_t &e : half) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    ctx->pin_text.release();
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_seq) (void)hipEventDestroy(ctx->ev_seq);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    dev_cache_trim();   // nothing of this process stays cached on the device once a ctx is gone
}

int pantax_hip_sync(pantax_hip_ctx *ctx) {
    if (!ctx) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int pantax_hip_timing_enable(pantax_hip_ctx *ctx, int on) {
    if (!ctx) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    PTX_TRY(collect_timings(ctx));
    ctx->timing = on != 0;
    return 0;
}

int pantax_hip_timing_filter(pantax_hip_ctx *ctx, const char *name) {
    if (!ctx) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    ctx->timing_filter = name ? name : "";
    return 0;
}

int pantax_hip_timing_reset(pantax_hip_ctx *ctx) {
    if (!ctx) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    PTX_TRY(collect_timings(ctx));
    ctx->acc.clear();
    return 0;
}

int pantax_hip_timing_get(pantax_hip_ctx *ctx, int cap, const char **names_out, uint64_t *launches_out, double *total_ms_out) {
    if (!ctx) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    PTX_TRY(collect_timings(ctx));
    int i = 0;
    for (auto &kv : ctx->acc) {
        if (i < cap) {
            if (names_out) names_out[i] = kv.first.c_str();
            if (launches_out) launches_out[i] = kv.second.first;
            if (total_ms_out) total_ms_out[i] = kv.second.second;
        }
        ++i;
    }
    return i;
}

}  // extern "C"


namespace ptx {

// ---- device allocation cache (see common.hpp) ----
namespace {
struct DevCache {
    struct Block { void *p; uint64_t epoch; };
    struct PerDevice {
        std::multimap<size_t, Block> free_blocks;   // by capacity
        size_t cached_bytes = 0;
        uint64_t free_epoch = 0, synced_epoch = 0;  // a block released at epoch e may be reused once synced_epoch >= e
    };
    std::mutex mu;
    std::map<int, PerDevice> dev;
};
DevCache &dev_cache() { static DevCache c; return c; }
inline size_t round_cap(size_t bytes) {
    if (bytes <= (1u << 20)) return (bytes + 4095) & ~size_t(4095);
    return (bytes + (1u << 20) - 1) & ~size_t((1u << 20) - 1);
}
}  // namespace

hipError_t dev_cache_alloc(void **p, size_t bytes, size_t *cap_out, int *dev_out) {
    const size_t want = round_cap(bytes ? bytes : 1);
    int dev = 0;
    (void)hipGetDevice(&dev);
    *dev_out = dev;
    DevCache &c = dev_cache();
    {
        std::unique_lock<std::mutex> lk(c.mu);
        DevCache::PerDevice &d = c.dev[dev];
        auto it = d.free_blocks.lower_bound(want);
        if (it != d.free_blocks.end() && it->first <= std::max(2 * want, want + (size_t(1) << 20))) {
            const size_t cap = it->first;
            const DevCache::Block b = it->second;
            d.free_blocks.erase(it);
            d.cached_bytes -= cap;
            const bool need_sync = b.epoch > d.synced_epoch;
            const uint64_t now = d.free_epoch;
            lk.unlock();
            if (need_sync) {   // kernels enqueued before the block was released may still be using it
                (void)hipDeviceSynchronize();
                std::lock_guard<std::mutex> g(c.mu);
                DevCache::PerDevice &d2 = c.dev[dev];
                if (now > d2.synced_epoch) d2.synced_epoch = now;
            }
            *p = b.p; *cap_out = cap;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {   // out of memory: give the cached blocks back and retry once
        (void)hipGetLastError();
        dev_cache_trim();
        e = hipMalloc(p, want);
    }
    *cap_out = e == hipSuccess ? want : 0;
    return e;
}

void dev_cache_free(void *p, size_t cap, int dev) {
    if (!p) return;
    DevCache &c = dev_cache();
    {
        std::lock_guard<std::mutex> g(c.mu);
        DevCache::PerDevice &d = c.dev[dev];
        if (cap && d.cached_bytes + cap <= DEV_CACHE_MAX) {
            d.free_blocks.emplace(cap, DevCache::Block{p, ++d.free_epoch});
            d.cached_bytes += cap;
            return;
        }
    }
    (void)hipFree(p);
}

void dev_cache_trim() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    DevCache &c = dev_cache();
    std::multimap<size_t, DevCache::Block> blocks;
    {
        std::lock_guard<std::mutex> g(c.mu);
        DevCache::PerDevice &d = c.dev[dev];
        blocks.swap(d.free_blocks);
        d.cached_bytes = 0;
    }
    for (auto &kv : blocks) (void)hipFree(kv.second.p);
}

namespace {
// [src + off, +n) or (fd, file_off + off, n) -> pinned slot, on a few threads (about 2 MB each: the copies are memory-bound)
void stage_chunk(uint8_t *slot, const uint8_t *src, int fd, uint64_t pos, uint64_t n) {
    constexpr int NTH_MAX = 64;
    static const int NTH_ENV = std::getenv("PANTAX_STAGE_THREADS") ? std::atoi(std::getenv("PANTAX_STAGE_THREADS")) : 16;
    static const int PIECE_SHIFT = std::getenv("PANTAX_STAGE_PIECE_KB") ? 10 + (int)std::log2((double)std::max(64, std::atoi(std::getenv("PANTAX_STAGE_PIECE_KB")))) : 21;
    static const int NTH_HW = (int)std::max(2u, std::min<unsigned>((unsigned)std::min(NTH_MAX, std::max(1, NTH_ENV)), std::thread::hardware_concurrency() / 2));
    const int nth = (int)std::max<uint64_t>(1, std::min<uint64_t>(NTH_HW, n >> PIECE_SHIFT));
    auto piece = [=](int t) {
        const uint64_t b = n * t / nth, e = n * (t + 1) / nth;
        if (src) { std::memcpy(slot + b, src + pos + b, e - b); return; }
        uint64_t done = b;
        while (done < e) {   // pread may return short
            const ssize_t r = ::pread(fd, slot + done, e - done, (off_t)(pos + done));
            if (r <= 0) { std::memset(slot + done, 0, e - done); break; }   // truncated file: the caller validated sizes, zeros keep the run defined
            done += (uint64_t)r;
        }
    };
    std::thread th[NTH_MAX];
    for (int t = 1; t < nth; ++t) th[t] = std::thread(piece, t);
    piece(0);
    for (int t = 1; t < nth; ++t) th[t].join();
}

int upload_staged(Ctx *ctx, void *d_dst, const void *src, int fd, uint64_t file_off, uint64_t size) {
    if (size == 0) return 0;
    static const uint64_t CH = (uint64_t)(std::getenv("PANTAX_STAGE_CH_MB") ? std::max(1, std::atoi(std::getenv("PANTAX_STAGE_CH_MB"))) : 16) << 20;
    if (src && size < (1ull << 20)) {   // small: not worth the staging
        PTX_HIP(ctx, hipMemcpyAsync(d_dst, src, size, hipMemcpyHostToDevice, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    PTX_HIP(ctx, ctx->pin_text.reserve(2 * CH));
    hipEvent_t ev[2] = {nullptr, nullptr};
    PTX_HIP(ctx, hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    PTX_HIP(ctx, hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int rc = 0;
    uint64_t i = 0;
    uint8_t *dst = static_cast<uint8_t *>(d_dst);
    for (uint64_t off = 0; off < size && rc == 0; off += CH, ++i) {
        const uint64_t n = std::min<uint64_t>(CH, size - off);
        uint8_t *slot = ctx->pin_text.p + (i & 1) * CH;
        if (i >= 2 && hipEventSynchronize(ev[i & 1]) != hipSuccess) { rc = fail(ctx, PANTAX_HIP_E_HIP, "hipEventSynchronize failed"); break; }
        stage_chunk(slot, static_cast<const uint8_t *>(src), fd, (src ? 0 : file_off) + off, n);
        if (hipMemcpyAsync(dst + off, slot, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipEventRecord(ev[i & 1], ctx->stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "upload failed");
    }
    if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "upload failed");
    (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
    return rc;
}
}  // namespace

int upload_big(Ctx *ctx, void *d_dst, const void *src, uint64_t size) { return upload_staged(ctx, d_dst, src, -1, 0, size); }
int upload_file(Ctx *ctx, void *d_dst, int fd, uint64_t file_off, uint64_t size) { return upload_staged(ctx, d_dst, nullptr, fd, file_off, size); }

void parallel_for(uint64_t n, int n_threads, const std::function<void(uint64_t, uint64_t)> &fn) {
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n) n_threads = n ? (int)n : 1;
    if (n_threads == 1) { fn(0, n); return; }
    // an exception on a worker thread would end the process (std::terminate): it is carried to the calling thread instead,
    // where the extern "C" entry points turn it into a status
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(n_threads);
    for (int t = 1; t < n_threads; ++t)
        th.emplace_back([&, t] {
            try { fn(n * t / n_threads, n * (t + 1) / n_threads); } catch (...) { err[t] = std::current_exception(); }
        });
    try { fn(0, n / n_threads); } catch (...) { err[0] = std::current_exception(); }
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

}  // namespace ptx
