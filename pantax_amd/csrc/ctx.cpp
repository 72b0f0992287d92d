// ctx.cpp -- context lifetime, error strings, HIP-event kernel timing (include/pantax_hip.h).
#include <cstdarg>
#include <cstring>
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
    if (ctx) { ctx->err = buf; g_thread_err = buf; g_thread_err_ctx = ctx; }
    else g_init_err = buf;
    return code;
}

KTimer::KTimer(Ctx *c, const char *nm) : ctx(c), name(nm) {
    if (!ctx->timing) return;
    if (!ctx->timing_filter.empty() && ctx->timing_filter != nm) return;   // only the named launch is bracketed
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
    return ctx->err.c_str();
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
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PANTAX_HIP_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) {
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
    ctx->pin_text.release();
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
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
