// ctx.cpp -- context lifetime, error strings, HIP-event kernel timing (include/pantax_hip.h).
#include <exception>
#include <cstdarg>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <fcntl.h>
#include <sched.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <fstream>
#include <cctype>
#include <cstddef>
#include <cstdlib>
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

// ---- options: one table, filled from the environment at init, changed through pantax_hip_set_option ----
namespace {
enum OptKind { O_BOOL, O_INT, O_U32, O_U64, O_STR };
struct OptDesc { const char *name; OptKind kind; size_t off; };
#define OPT(nm, kind, field) {nm, kind, offsetof(CtxConfig, field)}
const OptDesc OPTIONS[] = {
    OPT("hip_trace", O_BOOL, trace), OPT("stage_threads", O_INT, stage_threads), OPT("stage_ch_mb", O_INT, stage_ch_mb), OPT("stream_prio", O_BOOL, stream_prio), OPT("numa_bind", O_BOOL, numa_bind), OPT("dev_cache_gb", O_INT, dev_cache_gb),
    OPT("gaf_piece_bytes", O_U64, gaf_piece_bytes), OPT("db_path_steps_max", O_U64, db_path_steps_max), OPT("db_groups", O_INT, db_groups), OPT("trio_path", O_STR, trio_path), OPT("trio_rows", O_STR, trio_rows), OPT("trio_two_pass", O_BOOL, trio_two_pass), OPT("uniq_hash", O_INT, uniq_hash),
    OPT("mask", O_STR, mask), OPT("row_sort", O_STR, row_sort), OPT("objective", O_STR, objective),
    OPT("cov_general", O_BOOL, cov_general), OPT("cov_long", O_STR, cov_long), OPT("covl_shape", O_INT, covl_shape), OPT("cov_count", O_BOOL, cov_count), OPT("cov_self_clean", O_BOOL, cov_self_clean), OPT("cov_clean_async", O_INT, cov_clean_async), OPT("cov_arena_verify", O_BOOL, cov_arena_verify), OPT("ncs_no_prefix", O_BOOL, ncs_no_prefix), OPT("ncs_prefix_min", O_INT, ncs_prefix_min), OPT("walk_sum_in_bin", O_BOOL, walk_sum_in_bin), OPT("cov_item_groups", O_INT, cov_item_groups), OPT("tv_u", O_INT, tv_u), OPT("tv_rounds", O_INT, tv_rounds), OPT("tf_u", O_INT, tf_u), OPT("tf_rounds", O_INT, tf_rounds),
    OPT("rows_u", O_INT, rows_u), OPT("tb_slots", O_INT, tb_slots), OPT("trio_xcd", O_INT, trio_xcd), OPT("cov_shape", O_INT, cov_shape),
    OPT("covf_shape", O_INT, covf_shape), OPT("cov_xcd", O_INT, cov_xcd), OPT("group_bucket_bits", O_INT, group_bucket_bits), OPT("tv_ablate", O_U32, tv_ablate),
    OPT("cov_ablate", O_U32, cov_ablate), OPT("ssn_ablate", O_U32, ssn_ablate), OPT("no_absent_skip", O_BOOL, no_absent_skip), OPT("ssn_debug", O_BOOL, ssn_debug),
    OPT("scan_no_huge", O_BOOL, scan_no_huge), OPT("flag_rank_chained", O_BOOL, flag_rank_chained), OPT("ratio_kernel", O_BOOL, ratio_kernel),
    OPT("mask_pass", O_BOOL, mask_pass), OPT("trio_free_at_filter", O_BOOL, trio_free_at_filter), OPT("trio_after_step", O_BOOL, trio_after_step),
};
#undef OPT
}  // namespace

int ctx_set_option(CtxConfig &cfg, const char *name, const char *value) {
    if (!name) return PANTAX_HIP_E_INVALID;
    static const CtxConfig defaults;
    for (const OptDesc &o : OPTIONS) {
        if (std::strcmp(o.name, name) != 0) continue;
        char *dst = reinterpret_cast<char *>(&cfg) + o.off;
        const char *def = reinterpret_cast<const char *>(&defaults) + o.off;
        char *end = nullptr;
        switch (o.kind) {
        case O_BOOL: *reinterpret_cast<bool *>(dst) = value ? !(value[0] == '0' || value[0] == '\0' || value[0] == 'n' || value[0] == 'f') : *reinterpret_cast<const bool *>(def); return 0;
        case O_INT: { if (!value) { *reinterpret_cast<int *>(dst) = *reinterpret_cast<const int *>(def); return 0; }
                      const long v = std::strtol(value, &end, 10); if (end == value) return PANTAX_HIP_E_INVALID; *reinterpret_cast<int *>(dst) = (int)v; return 0; }
        case O_U32: { if (!value) { *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(def); return 0; }
                      const unsigned long long v = std::strtoull(value, &end, 10); if (end == value) return PANTAX_HIP_E_INVALID; *reinterpret_cast<uint32_t *>(dst) = (uint32_t)v; return 0; }
        case O_U64: { if (!value) { *reinterpret_cast<uint64_t *>(dst) = *reinterpret_cast<const uint64_t *>(def); return 0; }
                      const unsigned long long v = std::strtoull(value, &end, 10); if (end == value) return PANTAX_HIP_E_INVALID; *reinterpret_cast<uint64_t *>(dst) = (uint64_t)v; return 0; }
        case O_STR: *reinterpret_cast<std::string *>(dst) = value ? value : ""; return 0;
        }
    }
    return PANTAX_HIP_E_INVALID;
}

// the environment, once: PANTAX_<NAME> for every option of the table
static void config_from_env(CtxConfig &cfg) {
    for (const OptDesc &o : OPTIONS) {
        std::string env = "PANTAX_";
        for (const char *p = o.name; *p; ++p) env += (char)std::toupper((unsigned char)*p);
        if (const char *v = std::getenv(env.c_str())) (void)ctx_set_option(cfg, o.name, v);
    }
}

}  // namespace ptx

using namespace ptx;

extern "C" {

const char *pantax_hip_version(void) { return "pantax-hip 0.1.0 (gfx950)"; }

int pantax_hip_set_option(pantax_hip_ctx *ctx, const char *name, const char *value) {
    if (!ctx || !name) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    const int rc = ctx_set_option(ctx->cfg, name, value);
    if (rc == 0 && std::strcmp(name, "dev_cache_gb") == 0) dev_cache_set_max(ctx->cfg.dev_cache_gb < 0 ? -1 : (long long)ctx->cfg.dev_cache_gb << 30);
    return rc == 0 ? 0 : fail(ctx, rc, "set_option: unknown option or unparsable value: %s=%s", name, value ? value : "(default)");
}

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
    config_from_env(ctx->cfg);   // the only place the library reads the environment
    if (ctx->cfg.dev_cache_gb >= 0) dev_cache_set_max((long long)ctx->cfg.dev_cache_gb << 30);
    {   // the GPU's NUMA node and its CPUs (sysfs); anything missing = nothing is bound
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), dev) == hipSuccess) {
            for (char *c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
            std::ifstream f(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
            int node = -1;
            if (f >> node && node >= 0) {
                std::ifstream g("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
                std::string list;
                if (std::getline(g, list)) {
                    for (size_t pos = 0; pos < list.size();) {
                        const size_t comma = std::min(list.find(',', pos), list.size());
                        const std::string part = list.substr(pos, comma - pos);
                        const size_t dash = part.find('-');
                        const int a = std::atoi(part.c_str()), b = dash == std::string::npos ? a : std::atoi(part.c_str() + dash + 1);
                        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) ctx->numa.cpus.push_back(c);
                        pos = comma + 1;
                    }
                    if (!ctx->numa.cpus.empty()) ctx->numa.node = node;
                }
            }
        }
    }
    // The main stream at the highest priority, the side stream (index rebuild) at the lowest: in a stream of steps the rebuild for
    // step i+1 runs beside the tail of step i, which is the critical chain -- the rebuild has 2 ms of slack and fills what the
    // chain's narrow kernels (sample ranking, the LP workgroups) leave idle instead of taking wave slots from its wide ones
    // (cfg3: 6.80 -> 6.51 ms per step; option stream_prio=0 = equal priorities, for measurements)
    int prio_lo = 0, prio_hi = 0;
    const bool prio = ctx->cfg.stream_prio && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_lo != prio_hi;
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
    dev_cache_register_stream(dev, ctx->stream, true);
    dev_cache_register_stream(dev, ctx->stream2, true);
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
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); dev_cache_register_stream(ctx->device, ctx->stream2, false); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->stream_up) { (void)hipStreamSynchronize(ctx->stream_up); (void)hipStreamDestroy(ctx->stream_up); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_seq) (void)hipEventDestroy(ctx->ev_seq);
    dev_cache_register_stream(ctx->device, ctx->stream, false);
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
        std::vector<hipStream_t> streams;           // the compute streams of the ctx's on this device: what "the device is idle" means for a cached block
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

// The cap is PER DEVICE and sized from what is free when the device is first used: min(3/4 of the device, 9/10 of what was free) -- a process
// that shares its GPU (N ranks of a dry run on one device, parallel test workers) must not sit on memory its siblings need.  `dev_cache_gb`
// (option / pantax_hip_set_option) overrides it for every device of the process; 0 caches nothing.
static std::atomic<long long> g_dev_cache_cap_override{-1};
void dev_cache_set_max(long long bytes) { g_dev_cache_cap_override.store(bytes); }
size_t dev_cache_max(int dev) {
    const long long ov = g_dev_cache_cap_override.load();
    if (ov >= 0) return (size_t)ov;
    static std::mutex mu;
    static std::map<int, size_t> caps;
    std::lock_guard<std::mutex> g(mu);
    auto it = caps.find(dev);
    if (it != caps.end()) return it->second;
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t free_b = 0, total_b = 0, cap = (size_t)(48ull << 30);
    if (cur != dev) (void)hipSetDevice(dev);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b != 0) cap = std::min(total_b / 4 * 3, free_b / 10 * 9);
    if (cur != dev) (void)hipSetDevice(cur);
    caps[dev] = cap;
    return cap;
}
// a sibling process is short of memory when less than an eighth of the device is free: big blocks are then given back instead of kept
static bool dev_memory_is_tight() {
    size_t free_b = 0, total_b = 0;
    return hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b != 0 && free_b < total_b / 8;
}

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
            const std::vector<hipStream_t> streams = d.streams;
            lk.unlock();
            if (need_sync) {   // kernels enqueued before the block was released may still be using it
                // every kernel of this library runs on a ctx's main or side stream (copy streams -- the GAF upload's, the graph loader's -- are waited
                // for by their owners before they release anything): those are waited for, not the whole device -- a device-wide wait also sits
                // out the graph loader's transfers of the NEXT group of species (round 6)
                if (streams.empty()) (void)hipDeviceSynchronize();
                else for (hipStream_t st : streams) (void)hipStreamSynchronize(st);
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

// blocks released inside such a scope are known to be idle (the caller has waited for every stream that could have used them): they may be handed out
// again without the device-wide wait -- which would also wait for whatever ELSE is in flight, e.g. the graph loader's copy stream (round 6: that wait
// cost every group of the file seam ~4 ms per stage that allocated)
static thread_local int g_frees_are_idle = 0;
DevCacheIdleFrees::DevCacheIdleFrees() { ++g_frees_are_idle; }
DevCacheIdleFrees::~DevCacheIdleFrees() { --g_frees_are_idle; }

void dev_cache_free(void *p, size_t cap, int dev) {
    if (!p) return;
    DevCache &c = dev_cache();
    {
        std::lock_guard<std::mutex> g(c.mu);
        DevCache::PerDevice &d = c.dev[dev];
        // (the memory query costs microseconds: only blocks of 64 MB and more pay it)
        if (cap && d.cached_bytes + cap <= dev_cache_max(dev) && !(cap >= (size_t(64) << 20) && dev_memory_is_tight())) {
            d.free_blocks.emplace(cap, DevCache::Block{p, g_frees_are_idle ? 0 : ++d.free_epoch});
            d.cached_bytes += cap;
            return;
        }
    }
    (void)hipFree(p);
}

void dev_cache_register_stream(int dev, hipStream_t st, bool add) {
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> g(c.mu);
    auto &v = c.dev[dev].streams;
    v.erase(std::remove(v.begin(), v.end(), st), v.end());
    if (add) v.push_back(st);
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
// A crew of host threads that fills pinned slots -- [src + pos, +n) or (fd, pos, n) -- for the lifetime of ONE upload (round 2
// created and joined 16 threads per 16-MB chunk: ~0.5 ms of thread start-up beside 0.6 ms of copy).  The box delivers 80 GB/s
// of pread from the page cache on 8-16 threads and 56 GB/s of pinned host->device copy (tools/h2d_probe.py): the crew only has
// to stay ahead of the DMA (it takes 32 threads for that inside the pipeline: see upload_staged_pieces).
// bind the calling thread to a NUMA node's CPUs (and, while `prefer` is set, its new pages to that node: the pinned ring is allocated under
// it); restores what was there when it goes out of scope
struct NumaBind {
    cpu_set_t saved;
    bool bound = false, policy = false;
    NumaBind(const Ctx *ctx, bool prefer) {
        if (!ctx->cfg.numa_bind || ctx->numa.node < 0 || ctx->numa.cpus.empty()) return;
        if (sched_getaffinity(0, sizeof(saved), &saved) != 0) return;
        cpu_set_t want;
        CPU_ZERO(&want);
        int n = 0;
        for (int c : ctx->numa.cpus) if (CPU_ISSET(c, &saved)) { CPU_SET(c, &want); ++n; }   // only CPUs this process may use (cgroups / taskset)
        if (n == 0) return;
        bound = sched_setaffinity(0, sizeof(want), &want) == 0;
        if (bound && prefer && ctx->numa.node < 1024) {
#ifdef SYS_set_mempolicy
            unsigned long mask[16] = {0};
            mask[ctx->numa.node / 64] = 1ul << (ctx->numa.node % 64);
            policy = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 1024ul) == 0;
#endif
        }
    }
    ~NumaBind() {
#ifdef SYS_set_mempolicy
        if (policy) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
#endif
        if (bound) (void)sched_setaffinity(0, sizeof(saved), &saved);
    }
};

struct StageCrew {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int done = 0, nth = 1;
    bool quit = false;
    uint8_t *slot = nullptr; const uint8_t *src = nullptr; int fd = -1; uint64_t pos = 0, n = 0;
    const std::function<void(int, int)> *job = nullptr;   // run(): any per-thread job instead of the copy
    std::vector<std::thread> th;
    static void piece(uint8_t *slot, const uint8_t *src, int fd, uint64_t pos, uint64_t n, int t, int nth) {
        const uint64_t b = n * (uint64_t)t / (uint64_t)nth, e = n * (uint64_t)(t + 1) / (uint64_t)nth;
        if (src) { std::memcpy(slot + b, src + pos + b, e - b); return; }
        uint64_t at = b;
        while (at < e) {   // pread may return short
            const ssize_t r = ::pread(fd, slot + at, e - at, (off_t)(pos + at));
            if (r <= 0) { std::memset(slot + at, 0, e - at); break; }   // truncated file: the caller validated sizes, zeros keep the run defined
            at += (uint64_t)r;
        }
    }
    explicit StageCrew(int n_threads) : nth(n_threads < 1 ? 1 : n_threads) {
        for (int t = 1; t < nth; ++t)
            th.emplace_back([this, t] {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_go.wait(lk, [&] { return quit || gen != seen; });
                    if (quit) return;
                    seen = gen;
                    uint8_t *sl = slot; const uint8_t *sr = src; const int f = fd; const uint64_t p = pos, nn = n;
                    const std::function<void(int, int)> *jb = job;
                    lk.unlock();
                    if (jb) (*jb)(t, nth); else piece(sl, sr, f, p, nn, t, nth);
                    lk.lock();
                    if (++done == nth - 1) cv_done.notify_one();
                }
            });
    }
    ~StageCrew() {
        { std::lock_guard<std::mutex> g(mu); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
    void run(const std::function<void(int, int)> &fn) {   // fn(t, nth) on every thread of the crew; returns when all are done
        { std::lock_guard<std::mutex> g(mu); job = &fn; done = 0; ++gen; }
        cv_go.notify_all();
        fn(0, nth);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done == nth - 1; });
        job = nullptr;
    }
    void fill(uint8_t *slot_, const uint8_t *src_, int fd_, uint64_t pos_, uint64_t n_) {
        { std::lock_guard<std::mutex> g(mu); job = nullptr; slot = slot_; src = src_; fd = fd_; pos = pos_; n = n_; done = 0; ++gen; }
        cv_go.notify_all();
        piece(slot_, src_, fd_, pos_, n_, 0, nth);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done == nth - 1; });
    }
};

// host memory or a file -> HBM through a ring of pinned chunks: chunk i+1 (and i+2) is filled while chunk i is on its way.
// Chunk size follows the transfer (a sixth of it, 4..64 MB: pinning memory costs ~0.1 ms per MB, once per ctx); `stream`
// carries the copies.  The transfer may consist of several PIECES with their own destinations (the GAF load: a piece is
// tokenised while the next ones travel): the chunk pipeline runs across the piece borders without draining; before_piece(k)
// may block until piece k's destination is free (false aborts), after_piece(k) is called once the last chunk of piece k
// has been enqueued (the caller records an event there).
struct UploadPiece { void *d_dst; const uint8_t *src; uint64_t file_off, size; };
int upload_staged_pieces(Ctx *ctx, const UploadPiece *pieces, size_t n_pieces, int fd, hipStream_t stream, PinBuf &ring,
                         const std::function<bool(size_t)> &before_piece, const std::function<int(size_t)> &after_piece) {
    uint64_t total = 0;
    for (size_t k = 0; k < n_pieces; ++k) total += pieces[k].size;
    constexpr int SLOTS = 3;
    uint64_t CH = ctx->cfg.stage_ch_mb > 0 ? (uint64_t)ctx->cfg.stage_ch_mb << 20
                                           : std::min<uint64_t>(64ull << 20, std::max<uint64_t>(4ull << 20, ((total / 6) + (1 << 20) - 1) & ~(uint64_t)((1 << 20) - 1)));
    if (ring.n >= SLOTS * (4ull << 20) && ring.n / SLOTS > CH && ctx->cfg.stage_ch_mb <= 0) CH = std::min<uint64_t>(64ull << 20, (ring.n / SLOTS) & ~(uint64_t)((1 << 20) - 1));   // a larger ring is there already
    NumaBind numa_bind(ctx, true);   // the ring's pages, this thread and the crew it starts: on the GPU's NUMA node until the upload is over
    PTX_HIP(ctx, ring.reserve(SLOTS * CH));
    // 32 threads: at 16 the crew, not the DMA, bounds a 15-GB load (filling 413 of 420 ms = 37 GB/s of pread; 32: 211 of 293 ms = 52 GB/s,
    // 0.92 of the pinned copy rate); 48 and 64 fill no faster and slow the copies down (390 ms)
    const int nth = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, std::min(ctx->cfg.stage_threads, (int)std::thread::hardware_concurrency() / 2)), CH >> 20));
    StageCrew crew(nth);
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr};
    for (auto &e : ev) PTX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int rc = 0;
    uint64_t i = 0;
    const bool trace = ctx->cfg.trace;
    double t_wait = 0, t_fill = 0, t_enq = 0, t_gate = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_begin = now();
    for (size_t pk = 0; pk < n_pieces && rc == 0; ++pk) {
        const UploadPiece &pc = pieces[pk];
        const auto tg = now();
        if (before_piece && !before_piece(pk)) { rc = PANTAX_HIP_E_STATE; break; }
        t_gate += ms(tg, now());
        uint8_t *dst = static_cast<uint8_t *>(pc.d_dst);
        for (uint64_t off = 0; off < pc.size && rc == 0; off += CH, ++i) {
            const uint64_t n = std::min<uint64_t>(CH, pc.size - off);
            const int k = (int)(i % SLOTS);
            uint8_t *slot = ring.p + (uint64_t)k * CH;
            const auto t0 = now();
            if (i >= SLOTS && hipEventSynchronize(ev[k]) != hipSuccess) { rc = fail(ctx, PANTAX_HIP_E_HIP, "hipEventSynchronize failed"); break; }
            const auto t1 = now();
            crew.fill(slot, pc.src, fd, (pc.src ? 0 : pc.file_off) + off, n);
            const auto t2 = now();
            if (hipMemcpyAsync(dst + off, slot, n, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(ev[k], stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "upload failed");
            if (trace) { const auto t3 = now(); t_wait += ms(t0, t1); t_fill += ms(t1, t2); t_enq += ms(t2, t3); }
        }
        if (rc == 0 && after_piece) rc = after_piece(pk);
    }
    const auto t4 = now();
    if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "upload failed");
    if (trace)
        std::fprintf(stderr, "[upload_staged] %.1f MB in %.2f ms (%.1f GB/s): %zu piece(s), %llu chunks of %.0f MB, %d threads; waiting for a piece buffer %.2f, for a slot %.2f, "
                     "filling %.2f, enqueue %.2f, drain %.2f ms; NUMA node %d (%s)\n", total / 1e6, ms(t_begin, now()), total / 1e6 / ms(t_begin, now()), n_pieces,
                     (unsigned long long)i, CH / 1048576.0, nth, t_gate, t_wait, t_fill, t_enq, ms(t4, now()), ctx->numa.node, numa_bind.bound ? "bound" : "not bound");
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

int upload_staged(Ctx *ctx, void *d_dst, const void *src, int fd, uint64_t file_off, uint64_t size, hipStream_t stream, PinBuf &ring) {
    if (size == 0) return 0;
    if (src && size < (1ull << 20)) {   // small: not worth the staging
        PTX_HIP(ctx, hipMemcpyAsync(d_dst, src, size, hipMemcpyHostToDevice, stream));
        PTX_HIP(ctx, hipStreamSynchronize(stream));
        return 0;
    }
    const UploadPiece pc{d_dst, static_cast<const uint8_t *>(src), file_off, size};
    return upload_staged_pieces(ctx, &pc, 1, fd, stream, ring, nullptr, nullptr);
}
}  // namespace

// ---- many stretches (files / memory, as they are or narrowed from 64 to 32 bits) -> ONE contiguous device range ----
// The db load of the pipeline seam (a6): 1 000 species' graphs are 2 000 stretches of a few megabytes in as many files.  One upload per
// stretch (round 4) created a crew of threads and drained the copy queue 2 000 times; here the stretches are one logical byte string
// that travels in 64-MB chunks like the GAF text: every thread of the crew fills its share of the chunk -- whatever stretches it spans --
// and the chunk leaves as one DMA.  Files are opened by the thread that needs them (one descriptor per thread at a time: no limit on
// the number of species), bincode's 64-bit integers are narrowed on the way into the pinned chunk (read in 64-KB blocks: L2-resident).
namespace {
struct SegFiller {
    const UploadSeg *segs; size_t n_segs; const std::string *files;
    std::vector<uint64_t> prefix;            // [n_segs + 1] device byte offsets
    std::atomic<int64_t> bad{-1}, io_fail{-1};
    void note(std::atomic<int64_t> &a, int64_t k) { int64_t cur = a.load(); while ((cur < 0 || k < cur) && !a.compare_exchange_weak(cur, k)) {} }
    // device bytes [b, e) of the logical string -> out (which points at byte b)
    void fill(uint8_t *out, uint64_t b, uint64_t e) {
        size_t k = (size_t)(std::upper_bound(prefix.begin(), prefix.end(), b) - prefix.begin()) - 1;
        int fd = -1, fd_file = -1;
        std::vector<uint64_t> tmp;
        for (; b < e && k < n_segs; ++k) {
            const UploadSeg &s = segs[k];
            const uint64_t so = b - prefix[k], n = std::min(e, prefix[k + 1]) - b;
            if (n == 0) continue;
            if (s.hole) { out += n; b += n; continue; }
            const uint8_t *src = static_cast<const uint8_t *>(s.src);
            if (!src && s.file != fd_file) {
                if (fd >= 0) ::close(fd);
                fd = ::open(files[s.file].c_str(), O_RDONLY);
                fd_file = s.file;
                if (fd < 0) note(io_fail, (int64_t)k);
            }
            auto read_at = [&](uint8_t *dst, uint64_t off, uint64_t len) {
                if (src) { std::memcpy(dst, src + off, len); return; }
                uint64_t at = 0;
                while (fd >= 0 && at < len) {
                    const ssize_t r = ::pread(fd, dst + at, len - at, (off_t)(s.file_off + off + at));
                    if (r <= 0) break;
                    at += (uint64_t)r;
                }
                if (at < len) { std::memset(dst + at, 0, len - at); note(io_fail, (int64_t)k); }   // zeros keep the run defined; the caller fails
            };
            if (!s.narrow) read_at(out, so, n);
            else {
                constexpr uint64_t BLK = 131072;                      // 64-bit values per block: a megabyte per pread
                uint32_t *o32 = reinterpret_cast<uint32_t *>(out);
                uint64_t hi = 0;
                for (uint64_t i0 = so / 4, i1 = (so + n) / 4; i0 < i1; i0 += BLK) {
                    const uint64_t m = std::min(BLK, i1 - i0);
                    const uint64_t *in;
                    if (src) in = reinterpret_cast<const uint64_t *>(src) + i0;
                    else { tmp.resize(BLK); read_at(reinterpret_cast<uint8_t *>(tmp.data()), 8 * i0, 8 * m); in = tmp.data(); }
                    uint32_t *o = o32 + (i0 - so / 4);
                    for (uint64_t i = 0; i < m; ++i) { const uint64_t x = in[i]; o[i] = (uint32_t)x; hi |= x; }
                }
                if (hi >> 32) note(bad, (int64_t)k);
            }
            out += n; b += n;
        }
        if (fd >= 0) ::close(fd);
    }
};
}  // namespace

int upload_segments(Ctx *ctx, void *d_dst, const UploadSeg *segs, size_t n_segs, const std::string *files, int64_t *bad_seg, hipStream_t stream_arg) {
    const hipStream_t stream = stream_arg ? stream_arg : ctx->stream;
    if (bad_seg) *bad_seg = -1;
    SegFiller f{segs, n_segs, files};
    f.prefix.assign(n_segs + 1, 0);
    for (size_t k = 0; k < n_segs; ++k) {
        if (segs[k].out_bytes % 4) return fail(ctx, PANTAX_HIP_E_INVALID, "upload_segments: stretch %zu is not a whole number of 32-bit words", k);
        f.prefix[k + 1] = f.prefix[k] + segs[k].out_bytes;
    }
    const uint64_t total = f.prefix[n_segs];
    if (total == 0) return 0;
    constexpr int SLOTS = 3;
    PinBuf &ring = ctx->pin_text;
    uint64_t CH = std::min<uint64_t>(64ull << 20, std::max<uint64_t>(4ull << 20, ((total / 6) + (1 << 20) - 1) & ~(uint64_t)((1 << 20) - 1)));
    if (ring.n >= SLOTS * (4ull << 20) && ring.n / SLOTS > CH) CH = std::min<uint64_t>(64ull << 20, (ring.n / SLOTS) & ~(uint64_t)((1 << 20) - 1));   // a larger ring is there already
    NumaBind numa_bind(ctx, true);   // the ring's pages, this thread and the crew it starts: on the GPU's NUMA node until the upload is over
    PTX_HIP(ctx, ring.reserve(SLOTS * CH));
    const int nth = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, std::min(ctx->cfg.stage_threads, (int)std::thread::hardware_concurrency() / 2)), CH >> 20));
    StageCrew crew(nth);
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr};
    for (auto &e : ev) PTX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int rc = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    double t_fill = 0, t_wait = 0;
    uint64_t i = 0;
    for (uint64_t c0 = 0; c0 < total && rc == 0; c0 += CH, ++i) {
        const uint64_t c1 = std::min(total, c0 + CH);
        const int k = (int)(i % SLOTS);
        uint8_t *slot = ring.p + (uint64_t)k * CH;
        const auto t0 = std::chrono::steady_clock::now();
        if (i >= SLOTS && hipEventSynchronize(ev[k]) != hipSuccess) { rc = fail(ctx, PANTAX_HIP_E_HIP, "hipEventSynchronize failed"); break; }
        const auto t1 = std::chrono::steady_clock::now();
        const std::function<void(int, int)> job = [&](int t, int n) {
            const uint64_t words = (c1 - c0) / 4;
            const uint64_t b = c0 + 4 * (words * (uint64_t)t / (uint64_t)n), e = c0 + 4 * (words * (uint64_t)(t + 1) / (uint64_t)n);
            if (e > b) f.fill(slot + (b - c0), b, e);
        };
        crew.run(job);
        const auto t2 = std::chrono::steady_clock::now();
        t_wait += std::chrono::duration<double, std::milli>(t1 - t0).count(); t_fill += std::chrono::duration<double, std::milli>(t2 - t1).count();
        if (hipMemcpyAsync(static_cast<uint8_t *>(d_dst) + c0, slot, c1 - c0, hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipEventRecord(ev[k], stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "upload failed");
    }
    if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "upload failed");
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (ctx->cfg.trace) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        std::fprintf(stderr, "[upload_segments] %.1f MB in %.2f ms (%.1f GB/s): %zu stretches, %llu chunks of %.0f MB, %d threads; waiting for a slot %.2f, filling %.2f ms\n",
                     total / 1e6, ms, total / 1e6 / ms, n_segs, (unsigned long long)i, CH / 1048576.0, nth, t_wait, t_fill);
    }
    if (rc) return rc;
    if (f.io_fail.load() >= 0) {
        const UploadSeg &s = segs[f.io_fail.load()];
        return fail(ctx, PANTAX_HIP_E_IO, "cannot read %s (bytes from %llu)", s.file >= 0 ? files[s.file].c_str() : "<memory>", (unsigned long long)s.file_off);
    }
    if (bad_seg) *bad_seg = f.bad.load();
    return 0;
}

int upload_big(Ctx *ctx, void *d_dst, const void *src, uint64_t size) { return upload_staged(ctx, d_dst, src, -1, 0, size, ctx->stream, ctx->pin_text); }
int upload_file(Ctx *ctx, void *d_dst, int fd, uint64_t file_off, uint64_t size) { return upload_staged(ctx, d_dst, nullptr, fd, file_off, size, ctx->stream, ctx->pin_text); }
int upload_text_pieces(Ctx *ctx, size_t n_pieces, void *const *d_dst, const char *text, int fd, uint64_t file_base, const uint64_t *piece_off, const uint64_t *piece_end,
                       hipStream_t stream, const std::function<bool(size_t)> &before_piece, const std::function<int(size_t)> &after_piece) {
    std::vector<UploadPiece> pcs(n_pieces);
    for (size_t k = 0; k < n_pieces; ++k)
        pcs[k] = UploadPiece{d_dst[k], fd >= 0 ? nullptr : reinterpret_cast<const uint8_t *>(text) + piece_off[k], file_base + piece_off[k], piece_end[k] - piece_off[k]};
    return upload_staged_pieces(ctx, pcs.data(), n_pieces, fd, stream, ctx->pin_text, before_piece, after_piece);
}

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
