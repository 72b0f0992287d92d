// common.hpp -- internal state behind the opaque handles of include/pantax_hip.h.
// gfx950 only; no CPU fallback anywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <functional>
#include <mutex>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/pantax_hip.h"

#ifndef TRIO_LH_PACK
#define TRIO_LH_PACK 0   // 1: length and owner of a row in ONE 8-byte record (measurement builds)
#endif
namespace ptx {

struct Ctx;
#if TRIO_LH_PACK
using trio_len_t = uint2;
#else
using trio_len_t = uint32_t;
#endif

// ---- error plumbing: HIP errors become PANTAX_HIP_E_HIP + message, never abort -------------
int fail(Ctx *ctx, int code, const char *fmt, ...);
#define PTX_HIP(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return ptx::fail((ctx), PANTAX_HIP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                                     \
    } while (0)
// Entry of every C-ABI function that takes a ctx: calls on one ctx from several host threads (the reference calls its
// solver from rayon workers, profile.rs:3297-3304) are serialised; the lock is recursive because entry points call
// each other.  Also selects the ctx's device for the calling thread.
#define PTX_ENTER(ctx)                                            \
    std::lock_guard<std::recursive_mutex> ptx_lock__((ctx)->mu); \
    PTX_HIP(ctx, hipSetDevice((ctx)->device))
#define PTX_TRY(expr)               \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != 0) return rc__; \
    } while (0)

// ---- device buffer ---------------------------------------------------------------------------
// Device allocations go through a small per-process cache: hipFree of a large buffer costs around a millisecond (it
// waits for the device and unmaps), and the file seam allocates and releases some dozens of them per call.  A released
// block is kept (up to dev_cache_max() bytes in all) and handed to the next request of a similar size; before a block
// freed since the last device-wide wait is reused, the device is synchronised once -- the guarantee hipFree gave.
// cap of the cache: three quarters of the device's memory (a db of 1e4 strains with its scratch is ~100 GB: under the 48 GB of round 4 every
// call of the pipeline seam paid 0.7-1.2 s of hipMalloc for the blocks that did not fit the cache)
size_t dev_cache_max(int dev);
void dev_cache_set_max(long long bytes);   // < 0: per-device default
hipError_t dev_cache_alloc(void **p, size_t bytes, size_t *cap_out, int *dev_out);
void dev_cache_free(void *p, size_t cap, int dev);   // dev: the device the block was allocated on (the caller's current device may differ)
void dev_cache_trim();   // really free everything cached for the current device (pantax_hip_destroy)
void dev_cache_register_stream(int dev, hipStream_t st, bool add);   // a ctx's compute streams: what a released block may still be in use on
struct DevCacheIdleFrees { DevCacheIdleFrees(); ~DevCacheIdleFrees(); };   // scope: blocks released by this thread are idle already (see ctx.cpp)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t cap = 0;      // bytes of the underlying allocation (owned buffers)
    int dev = 0;         // device it lives on
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && owned) dev_cache_free(p, cap, dev);
        p = nullptr;
        n = 0;
        cap = 0;
        owned = true;
    }
    // non-owning window into a larger allocation (an arena that is zeroed / downloaded as one piece)
    void view(void *ptr, size_t count) {
        release();
        p = static_cast<T *>(ptr);
        n = count;
        owned = false;
    }
    hipError_t alloc(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = dev_cache_alloc((void **)&p, (count ? count : 1) * sizeof(T), &cap, &dev);
        if (e == hipSuccess) n = count ? count : 1;
        return e;
    }
    size_t bytes() const { return n * sizeof(T); }
    // take over another buffer's allocation
    void take(DevBuf &o) { release(); p = o.p; n = o.n; cap = o.cap; dev = o.dev; owned = o.owned; o.p = nullptr; o.n = 0; o.cap = 0; o.owned = true; }
};

// page-locked host staging (grow-only): a copy from / to pageable memory makes the runtime stage and wait,
// tens of microseconds per call; pinned copies are plain asynchronous DMA
struct PinBuf {
    uint8_t *p = nullptr;
    size_t n = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
    hipError_t reserve(size_t bytes) {
        if (bytes <= n && p) return hipSuccess;
        release();
        size_t want = bytes < 65536 ? 65536 : bytes;
        hipError_t e = hipHostMalloc((void **)&p, want, hipHostMallocDefault);
        if (e == hipSuccess) n = want;
        return e;
    }
};

struct TimedLaunch {
    const char *name;
    hipEvent_t start, stop;
};

// Every switch of the library, read ONCE from the environment by pantax_hip_init (PANTAX_<NAME>, upper case) and changed afterwards only
// through pantax_hip_set_option: no entry point reads the environment on its way (a host that calls setenv / std::env::set_var beside a
// running step would race with a read of the environment).  Product switches first; the rest selects in-tree HIP paths for tests and measurements.
struct CtxConfig {
    bool trace = false;              // hip_trace: wall time of the phases of the pipeline seam / db upload / GAF load on stderr
    int stage_threads = 32;          // host threads that fill the pinned upload ring (at 16 the filling, not the DMA, bounds a 15-GB load)
    int stage_ch_mb = 0;             // chunk size of the ring in MB (0: from the transfer size)
    bool stream_prio = true;         // main stream at the highest, side stream at the lowest priority
    int dev_cache_gb = -1;           // cap of the per-process cache of released device blocks in GB (-1: min(3/4 of the device, 9/10 of what was free at first use); 0: nothing cached)
    bool numa_bind = true;           // the upload crew and its pinned ring live on the GPU's NUMA node
    uint64_t gaf_piece_bytes = 0;    // largest piece of GAF text tokenised at once (0: a sixth of the text, 64 MiB .. 1 GiB)
    int db_groups = 0;               // file seam: groups of species that go through the device one after the other, the next one's graphs travelling meanwhile (0: by size; 1: one db)
    uint64_t db_path_steps_max = 0;  // path steps per resident db of the file seam (0: 3e9); a selection beyond it goes through the device group by group (tests lower it)
    // forced paths (tests compare them with the defaults)
    std::string trio_path;           // "block": every species through the node-block kernel; "bucket": global buckets
    std::string trio_rows;           // "path": lookup rows filed by the pass over the walks
    int uniq_hash = -1;              // bucket path: 1 LDS hash / 0 shuffles (-1: by mean bucket size)
    std::string mask;                // "walk": membership masks from the path walk
    std::string row_sort;            // "radix" / "nodes"
    std::string objective;           // "nodes": the LP objective summed over the nodes
    bool cov_general = false;        // every group through the kernel of the longer walks (coverage_fast_kernel<.., LONG>; cov_long=step: coverage_step_kernel)
    std::string cov_long;            // "step": round 5's coverage_step_kernel for the groups that hold steps of walks of more than 64 steps
    int covl_shape = -1;             // shape of the long-walk kernel: <U><groups per workgroup / 8><window / 1024><back / 256> (default 2234)
    bool cov_count = false;          // resident step: popcount_kernel as in the stage call
    int cov_clean_async = -1;        // resident step: the coverage arena's zero fill goes onto the side stream, beside the LPs (1; -1: arenas of 1 GiB and more); 0: in front of the coverage pass
    bool cov_self_clean = false;     // resident step: the last readers of the coverage arena zero it instead of a zero fill in front of every coverage pass.  OFF: measured
                                     // slower (node_cov_stats_kernel 2.5 -> 6.9 ms with the stores among its loads against 1.5 ms of zero fill at 1e4 strains)
    bool walk_sum_in_bin = false;    // the walk sums of long reads inside the binning pass instead of by walk_sum_kernel.  OFF: measured -- the row of 16 lanes that streams a
                                     // long walk in bin_slots_kernel adds 0.63 ms there at the cfg5 share where walk_sum_kernel takes 0.38 (cfg5 at full size: +2.7 against 2.9)
    int ncs_prefix_min = 48;         // average node length (bases) from which the node statistics count covered bases through the per-stretch prefix in LDS
    bool ncs_no_prefix = false;      // node statistics of long-node graphs through the per-lane word loop (round 5's kernel; tests compare, measurements)
    bool cov_arena_verify = false;   // tests: a coverage pass that skips its zero fill first checks that the arena IS zero (fails with PANTAX_HIP_E_STATE)
    // measurement shapes
    int cov_item_groups = 0;         // groups of 64 steps per work item of the short-read coverage kernel (0: 64)
    int tv_u = 4, tv_rounds = 4, tf_u = 8, tf_rounds = 1, rows_u = 1, tb_slots = 256, trio_xcd = 3, cov_shape = -1, covf_shape = -1, cov_xcd = 0, group_bucket_bits = 0;
    uint32_t tv_ablate = 0, cov_ablate = 0, ssn_ablate = 0;
    bool trio_two_pass = false;      // every build through records + prefix + rows kernel, as a db's first build (tests, measurements)
    bool no_absent_skip = false;     // the statistics / histogram passes of the step read the species the species level dropped like the others (tests compare, measurements)
    bool ssn_debug = false, scan_no_huge = false, flag_rank_chained = false, ratio_kernel = false, mask_pass = false, trio_free_at_filter = false,
         trio_after_step = false;
};
int ctx_set_option(CtxConfig &cfg, const char *name, const char *value);   // 0, or PANTAX_HIP_E_INVALID for an unknown name / unparsable value

// The host side of the big uploads (GAF text, graph arrays) is a crew of threads that pread into a pinned ring while the DMA engine empties
// it: both want the memory of the GPU's own NUMA node -- a filler on the far socket writes the ring across the inter-socket link, and the DMA
// reads it back the same way (round 4: 0.365 .. 0.49 s for the same 15-GB load from box to box).  pantax_hip_init looks the node up
// (/sys/bus/pci/devices/<gpu>/numa_node); the crew's threads are bound to its CPUs and the ring is allocated from it (option numa_bind=0: off).
struct NumaInfo {
    int node = -1;                   // -1: unknown / a single node: nothing is bound
    std::vector<int> cpus;           // the node's CPUs
};

struct Ctx {
    std::recursive_mutex mu;         // one call at a time per ctx (PTX_ENTER)
    CtxConfig cfg;
    NumaInfo numa;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_main = nullptr;   // the main stream while `stream` is swapped to the side stream (api_step.cpp), else null
    hipStream_t stream2 = nullptr;   // side stream of the resident step (the trio index does not depend on the reads)
    hipStream_t stream_up = nullptr; // copy stream of the file seam's graph loader thread (round 6; created on first use): the next group of species travels while this one's tables are built
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_seq = nullptr;     // orders the side stream behind everything enqueued on the main stream so far (steps run strictly one after the other on the device)
    std::string err;
    std::mutex err_mu;               // guards `err` alone (fail() may run before PTX_ENTER)
    bool timing = false;
    std::string timing_filter;   // non-empty: only launches of this name are timed (two events per step instead of ~200)
    std::vector<TimedLaunch> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    std::map<std::string, std::pair<uint64_t, double>> acc;
    int n_cu = 256;
    // scratch reused across calls
    DevBuf<uint64_t> d_scalars;  // small counters (n_abort, ...)
    DevBuf<uint32_t> d_scan_ws;  // chained-scan workspace: ticket + one state word per tile (primitives.hip)
    uint32_t scan_epoch = 0;
    DevBuf<uint32_t> d_scan_ws2; // the same for scans enqueued on the side stream (they may run beside scans of the main stream)
    uint32_t scan_epoch2 = 0;
    PinBuf pin_down;             // staging of small downloads (valid until the next download through it)
    PinBuf pin_text;             // ring of pinned chunks of the large uploads (upload_staged: GAF text, graph arrays)
    PinBuf pin_up;               // ring of small uploads, in two halves: a half is re-entered only after the copies issued from it
    size_t pin_up_off = 0;       // on its last lap have run (an event per stream, recorded when the ring leaves the half) -- steps enqueued
    hipEvent_t pin_up_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // back to back share it without a host wait in between
    bool pin_up_ev_set[2] = {false, false};
};

// RAII-less timing scope: records events around a launch when ctx->timing is on
struct KTimer {
    Ctx *ctx;
    hipEvent_t start = nullptr, stop = nullptr;
    const char *name;
    KTimer(Ctx *c, const char *nm);
    ~KTimer();
};

int collect_timings(Ctx *ctx);

constexpr uint32_t BIN_PREFIX = 2048;   // head of (species, qlen) fetched with the counters (equal-length test, profile.rs:312-319)
size_t bin_counter_words(uint32_t S);
size_t bin_result_words(uint32_t S);
struct Db; struct Reads;
int species_profile_launch(Ctx *ctx, const Db *db, const Reads *rd, const unsigned long long *d_counters, const double *d_avg_len, int filtered,
                           uint8_t *d_keep, double *d_absolute);
void species_profile_host(uint32_t S, const uint32_t *head_qlen, size_t n_head, const int64_t *read_count, const int64_t *base_sum,
                          const int64_t *less_multi, const int64_t *uniq_count, const double *avg_len, int filtered, uint8_t *keep_out,
                          double *absolute_out, double *abundance_out);
constexpr int PATH_TILE = 1024;   // path positions per workgroup of the per-path-step kernels
constexpr int LAD_MAXP = 64;     // candidate paths of a species on the one-word path (one u64 membership mask per node)
constexpr int LAD_WIDE_NW = 4;   // mask words of a "wide" species (65 .. 256 haplotypes); more haplotypes ("huge"): whole groups of four words
constexpr int LAD_WIDEP = 64 * LAD_WIDE_NW;   // candidate paths up to which a species' solver state is sized at compile time

// One batch = every species that has at least one candidate path, solved concurrently
// (one workgroup per species; the LPs are block-diagonal: profile.rs:3297-3319 runs them as
// independent rayon tasks).
struct LadBatch {
    uint32_t S = 0;
    // per species (host mirrors + device)
    bool prezeroed = false;             // the result arena and d_mask were zeroed ahead of strain_enqueue (strain_prezero)
    std::vector<int32_t> h_p;           // [S] number of candidates (0 = not solved)
    std::vector<uint32_t> h_cand;       // [H] at hap_off[s] + k: candidate k -> hap index within species
    DevBuf<int32_t> d_p;
    DevBuf<int32_t> d_hap_bit;          // [H] bit index of hap in its species' candidate list, -1 if none
    DevBuf<uint32_t> d_hap_nt;          // [H] unique-trio rows of every haplotype and [S] "the species has any", copied by the first
    DevBuf<uint8_t> d_sp_trio;          //     filter: the second filter then needs no trio table (the next step may be rebuilding them)
    DevBuf<uint64_t> d_mask;            // [V] candidate membership mask per node (the 0/1 coeff matrix, row-wise)
    DevBuf<double> d_ab;                // [V] node_abundance = bases / len  (profile.rs:980-990)
    DevBuf<double> d_c0;                // [S] sum of ab over the nodes with ab > 0 and no column (the row sort straight from the nodes sums it): the objective's part without rows
    bool rows_c0_valid = false;         // ... is there for the rows lad_prepare has just sorted
    bool masks_in_sort = false;         // the last lad_prepare formed the masks inside the row sort: d_mask holds nothing
    DevBuf<unsigned long long> d_ratio; // [H*2] at 2 * (hap_off[s] + k): sum cov, sum len of candidate k (exact integers)
    // species that can have more than 64 candidates (more than 64 haplotypes): LAD_WIDE_NW mask words per node in a side
    // array, d_mask then holds a 64-bit hash of those words (rows are grouped by it; the words of every pattern are
    // collected after the grouping, and a hash collision is detected there, not assumed away)
    const void *wide_for = nullptr;     // the Db the wide tables below were laid out for
    uint32_t n_wide = 0;                // species with more than LAD_MAXP haplotypes
    uint32_t n_huge = 0;                // ... of them with more than LAD_WIDEP: mask words, W / G and the column state sized at run time
    uint64_t Vw = 0;                    // their four-word mask groups (nodes x words per node / 4)
    DevBuf<uint32_t> d_wide_off;        // [S] first four-word group of the species in the wide arrays, 0xFFFFFFFF for the others
    DevBuf<uint32_t> d_wide_nw;         // [S] mask words per node and per pattern (LAD_WIDE_NW, or more for a huge species), 0 for the others
    DevBuf<uint64_t> d_wide_woff;       // [n_wide] first double of W (G: twice that), then [n_wide] first column of the huge column state
    DevBuf<double> d_huge_f64;          // huge species: x, c, d, ub, fac, score, deriv, g of lad_solve_body, 64 nw entries each
    DevBuf<int> d_huge_i32;             //               act_type, act_jk, dir, act_i0, act_i1
    DevBuf<int> d_pat_act;              // [K] huge species: the basis slot that holds pattern k, or -1
    DevBuf<uint32_t> d_wide_list;       // [n_wide] species ids
    DevBuf<uint32_t> d_wide_slot;       // [S] index into d_wide_list (the solver's W / G scratch), 0xFFFFFFFF for the others
    DevBuf<uint64_t> d_maskw;           // [Vw * NW]
    DevBuf<uint64_t> d_pat_or, d_pat_and;   // [Vw * NW] OR / AND of the mask words of the nodes of every pattern
    DevBuf<double> d_wide_W, d_wide_G;  // per wide species (64 nw)^2 and 2 (64 nw)^2: basis inverse / elimination scratch of the wide solvers
    // per species node stats
    DevBuf<double> d_amax;              // [S] max node abundance (profile.rs:1316-1319)
    DevBuf<uint32_t> d_nvalid;          // [S] #nodes with abundance > 0 (= n_eval, profile.rs:1380-1385, :1447)
    DevBuf<double> d_nzsum;             // [S] sum of min_depth-filtered non-zero abundances (profile.rs:1193-1201)
    DevBuf<uint32_t> d_nzcnt;           // [S]
    DevBuf<double> d_partial;           // per-chunk partials of the two-level reductions
    DevBuf<uint32_t> d_obj_done;        // [S] arrival counters of objective_kernel (zero between launches)
    // sorted LP rows (a_v > 0 and mask != 0), grouped into patterns
    uint64_t n_rows = 0;
    uint32_t K = 0;
    const double *row_a = nullptr;      // [n_rows] abundances sorted by (species, mask, a) (points into the sort buffers)
    DevBuf<uint32_t> d_counts;          // {n_rows, K, pattern overflow flag} kept on the device
    uint32_t k_cap = 0;
    DevBuf<uint64_t> d_pat_mask;        // [K]
    DevBuf<uint32_t> d_pat_start;       // [K+1]
    DevBuf<uint32_t> d_pat_species;     // [K]
    DevBuf<uint32_t> d_sp_pat_off;      // [S+1]
    std::vector<uint32_t> h_sp_pat_off;
    // solver scratch per pattern
    DevBuf<double> d_pat_eps, d_sc_s, d_sc_rho;
    DevBuf<uint32_t> d_sc_lo, d_sc_up, d_ls_lo, d_ls_hi, d_ls_mid;
    // solver in/out per species
    DevBuf<double> d_x, d_x2, d_obj, d_obj2;      // [H] x of solve 1 / 2 at hap_off[s] + k, [S] objectives
    DevBuf<int32_t> d_status, d_iters, d_status2, d_iters2;   // [S]
    DevBuf<uint8_t> d_fixed2, d_need2;  // [H] at hap_off[s] + k, [S]: second-solve decisions (second_filter_species)
};


// scratch of trio_index_build, kept across calls (grow-only) so a rebuild costs no hipMalloc
struct TrioScratch {
    DevBuf<uint32_t> zero_arena;   // uniq_q | first_cnt [| cnt | cursor]: what a build of the PATH route needs cleared goes first
    DevBuf<uint32_t> cnt, cursor, bucket_off, scan_tmp, first_cnt, d_tot;
    DevBuf<uint4> bucket;   // (q, b, c, global first node) per window
    DevBuf<uint32_t> uniq_q;       // path route: one bit per path position: its window occurs once in the species
    DevBuf<uint32_t> row_q;        // path route: window start of the row filed at every slot (the canonical order inside a node is made from it)
    // rows filed from the visit kernel's records (the species the visit table covers)
    DevBuf<uint64_t> vis_uq;       // [n_vgroups + 1] ballot of the unique visits of every group
    DevBuf<uint4> vis_rec;         // [n_vgroups * 8] the first eight unique windows of every group {window start, smaller end, larger end, middle}
    DevBuf<uint32_t> gprefix;      // [n_vgroups + 1] unique visits before the group = row of its first unique window; a function of the graphs alone: the first build's
                                   // values stay with the visit table and let later builds file in one pass (verified there)
    uint32_t gprefix_for = 0;      // the number of groups gprefix was computed for (0: not yet)
    DevBuf<uint32_t> group_sums;   // unique visits per tile of 4096 groups, then their prefix (group_tile_* kernels)
    DevBuf<uint32_t> hap_cnt;      // [H] first build of a db: rows per haplotype (-> hap_trio_off)
};

// ---- resident DB -----------------------------------------------------------------------------
inline uint64_t next_db_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
struct Db {
    const uint64_t uid = next_db_uid();   // never reused: what resident reads remember a db by (Reads::long_sums_db)
    uint32_t S = 0;
    uint64_t V = 0, H = 0, P = 0, L = 0;
    std::vector<int64_t> h_range_start, h_range_end;
    std::vector<uint64_t> h_node_off, h_hap_off, h_path_off;
    bool ranges_sorted_disjoint = true;
    // binning tables
    DevBuf<uint32_t> d_rng_start, d_rng_end, d_rng_idx;  // sorted by start (or file order when overlapping)
    // graph
    DevBuf<uint32_t> d_sp_first_id;  // [S] range_start as u32
    DevBuf<uint32_t> d_node_base;    // [S+1]
    DevBuf<uint64_t> d_bit_off;      // [V+1] prefix sum of node_len; node_len[v] = bit_off[v+1]-bit_off[v]
    DevBuf<uint32_t> d_node_len;     // [V] the same lengths as 4-byte gathers
    DevBuf<uint4> d_node_rec;        // [V] {bit_off lo, bit_off hi (8 bits) | #lookup rows (16 bits) << 8 | filter of the rows' pairs (8 bits) << 24, len, first lookup row}: one 16-byte gather per step
                                     //     carries the node AND the head of its unique-trio lookup rows (written by every trio build)
    DevBuf<uint64_t> d_path_off;     // [H+1]
    DevBuf<uint32_t> d_path_nodes;   // [P]
    DevBuf<uint32_t> d_hap_species;  // [H]
    DevBuf<uint64_t> d_hap_off;      // [S+1]
    DevBuf<uint8_t> d_all_same;      // [S] every haplotype of the species walks the same nodes (profile.rs:1188-1190)
    std::vector<uint8_t> h_all_same;
    DevBuf<uint2> d_tiles;           // path tiles {hap, chunk} ordered (species, chunk, hap); one workgroup each
    uint64_t n_tiles = 0;
    DevBuf<uint2> d_emit_tile_sp;    // [ceil(V / 2048)] {species of the first node, of the last node} of every 2048-node tile of the row compaction
    // node -> haplotypes (node_haps_build, stage_lad.hip; built once at upload): bit j of d_node_haps[v] = the walk of haplotype j of the
    // node's species visits v; species of more than 64 haplotypes are left at zero (nh_walk_too: there are such species)
    DevBuf<uint64_t> d_node_haps;    // [V]
    bool nh_built = false, nh_walk_too = false;
    DevBuf<uint32_t> d_tile_rank;    // [n_tiles] rank of the tile in path order (hap-major)
    DevBuf<uint32_t> d_hap_tile_off; // [H+1] first path-order tile of every haplotype
    // node-block run table of the walks (trio_runs_build, stage_trio.hip; a layout table like d_tiles, built once at upload):
    // every species' nodes are cut into blocks of TRIO_BLK nodes, every walk into maximal runs of consecutive positions
    // whose nodes lie in ONE block; runs are grouped by block.  The unique-trio build then gives each block to one
    // workgroup, which sees every window whose middle node lies in the block -- whatever haplotype it is on.  Only species the visit
    // table leaves to this path have blocks.
    // visit table of the walks (trio_visits_build, stage_trio.hip; the default path of the unique-trio build since round 4): the interior
    // positions listed node by node in groups of 64 visits that no node straddles
    bool trio_visit_ok = false;
    uint32_t n_vgroups = 0;
    std::vector<uint8_t> h_trio_slow; // [S] 1: the species holds a node with more than 64 visits (or the table was not built): node-block kernel
    DevBuf<uint32_t> d_trio_slow;    // [S] the same
    DevBuf<uint32_t> d_vis_pos;      // [64 n_vgroups] path position of the visit (the middle of its window), 0xFFFFFFFF pads at a group's tail
    DevBuf<uint64_t> d_vis_head;     // [n_vgroups] bit l: lane l holds the first visit of a node
    DevBuf<uint32_t> d_vis_nbase;    // [n_vgroups] node base of the group's species
    DevBuf<uint32_t> d_vis_sp;       // [n_vgroups] the group's species (the rows kernel finds a window's haplotype among that species' walks)
    DevBuf<uint32_t> d_node_visited; // [V / 32 + 2] bit v: node v has an interior visit
    bool trio_block_ok = false;      // every species left to the node-block kernel has < 2^27 nodes (the 64-bit LDS key packs (middle in block, lo, hi)) and P < 2^32
    uint32_t n_blocks = 0;
    uint64_t n_runs = 0;
    DevBuf<uint32_t> d_blk_base;     // [S+1] first block of every species
    DevBuf<uint32_t> d_blk_species;  // [n_blocks]
    DevBuf<uint32_t> d_blk_run_off;  // [n_blocks+1]
    DevBuf<uint4> d_blk_rec;         // [n_blocks+1] {first run, end run, global first node, species-local first node / 64 | node count << 24}
    DevBuf<uint4> d_runs;            // [n_runs] {first position, #positions, walk begin, walk end} (global path positions)
    // unique-trio index (a7).  ROWS ARE NUMBERED IN THE ORDER THEY ARE FILED (round 5): the species of the visit table in table order -- node after
    // node, a node's rows sorted by their pair of ends --, behind them the species left to the pass over the walks, node after node in the same
    // canonical order.  A row IS its lookup entry: d_trio_ent[r], d_trio_len[r], d_trio_hap[r], d_trio_bases[r] belong together, the lookup head
    // of a node {first row, #rows} rides in its node record.  The (species, hap, position) order the C ABI hands out (pantax_hip_trio_get,
    // trio_bases of pantax_hip_node_coverage) is a permutation made on request (trio_export_ensure): it costs nothing on the step's path.
    bool trio_built = false;
    bool trio_prefetched = false;   // pantax_hip_trio_index_prefetch built the index of the COMING step: that step's rebuild_trio is served by it
    bool trio_keys_built = false;   // d_trio_q (window start of every row: what the export order is made from) was written by the last build
    bool trio_perm_valid = false;   // d_trio_perm holds the export order of the last build
    uint64_t U = 0;
    bool cov_prepared = false;       // coverage_prepare ran for the coming coverage_launch
    bool cov_count_pending = false;  // d_cov of the last coverage pass is still to be counted from the bitmap (node_stats_launch does it)
    // round 6: in the resident step the last readers of the coverage arena (hap_rows_pass_kernel<0>: trio_bases; node_cov_stats_kernel: bases, bit vector,
    // full-node flags) zero what they read, and the next coverage pass skips its zero fill.  cov_self_clean: this step's readers clean (set by
    // strain_enqueue); cov_arena_clean + its signature: the arena is all zero in exactly this layout (reset by whatever dirties it).
    bool cov_self_clean = false, cov_arena_clean = false;
    uint64_t cov_arena_sig = 0;
    // ... or (cov_clean_async) the arena is zero-filled on the SIDE stream from behind those readers, beside the step's row sort and LPs: the next coverage
    // pass waits for ev_cov_clean instead of filling in front of itself
    size_t cov_arena_total = 0;
    hipEvent_t ev_cov_read = nullptr, ev_cov_clean = nullptr;
    bool cov_clean_pending = false;
    bool trio_sizes_known = false;   // U, the rows per haplotype and per species depend on the graphs only: kept across db_reset
    uint64_t U_known = 0;
    bool trio_layout_fast = false;   // the sizes were learnt by a build that filed the visit table's species from its records (else: every species by the pass over the walks)
    DevBuf<uint32_t> d_trio_first;   // [V+1] path route: first row of every node (scratch of the build; the heads in node_rec are what the step reads)
    DevBuf<uint2> d_trio_ent;        // [U] {smaller end, larger end} of the window (global node indices); the middle is the node that heads the row
#if TRIO_LH_PACK
    DevBuf<uint2> d_trio_len;        // [U] {summed length of the window's three nodes (profile.rs:712), the haplotype that owns the row as an index within its species}
#else
    DevBuf<uint32_t> d_trio_len;     // [U] summed length of the window's three nodes (profile.rs:712)
    DevBuf<uint16_t> d_trio_hap;     // [U] the haplotype that owns the row, as an index within its species
#endif
    DevBuf<uint32_t> d_trio_q;       // [U] window start (path position) of the row: export builds only
    DevBuf<uint32_t> d_trio_perm;    // [U] row of the e-th window in (species, hap, position) order: export builds only
    DevBuf<uint64_t> d_hap_trio_off; // [H+1] prefix of the rows per haplotype = offsets of the export order (the first filter reads the counts)
    std::vector<uint64_t> h_hap_trio_off;
    // per-haplotype statistics by key (hap_trio_stats_launch): the rows of a species are one contiguous block, cut into chunks
    std::vector<uint32_t> h_sp_row_order; // [S] the species in filing order
    struct NodeChunks { DevBuf<uint32_t> d_chunk_sp, d_sp_off; uint32_t n = 0; };   // node statistics: chunk -> species, first chunk of every species (chunks by size)
    NodeChunks node_chunks[2];       // [0] node_stats_kernel, [1] node_cov_stats_kernel
    DevBuf<uint4> d_stat_chunks;     // {species, first row, end row, first partial} per chunk of rows
    DevBuf<double> d_hs_x;         // [U] a9 statistics: the non-zero rows' abundances, compacted chunk by chunk (written by pass 0 of every step)
    DevBuf<uint16_t> d_hs_h;       // [U] ... their owners
    DevBuf<uint32_t> d_hs_n;       // [n_stat_chunks] ... entries per chunk
    uint32_t n_stat_chunks = 0;
    uint64_t n_stat_partials = 0;    // sum over chunks of the haplotypes of their species
    uint32_t stat_lds_haps = 0;      // most haplotypes of a species whose chunk accumulators live in LDS (65 .. 1024; up to 64: lane registers)
    bool stat_global_rows = false;   // a species of more than 1024 haplotypes accumulates in its chunks' own (zero-filled) rows of partials
    DevBuf<uint32_t> d_sp_chunk_off; // [S+1] first chunk of every species (species without rows: empty range)
    // coverage state (a8), resident for the strain step
    bool cov_done = false;
    // d_bases, d_trio_bases, the abort counter and d_bitmap are windows of one arena: one memset per coverage pass
    DevBuf<uint8_t> d_cov_arena;
    unsigned long long *d_abort = nullptr;
    DevBuf<unsigned long long> d_bases;      // [V]
    DevBuf<uint32_t> d_bitmap;               // [ceil(L/32)+1]
    uint64_t item_sel_layout = 0;            // ... made for the reads layout of this id (Reads::layout_id); n entries; on = the kernel takes the list (off: every item)
    uint32_t item_sel_n = 0;
    bool item_sel_on = false;
    DevBuf<uint32_t> d_item_sel;             // the work items of the short-read coverage kernel that meet the id ranges of this db's species (a db of SOME of the species)
    DevBuf<uint32_t> d_full;                 // 1 bit per node: some step covered the node whole (its bits are then not marked one by one)
    DevBuf<uint32_t> d_cov;                  // [V]
    DevBuf<unsigned long long> d_trio_bases; // [U]
    DevBuf<uint8_t> d_active;                // [S]
    DevBuf<uint8_t> d_ones;                  // [S] all ones: the `active` table of a coverage pass that deselects nothing
    DevBuf<unsigned long long> d_counters;   // [4*S] species counters of bin_reads
    // persistent scratch of the derived stages
    TrioScratch trio_scratch;
    LadBatch lad;
    DevBuf<uint32_t> d_hap_nnz;              // [H]
    DevBuf<double> d_hap_mean;               // [H]
    DevBuf<double> d_hap_part;               // chunk partials of the three passes of the per-hap trio statistics
    DevBuf<uint8_t> d_arena;                 // every small result of the strain step, contiguous: one memset, one download
    PinBuf h_arena[2];               // pinned mirrors of d_arena: one per step in flight (a caller may enqueue step i+1 before it collects step i)
    DevBuf<double> d_avg_len, d_sp_abs;   // resident step: species lengths in, predicted_coverage out (d_active holds keep)
    DevBuf<uint8_t> d_sp_out;        // backing store of d_sp_abs + d_active in a resident step (one download)
    PinBuf h_sp_out[2];              // [S f64 absolute][S u8 keep], per step in flight
    hipEvent_t ev_step[2] = {nullptr, nullptr};   // recorded behind the last download of the step that uses the slot
    pantax_hip_step_config step_cfg[2];
    int step_enq = 0, step_col = 0, step_inflight = 0;   // slot of the next enqueue / the next collect; steps enqueued and not yet collected
    hipEvent_t ev_trio_free = nullptr;   // recorded behind the last reader of the unique-trio tables in a step: the next step's rebuild waits
    bool trio_free_valid = false;        // for this, not for the whole previous step (its row sort and LPs run beside the rebuild)
    bool trio_free_pending = false;      // the event is still to be recorded by lad_prepare, behind the row compaction
    ~Db() { for (hipEvent_t e : ev_step) if (e) (void)hipEventDestroy(e); if (ev_trio_free) (void)hipEventDestroy(ev_trio_free);
            if (ev_cov_read) (void)hipEventDestroy(ev_cov_read); if (ev_cov_clean) (void)hipEventDestroy(ev_cov_clean); }
    // LP-row staging (lad_prepare)
    DevBuf<uint32_t> d_scan_tmp, d_sort_table, d_ss_ws;
    DevBuf<uint64_t> d_ka[3], d_kb[3];
    DevBuf<uint64_t> d_row16;   // [4 V] the 16-byte staged and bucketed records of the node-order row sort (sample_sort_nodes.hip)
};

struct Reads {
    uint64_t R = 0, T = 0;
    DevBuf<uint32_t> d_step_off, d_node_id, d_pstart, d_pend, d_qlen;
    DevBuf<uint8_t> d_mapq, d_flags;
    DevBuf<uint64_t> d_id_hash;      // 64-bit hash of every read id (device tokenizer; the duplicate-id rule, profile.rs:361-437): fetched by the host only when two reads share one
    bool has_flags = false;
    DevBuf<int32_t> d_species;
    std::vector<int32_t> h_pre_species;   // first rows of species/qlen, fetched with the counters (equal-length test)
    std::vector<uint32_t> h_pre_qlen;
    // locus-grouped copy of the stream the coverage kernel walks (built once per upload)
    // Walks of at most 64 steps never straddle a 64-step boundary of the stream (pad steps carry slot 0xFFFFFFFF),
    // so a wave of the coverage kernel always holds whole reads.
    uint64_t T_pad = 0;              // steps in the padded stream
    DevBuf<uint32_t> d_g_node_id, d_g_group_slot, d_slot_of;  // [T_pad] node ids, [T_pad / 64] slot of the read that owns each 64-step group's first step, [R] read -> slot (~0: no walk)
    DevBuf<uint4> d_g_read_rec;      // [R'] {first step, #steps, pstart, pend}
    DevBuf<uint8_t> d_g_step_dup;    // [T_pad] step codes: walks <= 64 steps: distance back to the first occurrence of the step's node (0 none);
                                     //         longer walks: 0x80 | (node occurred earlier in the walk); both | 0x40 on a walk's first step; pads 0xFF
    DevBuf<uint32_t> d_long_sum;     // [R'] walks > 64 steps: node lengths of all steps but the last (walk_sum_kernel), else unused
    DevBuf<uint32_t> d_long_len0;    // [R'] walks > 64 steps: length of the walk's first node (walk_sum_kernel)
    uint32_t n_long = 0;             // walks of more than 64 steps
    DevBuf<uint2> d_g_slot_rec;      // [R'] {species of the slot's read (coding below), node base - first node id of that species}, written by the binning pass
    DevBuf<uint2> d_g_qm;            // [R'] {read length, MAPQ} in slot order (the binning pass runs over the slots)
    DevBuf<uint8_t> d_g_flag;        // [R'] drop flags in slot order, refreshed when the flags changed (g_flags_valid)
    uint32_t n_slots = 0;            // reads that own a slot (non-empty walk)
    DevBuf<uint2> d_g_items;         // [n_items] work items of the short-read coverage kernel: groups [x, y) whose reads start inside one block of 2048 node ids
    uint32_t n_items = 0;
    int item_blk_shift = 0;          // log2 of the node-block size the items were cut at (0: plain cuts of the stream)
    bool g_flags_valid = false;
    bool species_valid = false;      // d_species (file order) reflects the last binning pass; species_ensure() gathers it from the slots
    bool binned = false;
    bool grouped = true;             // false: columns only (a slice that will be routed away, stage_route.hip; the file seam until its graphs travel): no locus-grouped copy, no coverage pass
    uint64_t layout_id = 0;          // a new number for every locus-grouped copy built (build_step_read)
    uint64_t long_sums_db = 0;       // Db::uid of the db whose binning pass filled d_long_sum / d_long_len0 (0: nobody: walk_sum_kernel does it)
    std::vector<uint32_t> h_item_block;   // node block (first node id >> item_blk_shift) of every work item of the short-read coverage kernel, ascending (host)
    uint32_t max_node_id = 0;        // largest node id of the walks (the device tokenizer notes it: what a later reads_group() sizes its buckets by)
};

// slot record (Reads::d_g_slot_rec).x: >= 0 species, the coverage pass uses the slot; -1 "U"; -2 - s: binned to species s but dropped
// before get_node_abundances (drop flag)
__host__ __device__ inline int slot_species(int32_t x) { return x >= -1 ? x : -x - 2; }

// node record fields (Db::d_node_rec): the coverage bitmap of one GPU holds < 2^40 bases and a node heads < 2^24 lookup rows
constexpr uint64_t NODE_REC_MAX_BITS = 1ull << 40;
constexpr uint32_t NODE_REC_MAX_ROWS = 1u << 16;
// lookup head of a node inside its record: rows = the unique windows whose middle the node is; the 8-bit filter has the bit nr_pair_bit(lo, hi)
// of every row's pair of ends set -- a window whose bit is clear is not among the rows, and the coverage pass does not fetch them (round 4: 40 % of
// the steps have a middle node with rows, 8 % hit one).  A builder that does not compute the filter stores 0xFF (never wrong, never skips).
__host__ __device__ inline uint32_t nr_rows(uint32_t y) { return (y >> 8) & 0xFFFFu; }
__host__ __device__ inline uint32_t nr_filter(uint32_t y) { return y >> 24; }
__host__ __device__ inline uint32_t nr_head(uint32_t y_old, uint32_t rows, uint32_t filter) { return (y_old & 0xFFu) | ((rows & 0xFFFFu) << 8) | (filter << 24); }   // (a count beyond the field is reported by the builders, never spilt into the filter)
__host__ __device__ inline uint32_t nr_pair_bit(uint32_t lo, uint32_t hi) { return 1u << (((lo * 0x9E3779B1u) ^ (hi * 0x85EBCA77u)) >> 29); }
__host__ __device__ inline uint64_t nr_bit_off(const uint4 &r) { return ((uint64_t)(r.y & 0xFFu) << 32) | r.x; }
__host__ __device__ inline uint4 nr_make(uint64_t bit_off, uint32_t len) { return make_uint4((uint32_t)bit_off, (uint32_t)(bit_off >> 32) & 0xFFu, len, 0u); }

template <class T>
int upload(Ctx *ctx, DevBuf<T> &dst, const T *src, size_t n) {
    PTX_HIP(ctx, dst.alloc(n));
    if (n) PTX_HIP(ctx, hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}
// A stretch of a 32-bit device array as it lies on the host: in memory (`src`) or in a file (`file` = index into the paths the
// upload is given, bytes from `file_off`); `narrow`: the source holds 64-bit little-endian integers (bincode's i64 lengths / usize
// node ids, zip.rs:171-190) that become 32-bit ones on the way -- a value beyond 2^32 - 1 (or negative) fails the upload.
struct UploadSeg {
    const void *src = nullptr;
    int32_t file = -1;
    uint64_t file_off = 0;
    uint64_t out_bytes = 0;   // bytes this stretch occupies on the device (a multiple of 4); the source holds twice as many when narrow
    bool narrow = false;
    bool hole = false;        // nothing is filled (the bytes travel as they lie in the ring): a stretch another pass writes on the device
};
// segments back to back -> d_dst, as ONE chunk pipeline through the pinned ring (a crew of host threads fills a chunk -- pread /
// memcpy / narrowing -- while the chunks before it travel).  *bad_seg (optional) = first segment that held a value beyond 32 bits, or -1.
int upload_segments(Ctx *ctx, void *d_dst, const UploadSeg *segs, size_t n_segs, const std::string *files, int64_t *bad_seg, hipStream_t stream = nullptr /* null: ctx->stream */);

// one species' graph as the db upload takes it: where its 32-bit node lengths and its walks (32-bit species-local node ids, haplotype
// after haplotype) come from, and the walks' local CSR offsets (path_off may start anywhere).  Nothing of the big arrays is touched on
// the host: lengths > 0, walks inside the graph and the identical-walk test are checked on the device (stage_db.hip).
// Round 6 (image format 4): the walks of a species PACKED -- blocks of PK_BLOCK consecutive positions of its concatenated walks, every block
// {first node id (u32), zigzag deltas of 1, 2 or 4 bytes, PK_BLOCK of them (the first is 0)}; node ids run along a walk in steps of one or
// two (a pangenome graph is numbered along its backbone), so 97 % of the blocks take one byte per step: the walks cross PCIe at a quarter
// of their size and are unpacked in HBM (walks_unpack_kernel, stage_db.hip).  `off` = payload offset of every block in units of PK_UNIT bytes
// (n_blocks + 1 entries; a block's width is the difference).  Node lengths may lie as u16 (every length of the species below 2^16).
constexpr uint32_t PK_BLOCK = 256, PK_UNIT = 256;
struct PackedWalks {
    uint64_t n_blocks = 0, payload_bytes = 0;   // payload: n_blocks blocks of PK_BLOCK x {1, 2, 4} bytes
    UploadSeg first_seg, off_seg, payload_seg;  // u32[n_blocks] | u32[n_blocks + 1] | bytes
};
struct GraphPart {
    uint64_t n_nodes = 0, n_haps = 0;
    const uint64_t *path_off = nullptr;
    UploadSeg len_seg;                 // len16: out_bytes = 2 * n_nodes rounded up to 4 (u16 lengths, widened on the device)
    bool len16 = false;
    std::vector<UploadSeg> walk_segs;  // empty when `packed`
    bool packed = false;
    PackedWalks pk;
};
// stage_db.hip: the packed walks / 16-bit lengths of the species that have them -> d_path_nodes / d_node_len (the others' stretches are left alone)
struct UnpackSpecies { uint32_t blk_base, off_base, payload_base /* PK_UNIT bytes */, out_base, n_steps; };
int walks_unpack_launch(Ctx *ctx, const UnpackSpecies *d_table, uint32_t n_species, uint32_t n_blocks, const uint32_t *d_first, const uint32_t *d_off,
                        const uint8_t *d_payload, uint32_t *d_path_nodes, hipStream_t stream = nullptr /* null: ctx->stream (timed) */);
struct WidenSpecies { uint64_t src_base /* u16 index */, dst_base; uint64_t n; };
int lens_widen_launch(Ctx *ctx, const WidenSpecies *d_table, uint32_t n_species, uint64_t n_total, const uint16_t *d_len16, uint32_t *d_node_len, hipStream_t stream = nullptr);
// stage_db.hip: node tables (global prefix of the lengths), walk check and identical-walk test on the device
int node_tables_launch(Ctx *ctx, Db *db, uint32_t *d_flags /* [2]: {a node of length 0, 1 + first haplotype that leaves its graph (0xFFFFFFFF: none)} */);
int db_upload_parts(Ctx *ctx, uint32_t S, const int64_t *range_start, const int64_t *range_end, const GraphPart *parts, const std::string *files,
                    pantax_hip_db **out);
// the same in three steps (round 6: the file seam loads the NEXT group of species on a thread of its own while this group's tables are built):
// begin -- host tables, small uploads, the big arrays allocated (calling thread, ctx->stream); arrays -- node lengths and walks from their files into
// HBM, unpacked where they come packed (any thread that has made the ctx's device current; `stream`: its copy stream, waited for before it returns;
// touches neither ctx->stream nor the ctx's staging buffers); finish -- node tables, checks, tiles, visit table, node -> haplotypes (ctx->stream)
int db_upload_begin(Ctx *ctx, uint32_t S, const int64_t *range_start, const int64_t *range_end, const GraphPart *parts, pantax_hip_db **out);
int db_upload_arrays(Ctx *ctx, pantax_hip_db *db, const GraphPart *parts, const std::string *files, hipStream_t stream);
int db_upload_finish(Ctx *ctx, pantax_hip_db *db);
// large pageable host buffers (mmapped text, graph arrays) -> HBM through two pinned chunks: a few threads copy the next
// chunk into pinned memory while the previous one is on its way over PCIe (a plain copy from pageable memory is staged
// by one runtime thread at ~10 GB/s).  Returns after the last chunk has arrived.
int upload_big(Ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
// the same from an open file (pread straight into the pinned chunks: no page of the file is mapped or faulted in)
int upload_file(Ctx *ctx, void *d_dst, int fd, uint64_t file_off, uint64_t bytes);
// a text in pieces [piece_off[k], piece_end[k]) with their own destinations, as ONE chunk pipeline on `stream` (the GAF load: piece k is
// tokenised while k+1.. travel).  before_piece(k) may block until d_dst[k] is free (false aborts); after_piece(k) runs once the last
// chunk of piece k has been enqueued (record an event there).  fd >= 0: pread from the file at file_base + offset, else from `text`.
int upload_text_pieces(Ctx *ctx, size_t n_pieces, void *const *d_dst, const char *text, int fd, uint64_t file_base, const uint64_t *piece_off, const uint64_t *piece_end,
                       hipStream_t stream, const std::function<bool(size_t)> &before_piece, const std::function<int(size_t)> &after_piece);
// fn(begin, end) over [0, n) split across up to n_threads host threads (the calling thread takes the first slice)
void parallel_for(uint64_t n, int n_threads, const std::function<void(uint64_t, uint64_t)> &fn);
// small host -> device copies go through the pinned ring (the source may be reused as soon as this returns)
constexpr size_t PIN_UP_RING = 1u << 20, PIN_UP_MAX = 1u << 16;
template <class T>
int upload_small(Ctx *ctx, DevBuf<T> &dst, const T *src, size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes == 0 || bytes > PIN_UP_MAX) return upload(ctx, dst, src, n);
    PTX_HIP(ctx, dst.alloc(n));
    PTX_HIP(ctx, ctx->pin_up.reserve(PIN_UP_RING));
    size_t off = (ctx->pin_up_off + 63) & ~(size_t)63;
    constexpr size_t HALF = PIN_UP_RING / 2;
    const int h_old = ctx->pin_up_off ? (int)((ctx->pin_up_off - 1) / HALF) : 0;   // half of the last byte handed out
    if (off < HALF && off + bytes > HALF) off = HALF;        // would straddle the middle: start of the upper half (bytes <= PIN_UP_MAX <= HALF)
    if (off + bytes > PIN_UP_RING) off = 0;                  // past the end: start of the lower half
    const int h_new = off >= HALF ? 1 : 0;
    if (h_new != h_old) {
        // leaving h_old: mark what was issued from it on either stream; entering h_new: its last lap's copies must have run.  The
        // records go onto the REAL main stream and the side stream: ctx->stream may be swapped to the side stream at this moment (the
        // trio build of a step), and copies issued from this half on the main stream before the swap must be covered too
        // (round-3 advisor finding)
        for (int k = 0; k < 2; ++k) {
            hipStream_t st = k == 0 ? (ctx->stream_main ? ctx->stream_main : ctx->stream) : ctx->stream2;
            if (!st) continue;
            if (!ctx->pin_up_ev[h_old][k]) PTX_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_up_ev[h_old][k], hipEventDisableTiming));
            PTX_HIP(ctx, hipEventRecord(ctx->pin_up_ev[h_old][k], st));
        }
        ctx->pin_up_ev_set[h_old] = true;
        if (ctx->pin_up_ev_set[h_new])
            for (int k = 0; k < 2; ++k) if (ctx->pin_up_ev[h_new][k]) PTX_HIP(ctx, hipEventSynchronize(ctx->pin_up_ev[h_new][k]));
    }
    std::memcpy(ctx->pin_up.p + off, src, bytes);
    ctx->pin_up_off = off + bytes;
    PTX_HIP(ctx, hipMemcpyAsync(dst.p, ctx->pin_up.p + off, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}
template <class T>
int download(Ctx *ctx, T *dst, const T *src_dev, size_t n) {
    if (n) PTX_HIP(ctx, hipMemcpyAsync(dst, src_dev, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

inline int grid_for(uint64_t work, int block, int max_blocks = 256 * 8) {
    uint64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

// ---- stage entry points (host launchers, defined in the .hip files) ---------------------------
int bin_reads_launch(Ctx *ctx, const Db *db, Reads *rd, unsigned long long *d_counters /*[4*S]*/);
int species_ensure(Ctx *ctx, Reads *rd);   // d_species in file order (resident reads keep the species per slot)
int coverage_arena_clean_async(Ctx *ctx, Db *db);   // resident step: the arena's zero fill on the side stream (option cov_clean_async)
int coverage_prepare(Ctx *ctx, Db *db, Reads *rd, bool with_trio);   // optional, ahead of coverage_launch (needs the binning and db->U only)
// defer_count: leave node_base_cov (popcount_kernel) to the node statistics pass that follows in the resident step (db->cov_count_pending)
int coverage_launch(Ctx *ctx, Db *db, Reads *rd, const uint8_t *d_active, bool with_trio, bool defer_count = false);
// the locus-grouped copy of reads that were tokenised / uploaded as plain columns (round 6: the file seam bins the plain columns for the species
// decision and builds the copy while the first graphs travel); no-op on grouped reads
int reads_group(Ctx *ctx, Reads *rd);
int trio_index_build(Ctx *ctx, Db *db, bool with_keys = true);
int trio_keys_ensure(Ctx *ctx, Db *db);
#if TRIO_LH_PACK
#define TRIO_HAP_PTR(db) ((const uint16_t *)nullptr)
#else
#define TRIO_HAP_PTR(db) ((const uint16_t *)(db)->d_trio_hap.p)
#endif
int trio_export_ensure(Ctx *ctx, Db *db); // d_trio_perm: the (species, hap, position) order of the rows, for the exporters
int trio_export_u64(Ctx *ctx, Db *db, const unsigned long long *d_src, unsigned long long *d_dst);   // d_dst[e] = d_src[row of e]
int hap_stats_layout(Ctx *ctx, Db *db, const uint64_t *sp_first_row, const uint64_t *sp_rows);       // first build of a db: the chunk table of hap_trio_stats_launch
int trio_visits_build(Ctx *ctx, Db *db); // end of db upload: the visit table (and which species it leaves to the node-block kernel)
int trio_runs_build(Ctx *ctx, Db *db);   // end of db upload, after trio_visits_build: the node-block run table of those species
int node_haps_build(Ctx *ctx, Db *db);   // end of db upload: node -> haplotypes (the LP's membership masks built by node)
bool use_node_haps(const Ctx *ctx, const Db *db);
struct HostReads;
// stage_gaf.hip: text -> host columns (+ walks unless `resident` is given, which then owns the packed reads in HBM)
int gaf_tokenize_device(Ctx *ctx, const char *text, uint64_t size, HostReads &out, Reads *resident = nullptr, int fd = -1, uint64_t file_base = 0, bool group = true,
                        bool want_id_spans = false, bool want_host_columns = true);
int build_step_read(Ctx *ctx, Reads *rd, uint32_t max_node_id);
// the per-read host columns of resident reads (read_len, mapq, flags, id hashes), on request: a caller that needs none of them -- the
// pipeline seam on distinct read ids without a binning report -- never brings 14 bytes per read back over PCIe
int reads_host_columns(Ctx *ctx, const Reads *rd, HostReads &out);
// stage_route.hip: binned reads -> one message per owner rank, and back to resident reads on the owner
struct Route {
    int W = 0;
    std::vector<uint64_t> n_reads, n_steps, word_off;   // per owner; word_off [W+1] (32-bit words)
    DevBuf<uint32_t> d_send;                            // the W messages back to back
    PinBuf h_send;                                      // host copy, made on demand
    bool h_valid = false;
};
int route_pack(Ctx *ctx, const Db *db, const Reads *rd, const int32_t *owner_of_species, int W, Route &rt);
int reads_from_routed(Ctx *ctx, const uint32_t *d_recv, int W, const uint64_t *n_reads_from, const uint64_t *n_steps_from, bool group, Reads *rd);

}  // namespace ptx

struct pantax_hip_ctx : ptx::Ctx {};
struct pantax_hip_db : ptx::Db {};
struct pantax_hip_reads : ptx::Reads {};
