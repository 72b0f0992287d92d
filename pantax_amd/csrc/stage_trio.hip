// stage_trio.hip -- a7: the unique-trio index (trio_nodes_info, profile.rs:658-740) on device.
//
// Reference: every 3-window of every haplotype walk, canonicalised by swapping the ends when
// w[0] > w[2] (:672-678); count_per_trio counts every (hap, position) occurrence (:688-702); a
// trio is strain-specific ("unique") iff that count is exactly 1 (:708-716); its length is the
// sum of its three node lengths (:712).  The reference keeps a dense trio x hap presence matrix;
// a unique trio has exactly one owner, so an owner index per row carries the same information.
//
// Device plan (all species of the db in one batch; no sort needed):
//   1. count windows per MIDDLE node b (global node index)                     [4P in, atomics on 4V]
//   2. exclusive scan -> bucket offsets; scatter (q, b, c) into the buckets      [4P in, 12P out]
//   3. a window is unique iff no other entry of its (short) bucket has the same two ends (a,c)
//   4. exclusive scan of uniq_q over q -> row number in (species, hap, position) order
//   5. per-node counts of unique windows -> CSR lookup arrays (trio_first, trio_bc, trio_row)
//   6. fill the row-order arrays (abc, hap, len) and hap_trio_off
// Buckets are short (a node starts/ends a handful of windows per haplotype), so step 3 is a few
// compares per window; slot order inside a bucket is arbitrary but no output depends on it.
// Row order (species, hap, position) replaces the reference's FxHashSet iteration order, which
// is arbitrary; results are compared as keyed sets.
#include <algorithm>
#include <cstdlib>
#include "primitives.hpp"
#include "wave.hpp"
#include "scan_chained.hpp"

namespace ptx {

// Work decomposition of every per-path-step kernel: one workgroup per tile = PATH_TILE consecutive
// positions of ONE haplotype (tile table built at db upload).  Tiles are ordered (species, chunk, hap):
// the haplotypes of a species are largely collinear, so neighbouring workgroups touch the same node
// buckets at the same time (L2 write-combining of the bucket scatter) and no per-position search for the
// owning haplotype is needed.
#define TRIO_GRAPH_ARGS const uint2 *__restrict__ tiles, const uint64_t *__restrict__ path_off, const uint32_t *__restrict__ path_nodes, \
                        const uint32_t *__restrict__ hap_species, const uint32_t *__restrict__ node_base
#define TILE_LOOP(q, h, qend)                                                         \
    const uint2 tile__ = tiles[blockIdx.x];                                           \
    const bool pad__ = tile__.x == 0xFFFFFFFFu;   /* filler that keeps chunk groups XCD-aligned */ \
    const uint32_t h = pad__ ? 0u : tile__.x;                                         \
    const uint64_t qend = pad__ ? 0ull : path_off[h + 1];                             \
    const uint64_t qt0__ = pad__ ? 0ull : path_off[h] + (uint64_t)tile__.y * PATH_TILE; \
    for (uint64_t q = qt0__ + threadIdx.x; q < qt0__ + PATH_TILE && q < qend; q += 256)

// canonical window that STARTS at q of hap h (profile.rs:672-678: the ends are swapped when w[0] > w[2], the middle stays):
// (a, b, c) = (smaller end, middle, larger end); false if q starts no window.  Every occurrence of a window -- either
// orientation, any haplotype -- has the same MIDDLE node, so g = the global index of b is the key every table of this file is
// grouped by (round 4; rounds 1-3 grouped by the smaller end, which made a position own up to two windows).
__device__ __forceinline__ bool window_of(uint64_t q, uint64_t qend, uint32_t nb, const uint32_t *__restrict__ path_nodes, uint32_t &g,
                                          uint32_t &a, uint32_t &b, uint32_t &c) {
    if (q + 2 >= qend) return false;
    a = path_nodes[q]; b = path_nodes[q + 1]; c = path_nodes[q + 2];
    if (a > c) { uint32_t t = a; a = c; c = t; }
    g = nb + b;
    return true;
}

// 1. bucket sizes: windows per middle node
__global__ void __launch_bounds__(256) trio_count_kernel(TRIO_GRAPH_ARGS, uint32_t *__restrict__ cnt) {
    TILE_LOOP(q, h, qend) {
        const uint32_t nb = node_base[hap_species[h]];
        uint32_t g, a, b, c;
        if (window_of(q, qend, nb, path_nodes, g, a, b, c)) atomicAdd(&cnt[g], 1u);
    }
}
// uniq flags of the path positions: ONE BIT per position (a byte per position cost 8x the zero-fill before every build and
// 8x the reads of the two passes that rank the unique windows)
__device__ __forceinline__ void uniq_mark(uint32_t *__restrict__ bits, uint32_t q) { atomicOr(&bits[q >> 5], 1u << (q & 31u)); }
// 2. scatter windows into their bucket (slot order inside a bucket is arbitrary and irrelevant)
__global__ void __launch_bounds__(256) trio_fill_kernel(TRIO_GRAPH_ARGS, const uint32_t *__restrict__ bucket_off,
                                                        uint32_t *__restrict__ cursor, uint4 *__restrict__ bucket) {
    TILE_LOOP(q, h, qend) {
        const uint32_t nb = node_base[hap_species[h]];
        uint32_t g, a, b, c;
        if (!window_of(q, qend, nb, path_nodes, g, a, b, c)) continue;
        uint32_t slot = bucket_off[g] + atomicAdd(&cursor[g], 1u);
        bucket[slot] = make_uint4((uint32_t)q, a, c, g);   // one 16-byte record per window: {start position, smaller end, larger end, middle}
    }
}
// 3. a window is unique iff no other window of its bucket has the same (b,c): count == 1 (profile.rs:688-709).
//    Entries of a bucket are contiguous, so a wave compares its 64 consecutive entries through shuffles (one
//    16-byte load per window instead of one per pair); only the part of a bucket that lies outside the wave's
//    64 entries is read from memory.
__global__ void __launch_bounds__(256) trio_uniq_kernel(uint64_t n_win, const uint4 *__restrict__ bucket,
                                                        const uint32_t *__restrict__ bucket_off, uint32_t *__restrict__ uniq_q,
                                                        uint32_t *__restrict__ first_cnt) {
    const int lane = threadIdx.x & 63;
    for (uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) - lane; base < n_win; base += (uint64_t)gridDim.x * 256) {
        const uint64_t i = base + lane;
        const bool valid = i < n_win;
        uint4 me = make_uint4(0u, 0u, 0u, 0xFFFFFFFFu);
        if (valid) me = bucket[i];
        const uint32_t g = me.w;
        bool dup = false;
        // my place inside my bucket tells which lower lanes share it (entries of a bucket are contiguous): no id shuffle
        uint32_t b0 = 0, b1 = 0;
        if (valid) { b0 = bucket_off[g]; b1 = bucket_off[g + 1]; }
        const uint32_t below = valid ? (uint32_t)(i - b0) : 0u;         // entries of my bucket before me
        for (int d = 1; d < 64; ++d) {
            const bool same = valid && lane >= d && (uint32_t)d <= below;
            if (!__any(same)) break;   // no pair at distance d means none further apart
            const uint32_t oy = __shfl(me.y, lane - d), oz = __shfl(me.z, lane - d);
            const unsigned long long eq = __ballot(same && oy == me.y && oz == me.z);
            if ((eq >> lane) & 1ull) dup = true;                            // my partner is d below
            if (lane + d < 64 && ((eq >> (lane + d)) & 1ull)) dup = true;   // my partner is d above
        }
        if (valid && !dup) {
            const uint64_t wend = base + 64;
            for (uint64_t j = b0; j < b1 && j < base && !dup; ++j) { const uint4 o = bucket[j]; if (o.y == me.y && o.z == me.z) dup = true; }
            for (uint64_t j = (wend > b0 ? wend : b0); j < b1 && !dup; ++j) { const uint4 o = bucket[j]; if (o.y == me.y && o.z == me.z) dup = true; }
        }
        if (valid && !dup) { uniq_mark(uniq_q, me.x); atomicAdd(&first_cnt[g], 1u); }
    }
}
// 3'. the same test through an LDS hash table, for graphs where many haplotypes share their nodes (buckets of tens of
//     entries, nearly all of them equal): a workgroup owns the buckets that START inside its slice of UNIQ_CH
//     entries (bucket_off is searched for the two slice ends), so every bucket is seen whole by one workgroup.  Each
//     entry claims or joins the table slot of its (g, b, c) and counts itself there; unique <=> the count is 1.  O(1)
//     per window instead of O(bucket).  A slice that does not fit the LDS copy (one bucket of thousands of entries)
//     falls back to scanning the bucket in memory.
constexpr uint32_t UNIQ_CH = 1024, UNIQ_CAP = 1536, UNIQ_SLOTS = 4096;
__device__ __forceinline__ uint32_t uniq_hash(uint32_t g, uint32_t b, uint32_t c) {
    uint32_t h = g * 0x9E3779B1u;
    h = (h ^ b) * 0x85EBCA77u;
    h = (h ^ c) * 0xC2B2AE3Du;
    return (h ^ (h >> 15)) & (UNIQ_SLOTS - 1);
}
__global__ void __launch_bounds__(256) trio_uniq_lds_kernel(uint64_t n_win, uint32_t V, const uint4 *__restrict__ bucket,
                                                            const uint32_t *__restrict__ bucket_off, uint32_t *__restrict__ uniq_q,
                                                            uint32_t *__restrict__ first_cnt) {
    __shared__ uint32_t s_g[UNIQ_CAP], s_b[UNIQ_CAP], s_c[UNIQ_CAP];
    __shared__ uint32_t s_tab[UNIQ_SLOTS], s_cnt[UNIQ_SLOTS];
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    // first bucket start >= x (bucket_off[0..V] ascending, bucket_off[V] = n_win)
    auto first_start = [&](uint64_t x) -> uint64_t {
        if (x >= n_win) return n_win;
        uint32_t lo = 0, hi = V;                       // smallest v with bucket_off[v] >= x
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)bucket_off[mid] < x) lo = mid + 1; else hi = mid; }
        return bucket_off[lo];
    };
    const uint64_t lo = first_start((uint64_t)blockIdx.x * UNIQ_CH), hi = first_start((uint64_t)(blockIdx.x + 1) * UNIQ_CH);
    if (hi <= lo) return;
    const uint32_t n = (uint32_t)(hi - lo);
    if (n > UNIQ_CAP) {                                // oversized bucket(s): plain scan of each entry's bucket
        for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
            const uint4 me = bucket[i];
            bool dup = false;
            for (uint32_t j = bucket_off[me.w], e = bucket_off[me.w + 1]; j < e && !dup; ++j)
                if (j != i) { const uint4 o = bucket[j]; dup = o.y == me.y && o.z == me.z; }
            if (!dup) { uniq_mark(uniq_q, me.x); atomicAdd(&first_cnt[me.w], 1u); }
        }
        return;
    }
    for (uint32_t k = threadIdx.x; k < UNIQ_SLOTS; k += 256) { s_tab[k] = EMPTY; s_cnt[k] = 0; }
    uint32_t my_q[UNIQ_CAP / 256], my_slot[UNIQ_CAP / 256];
#pragma unroll
    for (int k = 0; k < (int)(UNIQ_CAP / 256); ++k) {
        const uint32_t t = threadIdx.x + k * 256;
        if (t < n) { const uint4 me = bucket[lo + t]; my_q[k] = me.x; s_b[t] = me.y; s_c[t] = me.z; s_g[t] = me.w; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (int)(UNIQ_CAP / 256); ++k) {
        const uint32_t t = threadIdx.x + k * 256;
        if (t >= n) continue;
        const uint32_t g = s_g[t], b = s_b[t], c = s_c[t];
        uint32_t h = uniq_hash(g, b, c);
        for (;;) {
            uint32_t cur = s_tab[h];
            if (cur == EMPTY) cur = atomicCAS(&s_tab[h], EMPTY, t);
            if (cur == EMPTY || (s_g[cur] == g && s_b[cur] == b && s_c[cur] == c)) break;
            h = (h + 1) & (UNIQ_SLOTS - 1);
        }
        atomicAdd(&s_cnt[h], 1u);
        my_slot[k] = h;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (int)(UNIQ_CAP / 256); ++k) {
        const uint32_t t = threadIdx.x + k * 256;
        if (t < n && s_cnt[my_slot[k]] == 1u) { uniq_mark(uniq_q, my_q[k]); atomicAdd(&first_cnt[s_g[t]], 1u); }
    }
}
// 3''. Uniqueness by node block, no global scatter (round 2's default; since round 4 the path of species that hold a node with
//      more than 64 visits -- everything else goes through the visit table below).  Every species' nodes are cut into blocks of
//      TRIO_BLK consecutive local ids; the walks were cut at upload into runs of consecutive positions inside one block
//      (trio_runs_build below).  A window is OWNED by the position of its MIDDLE node: position p owns (p-1, p, p+1), whose
//      canonical key is (min(n[p-1], n[p+1]), n[p], max(..)) (profile.rs:672-678) in either orientation.  So the wave of a block
//      meets EVERY occurrence of every window whose middle lies in the block -- all haplotypes -- and count_per_trio == 1
//      (profile.rs:688-709) is decided in an LDS hash table keyed by (middle - block start, smaller end, larger end) packed into
//      64 bits.  Collinear haplotypes collapse in LDS; HBM sees the walks once (4P) and one bit per UNIQUE window.  A block whose
//      distinct windows overflow the table is redone in 2, 4, ... sub-passes over disjoint key classes (exact: all occurrences
//      of a key fall into the same class).
constexpr int TRIO_BLK_SHIFT = 6, TRIO_BLK = 1 << TRIO_BLK_SHIFT;
constexpr unsigned long long TB_EMPTY = ~0ull;
constexpr uint32_t TB_MULTI = 0xFFFFFFFFu;
constexpr int TB_UNR = 4;
// slot = {64-bit key, u32 q}: q is the position of the window's only occurrence, or TB_MULTI once a second one arrived
// (both sides use atomicMax, so the outcome does not depend on who comes first; positions are < 2^32 - 1)
template <int TB_SLOTS>
__device__ __forceinline__ void tb_insert(unsigned long long *s_key, uint32_t *s_q, uint32_t *s_over, uint32_t a_l, uint32_t b, uint32_t c,
                                          uint32_t q, uint32_t sub_mask, uint32_t sub_j) {
    const unsigned long long key = ((unsigned long long)a_l << 54) | ((unsigned long long)b << 27) | c;
    // hash of the key from full-rate 24-bit multiplies (a 64-bit multiply is four quarter-rate ones, and this kernel is bound
    // by VALU issue): b and c are the block's neighbours, their low bits carry the entropy; the full key decides equality
    uint32_t mix = __umul24(b, 0x9E3779u) + __umul24(c, 0x85EBCBu) + __umul24(a_l, 0x27D4EBu);
    mix ^= mix >> 13;
    if (((mix >> 16) & sub_mask) != sub_j) return;
    uint32_t h = mix & (TB_SLOTS - 1);
    for (int probes = 0; probes < TB_SLOTS; ++probes) {
        unsigned long long cur = s_key[h];
        if (cur == TB_EMPTY) cur = atomicCAS(&s_key[h], TB_EMPTY, key);
        if (cur == TB_EMPTY) { atomicMax(&s_q[h], q); return; }
        if (cur == key) { s_q[h] = TB_MULTI; return; }    // plain store of the maximum: nothing can undo it
        h = (h + 1) & (TB_SLOTS - 1);
    }
    *s_over = 1u;
}
// ONE WAVE per block of TRIO_BLK nodes (a workgroup is one wave: no workgroup barrier anywhere, two dozen independent
// waves per CU hide each other's trips to memory; a 256-thread workgroup per 256-node block spent most of its life in
// barriers and fixed overhead).  blk_rec[gb] = {first run, end run, global index of the block's first node, its
// species-local id / 64 | the block's node count << 24} (the blocks of one launch need not be neighbours: only the species the
// visit table leaves to this kernel have any).
template <int TB_SLOTS>
__global__ void __launch_bounds__(64) trio_block_kernel(const uint4 *__restrict__ blk_rec, const uint4 *__restrict__ runs,
                                                        const uint32_t *__restrict__ path_nodes, uint32_t *__restrict__ uniq_q,
                                                        uint32_t *__restrict__ first_cnt, uint32_t *__restrict__ err) {
    __shared__ unsigned long long s_key[TB_SLOTS];
    __shared__ uint32_t s_q[TB_SLOTS], s_ncnt[TRIO_BLK], s_over, s_pref[64];
    __shared__ uint4 s_run[64];
    const uint4 rec = blk_rec[blockIdx.x];
    const uint32_t nn = rec.w >> 24;                                    // nodes of the block (the last block of a species holds fewer than 64)
    const uint32_t r0 = rec.x, r1 = rec.y, n0 = (rec.w & 0xFFFFFFu) << TRIO_BLK_SHIFT;
    const uint32_t lane = threadIdx.x;
    for (uint32_t nsub = 1;; nsub <<= 1) {
        s_ncnt[lane] = 0;
        bool over = false;
        for (uint32_t j = 0; j < nsub && !over; ++j) {
            for (int i = lane; i < TB_SLOTS; i += 64) { s_key[i] = TB_EMPTY; s_q[i] = 0; }
            if (lane == 0) s_over = 0;
            __syncthreads();
            // the block's runs go to LDS 64 at a time; their positions are then handed out flat over the wave, TB_UNR per
            // lane and round, all loads of a round issued before the first table operation
            for (uint32_t rb = r0; rb < r1; rb += 64) {
                const uint32_t n_r = r1 - rb < 64u ? r1 - rb : 64u;
                uint4 run = make_uint4(0u, 0u, 0u, 0u);
                if (lane < n_r) run = runs[rb + lane];
                const uint32_t incl = wave_incl_scan_dpp(run.y);
                const uint32_t total = __shfl(incl, 63);
                s_run[lane] = run; s_pref[lane] = incl - run.y;
                __syncthreads();
                uint32_t lo_carry = 0;   // run of the last flat index handed out so far: the indices only grow, so does the run
                for (uint32_t idx0 = lane; idx0 < total; idx0 += 64 * TB_UNR) {
                    uint32_t x[TB_UNR], pp[TB_UNR], b1[TB_UNR], c1[TB_UNR];
                    bool md[TB_UNR];
#pragma unroll
                    for (int u = 0; u < TB_UNR; ++u) {
                        const uint32_t idx = idx0 + u * 64;
                        md[u] = false;
                        x[u] = pp[u] = b1[u] = c1[u] = 0u;
                        uint32_t lo = lo_carry;
                        if (idx < total) {
                            // last run whose first flat index is <= idx: a short walk forward from the previous group's last run
                            // (runs are mostly longer than a wave, so a group of 64 indices crosses one or two run borders;
                            // entries past the block's runs hold `total` and stop the walk)
                            while (lo < 63u && s_pref[lo + 1] <= idx) ++lo;
                            const uint4 rn = s_run[lo];
                            const uint32_t pos = rn.x + (idx - s_pref[lo]);
                            pp[u] = pos;
                            x[u] = path_nodes[pos];
                            md[u] = pos > rn.z && pos + 1 < rn.w;     // the middle of a window: a neighbour on either side inside the walk
                            if (md[u]) { b1[u] = path_nodes[pos - 1]; c1[u] = path_nodes[pos + 1]; }
                        }
                        lo_carry = __shfl(lo, 63);   // lane 63 holds the group's largest index (or, past the end, the carry itself)
                    }
#pragma unroll
                    for (int u = 0; u < TB_UNR; ++u) {
                        if (md[u]) tb_insert<TB_SLOTS>(s_key, s_q, &s_over, x[u] - n0, min(b1[u], c1[u]), max(b1[u], c1[u]), pp[u] - 1, nsub - 1, j);   // flagged at the window's START
                    }
                }
                __syncthreads();   // s_run / s_pref are reused by the next 64 runs
            }
            over = s_over != 0;
            if (!over)
                for (int i = lane; i < TB_SLOTS; i += 64) {
                    const unsigned long long k = s_key[i];
                    const uint32_t q = s_q[i];
                    if (k != TB_EMPTY && q != TB_MULTI) { uniq_mark(uniq_q, q); atomicAdd(&s_ncnt[(uint32_t)(k >> 54)], 1u); }
                }
            __syncthreads();
        }
        if (!over) break;
        if (nsub >= (1u << 20)) { if (lane == 0) atomicAdd(err, 1u); break; }   // cannot happen short of 2^31 equal hashes; never silent
    }
    __syncthreads();
    if (lane < nn) first_cnt[rec.z + lane] = s_ncnt[lane];
}

// 3v. THE DEFAULT (round 4): uniqueness through the VISIT TABLE -- the walks transposed.  All occurrences of a window share
//     its middle node b (window_of above), so count_per_trio == 1 (profile.rs:688-709) is a question about the visits of b:
//     among all interior positions p with n[p] = b -- every haplotype of the species -- does the unordered pair
//     {n[p-1], n[p+1]} occur exactly once?  The visit table (trio_visits_build, once at upload: a layout of the walks like
//     the tile and run tables, the CSC to the walks' CSR) lists the interior positions node by node in groups of 64: a
//     node's visits never straddle a group (pads fill the tail), so ONE WAVE holds every visit of the handful of nodes of its
//     group; the visits of a node are kept sorted by their pair (an order of the table, like the order of an adjacency list), so
//     equal pairs are neighbours and lane l decides its window by comparing its pair with the lanes l - 1 and l + 1 of its node's
//     stretch -- no hash table, no LDS, no atomics, no loop on the way to the decision.  A step reads the table (4 B per visit) and
//     gathers the three consecutive walk entries of every visit (collinear haplotypes: the lanes of one haplotype read
//     neighbouring addresses, the lines are reused by the next groups of the wave).  A species that holds a node with more
//     than 64 interior visits (more than 64 haplotypes, or walks that keep returning to a node) is left to the node-block
//     kernel above; both write the same two outputs: one flag bit per window start and the count of unique windows per node.
constexpr uint32_t VIS_PAD = 0xFFFFFFFFu;
constexpr int VIS_MAX = 64;            // visits of a node that one wave decides
constexpr int VIS_CHUNK_SHIFT = 8;     // nodes per layout chunk (a chunk starts on a group border)
struct __attribute__((packed, aligned(4))) U32x3 { uint32_t x, y, z; };

// -DTV_ABLATE builds (never the product library) read PANTAX_TV_ABLATE: bit 0 no flag atomics, bit 1 no count stores, bit 2 no
// comparisons -- wrong results, for timing only
#ifdef TV_ABLATE
#define TV_ABL(bit) (ablate & (bit))
#else
#define TV_ABL(bit) false
#endif
// ROWS (the default): the wave hands its unique windows to trio_rows_kernel -- the group's ballot of unique visits and, for the first VIS_REC
// of them, a 16-byte record {window start, smaller end, larger end, middle (global node indices)}.  The rows of the index are then numbered IN
// THIS ORDER (round 5): a row = the rank of its visit among the unique visits of the table, i.e. a scan over the groups' counts (a tenth of the
// nodes) and one pass over the records -- no flag bit per path position, no ranks of flags, no scatter in (hap, position) order.  A group with more
// unique visits than records (one in seven at ten strains per species) is read again by trio_rows_kernel.
// !ROWS (option trio_rows=path; tests): one flag bit per unique window start + the count of unique windows per node, the inputs of the pass over
// the walks (trio_lookup_kernel) that also serves the species the visit table leaves to the node-block kernel.
constexpr int VIS_REC = 8;
template <int U, bool ROWS>
__global__ void __launch_bounds__(256) trio_visit_kernel(uint32_t NG, uint32_t rounds, const uint32_t *__restrict__ vis_pos, const uint64_t *__restrict__ vis_head,
                                                         const uint32_t *__restrict__ vis_nbase, const uint32_t *__restrict__ path_nodes,
                                                         uint32_t *__restrict__ uniq_q, uint32_t *__restrict__ first_cnt, uint32_t *__restrict__ err, uint32_t ablate,
                                                         unsigned long long *__restrict__ vis_uq, uint4 *__restrict__ vis_rec, uint32_t xcd_chunks) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // xcd_chunks != 0 (= the number of workgroups' worth of groups): workgroups go to the XCDs round-robin, so XCD x is given the x-th
    // contiguous eighth of the table -- neighbouring chunks read neighbouring lines of the same walks, and meet in ONE L2
    uint32_t blk = blockIdx.x;
    if (xcd_chunks) { blk = (blockIdx.x & 7u) * ((xcd_chunks + 7u) / 8u) + (blockIdx.x >> 3); if (blk >= xcd_chunks) blk = 0xFFFFFFu; }
    uint32_t g0 = blk == 0xFFFFFFu ? NG : (blk * 4u + wave) * ((uint32_t)U * rounds);      // this wave's U x rounds consecutive groups
    for (uint32_t r = 0; r < rounds && g0 < NG; ++r, g0 += U) {
        uint32_t q[U], nb[U];
        uint64_t heads[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + (uint32_t)u < NG ? g0 + (uint32_t)u : g0;   // wave-uniform
            q[u] = vis_pos[(uint64_t)g * 64 + lane];
            heads[u] = vis_head[g]; nb[u] = vis_nbase[g];
        }
        __builtin_amdgcn_sched_barrier(0);       // all U table loads leave before the first of them is waited for
        U32x3 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            valid[u] = g0 + (uint32_t)u < NG && q[u] != VIS_PAD;
            w[u] = *reinterpret_cast<const U32x3 *>(path_nodes + (valid[u] ? q[u] - 1u : 0u));   // an interior position: p - 1 and p + 1 exist
        }
        __builtin_amdgcn_sched_barrier(0);       // ... and all U gathers before the first decision
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t lo = min(w[u].x, w[u].z), hi = max(w[u].x, w[u].z);
            const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid[u]);
            const unsigned long long hd = heads[u] & vmask;
            // The visits of a node are SORTED by (smaller end, larger end) -- the table's order, fixed at upload -- so equal pairs sit in
            // neighbouring lanes: a window occurs once iff its pair differs from the pair of the lane below AND of the lane above
            // inside its node's stretch.  The order itself is checked on the way (a table that is not sorted is reported, never
            // silently trusted): one DPP shift and three compares per visit, no loop over the stretch.
            const unsigned long long inb = vmask & ~hd;                      // lanes with a lane of their own stretch below them
            const uint32_t slo = wave_shr1z(lo), shi = wave_shr1z(hi);       // the pair of the lane below (DPP moves)
            unsigned long long eq = __builtin_amdgcn_ballot_w64(slo == lo && shi == hi) & inb;
            if (TV_ABL(4u)) eq = 0ull;
            const unsigned long long bad = __builtin_amdgcn_ballot_w64(slo > lo || (slo == lo && shi > hi)) & inb;
            if (bad && lane == 0) atomicAdd(err, 1u);
            const unsigned long long dup = eq | (eq >> 1);                   // both partners are not unique
            const unsigned long long uq = vmask & ~dup;
            if (ROWS) {
                const uint32_t g = g0 + (uint32_t)u;
                if (g < NG) {
                    if (lane == 0) vis_uq[g] = uq;
                    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(uq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)uq, 0u));   // unique visits in the lanes below
                    if (((uq >> lane) & 1ull) && rk < (uint32_t)VIS_REC && !TV_ABL(2u))
                        vis_rec[(uint64_t)g * VIS_REC + rk] = make_uint4(q[u] - 1u, nb[u] + lo, nb[u] + hi, nb[u] + w[u].y);
                }
            } else {
                if (((uq >> lane) & 1ull) && !TV_ABL(1u)) uniq_mark(uniq_q, q[u] - 1u);          // flagged at the window's start
                if (((hd >> lane) & 1ull) && !TV_ABL(2u)) {                                      // the head lane stores its node's count of unique windows
                    const unsigned long long he = hd | (~vmask & (vmask + 1ull));   // the first pad lane closes the last stretch (pads sit at the tail)
                    const unsigned long long above = he & ~((2ull << lane) - 1ull);
                    const int end = above ? __builtin_ctzll(above) : 64;
                    const unsigned long long m = (end == 64 ? ~0ull : (1ull << end) - 1ull) & ~((1ull << lane) - 1ull);
                    first_cnt[nb[u] + w[u].y] = (uint32_t)__popcll(uq & m);
                }
            }
        }
    }
}

// ---- the visit table (upload time; a function of the graphs alone) ----
// interior visits per node (a position with a neighbour on either side inside its walk is the middle of one window)
__global__ void __launch_bounds__(256) visit_count_kernel(TRIO_GRAPH_ARGS, uint32_t *__restrict__ cnt) {
    TILE_LOOP(q, h, qend) {
        if (q > path_off[h] && q + 1 < qend) atomicAdd(&cnt[node_base[hap_species[h]] + path_nodes[q]], 1u);
    }
}
// visited[v] = the node has a visit (its count is stored by every build; the others read as zero), slow[s] = the species holds a
// node with more than VIS_MAX visits
__global__ void __launch_bounds__(256) visit_flags_kernel(uint64_t V, uint32_t S, const uint32_t *__restrict__ node_base, const uint32_t *__restrict__ cnt,
                                                          uint32_t *__restrict__ visited, uint32_t *__restrict__ slow) {
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = v < V ? cnt[v] : 0u;
    const unsigned long long bal = __ballot(c != 0u);
    if ((threadIdx.x & 31) == 0 && v < V + 32) visited[v >> 5] = (uint32_t)(bal >> (threadIdx.x & 32));
    if (c > (uint32_t)VIS_MAX) {
        uint32_t lo = 0, hi = S;                                             // last s with node_base[s] <= v
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)node_base[mid] <= v) lo = mid; else hi = mid; }
        slow[lo] = 1u;
    }
}
// one thread packs the nodes of a chunk {first node, end node, node base of the species, species} into groups of 64 visits that no node
// straddles: FIRST FIT over a few open groups (round 5; rounds 4's next-fit closed a group as soon as the next node did not fit -- one
// 50-visit node per group at fifty strains per species, 22 % pads; 7 % at ten).  The order of the nodes inside a chunk is then the order of
// their placement, not of their ids: nothing depends on it (a node's visits stay one stretch of one group, its rows one block).
constexpr int VIS_OPEN = 4;
struct VisPack {
    uint32_t fill[VIS_OPEN], gidx[VIS_OPEN], n_groups;
    unsigned long long heads[VIS_OPEN];                  // head lanes of the open groups (bit = a node's first visit)
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < VIS_OPEN; ++j) { fill[j] = 64u; gidx[j] = 0xFFFFFFFFu; heads[j] = 0ull; }
        n_groups = 0u;
    }
    // -> slot of the node's first visit, relative to the chunk's first group.  A group that is closed to make room is handed to `closed`
    // (group index relative to the chunk, its head mask): a group belongs to ONE chunk, so its mask is a plain store of the packing thread
    template <class Closed>
    __device__ __forceinline__ uint32_t place(uint32_t k, Closed &&closed) {
        int best = -1;
#pragma unroll
        for (int j = VIS_OPEN - 1; j >= 0; --j) if (fill[j] + k <= 64u) best = j;       // the first open group it fits
        if (best < 0) {                                                                 // none: the fullest one is closed, a new group opened in its place
            best = 0;
#pragma unroll
            for (int j = 1; j < VIS_OPEN; ++j) if (fill[j] > fill[best]) best = j;
#pragma unroll
            for (int j = 0; j < VIS_OPEN; ++j) if (j == best) { if (gidx[j] != 0xFFFFFFFFu) closed(gidx[j], heads[j]); fill[j] = 0u; gidx[j] = n_groups; heads[j] = 0ull; }
            ++n_groups;
        }
        uint32_t slot = 0;
#pragma unroll
        for (int j = 0; j < VIS_OPEN; ++j) if (j == best) { slot = gidx[j] * 64u + fill[j]; heads[j] |= 1ull << fill[j]; fill[j] += k; }
        return slot;
    }
    template <class Closed>
    __device__ __forceinline__ void finish(Closed &&closed) {
#pragma unroll
        for (int j = 0; j < VIS_OPEN; ++j) if (gidx[j] != 0xFFFFFFFFu) closed(gidx[j], heads[j]);
    }
};
// PLACE = false: the number of groups every chunk needs (-> scan -> first group of every chunk); PLACE = true: the nodes' slots, the groups' head
// masks / node bases / species.  A workgroup of 64 threads takes 64 consecutive chunks: the wave loads their nodes' counts into LDS (coalesced; a
// count of the visit table's species is at most 64: a byte), every thread then packs ITS chunk from LDS, and the wave writes the slots back
// coalesced.  (Round 4-5's first version had every thread read its chunk's counts from memory, 1 KB apart from its neighbour's: 81 + 37 GB of
// sector traffic for 1.3 GB of counts at 1e4 strains, 22 + 4.5 ms.)
constexpr int VP_CHUNKS = 64, VP_CNT_STRIDE = 260 /* bytes */, VP_SLOT_STRIDE = 258 /* u16: 129 words -> the threads' rows start on different banks */;
template <bool PLACE>
__global__ void __launch_bounds__(64) visit_pack_kernel(uint32_t NC, const uint4 *__restrict__ chunks, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ chunk_groups,
                                                        const uint32_t *__restrict__ chunk_gbase, uint32_t *__restrict__ vslot, unsigned long long *__restrict__ head,
                                                        uint32_t *__restrict__ gnbase, uint32_t *__restrict__ gsp) {
    __shared__ uint8_t s_cnt[VP_CHUNKS * VP_CNT_STRIDE];
    __shared__ uint16_t s_slot[PLACE ? VP_CHUNKS * VP_SLOT_STRIDE : 1];
    const uint32_t c0 = blockIdx.x * VP_CHUNKS, lane = threadIdx.x;
    const uint32_t nj = min((uint32_t)VP_CHUNKS, NC - c0);
    for (uint32_t j = 0; j < nj; ++j) {
        const uint4 ch = chunks[c0 + j];                                         // (workgroup-uniform)
        for (uint32_t i = lane; i < ch.y - ch.x; i += 64) s_cnt[j * VP_CNT_STRIDE + i] = (uint8_t)min(cnt[ch.x + i], 255u);
    }
    __syncthreads();
    {
        uint32_t n = 0;
        if (lane < nj) { const uint4 ch = chunks[c0 + lane]; n = ch.y - ch.x; }
        VisPack pk;
        pk.init();
        // a group's head mask: one plain 8-byte store by the packing thread when the group is closed (the first version issued one memory-side
        // atomicOr per NODE: 3.2e8 at 1e4 strains, 15 of the kernel's 17 ms)
        const uint32_t gb = (PLACE && lane < nj) ? chunk_gbase[c0 + lane] : 0u;
        auto closed = [&](uint32_t g, unsigned long long m) { if (PLACE) head[gb + g] = m; };
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t k = s_cnt[lane * VP_CNT_STRIDE + i];
            if (!k) continue;
            const uint32_t slot = pk.place(k, closed);                           // relative to the chunk's first group: below 256 groups x 64
            if (PLACE) s_slot[lane * VP_SLOT_STRIDE + i] = (uint16_t)slot;
        }
        pk.finish(closed);
        if (!PLACE) { if (lane < nj) chunk_groups[c0 + lane] = pk.n_groups; return; }
    }
    __syncthreads();
    for (uint32_t j = 0; j < nj; ++j) {
        const uint4 ch = chunks[c0 + j];
        const uint32_t gbase = chunk_gbase[c0 + j], base = gbase << 6, ng = chunk_gbase[c0 + j + 1] - gbase;
        for (uint32_t i = lane; i < ch.y - ch.x; i += 64)
            if (s_cnt[j * VP_CNT_STRIDE + i]) vslot[ch.x + i] = base + s_slot[j * VP_SLOT_STRIDE + i];
        for (uint32_t g = lane; g < ng; g += 64) { gnbase[gbase + g] = ch.z; gsp[gbase + g] = ch.w; }   // the chunk's groups: its species' node base / species
    }
}
__global__ void __launch_bounds__(256) visit_fill_kernel(TRIO_GRAPH_ARGS, const uint32_t *__restrict__ slow, const uint32_t *__restrict__ vslot,
                                                         uint32_t *__restrict__ cnt /* counted back down to zero */, uint32_t *__restrict__ vis_pos) {
    TILE_LOOP(q, h, qend) {
        const uint32_t sp = hap_species[h];
        if (slow[sp] || !(q > path_off[h] && q + 1 < qend)) continue;
        const uint32_t g = node_base[sp] + path_nodes[q];
        vis_pos[vslot[g] + atomicSub(&cnt[g], 1u) - 1u] = (uint32_t)q;
    }
}
// the visits of every node sorted by (smaller end, larger end, position) of their window (the fill's atomics left them in arrival
// order): equal windows become neighbours, which is what trio_visit_kernel's two-neighbour test relies on -- and checks -- and the
// table is the same on every upload.  One wave per group, ranks by shuffles inside the node's stretch.
__global__ void __launch_bounds__(256) visit_sort_kernel(uint32_t NG, uint32_t *__restrict__ vis_pos, const uint64_t *__restrict__ vis_head,
                                                         const uint32_t *__restrict__ path_nodes) {
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= NG) return;
    const int lane = threadIdx.x & 63;
    const uint32_t q = vis_pos[(uint64_t)g * 64 + lane];
    const bool valid = q != VIS_PAD;
    uint32_t lo = 0, hi = 0;
    if (valid) { const U32x3 w = *reinterpret_cast<const U32x3 *>(path_nodes + (q - 1u)); lo = min(w.x, w.z); hi = max(w.x, w.z); }
    const unsigned long long vmask = __ballot(valid), hd = vis_head[g] & vmask;
    const unsigned long long he = hd | (~vmask & (vmask + 1ull));
    const unsigned long long upto = hd & ((2ull << lane) - 1ull), above = he & ~((2ull << lane) - 1ull);
    const int start = upto ? 63 - __builtin_clzll(upto) : lane, end = above ? __builtin_ctzll(above) : 64;
    // LONG stretches (a node of dozens of visits: fifty strains per species): the ranks below cost a round per distance, 49 of them -- the whole wave
    // is sorted instead by (stretch, smaller end, larger end, position) in a bitonic network of 21 exchanges, pads (stretch 64) last: a lane's
    // sorted place IS its slot, because the stretches are the wave's lanes in order (115 -> 70 ms per db of 2.8e9 path steps)
    if (__builtin_amdgcn_ballot_w64(valid && end - start > 24) != 0ull) {
        uint32_t k0 = valid ? (uint32_t)start : 64u, k1 = lo, k2 = hi, k3 = q;
        for (int k = 2; k <= 64; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int partner = lane ^ j;
                const uint32_t p0 = __shfl(k0, partner), p1 = __shfl(k1, partner), p2 = __shfl(k2, partner), p3 = __shfl(k3, partner);
                const bool p_less = p0 < k0 || (p0 == k0 && (p1 < k1 || (p1 == k1 && (p2 < k2 || (p2 == k2 && p3 < k3)))));
                const bool keep_min = ((lane & k) == 0) == (lane < partner);     // ascending blocks keep the smaller key in the lower lane
                const bool take = keep_min ? p_less : !p_less;                   // (keys are distinct: positions differ; pads equal each other -- either stays)
                if (take && !(p0 == k0 && p1 == k1 && p2 == k2 && p3 == k3)) { k0 = p0; k1 = p1; k2 = p2; k3 = p3; }
            }
        if (k0 != 64u) vis_pos[(uint64_t)g * 64 + lane] = k3;
        return;
    }
    int rank = 0;
    // every pair of a stretch is compared ONCE, by its upper lane (positions are distinct: the order is total and strict); the lower lane reads the
    // outcome from the ballot -- three shuffles per distance instead of six (28.8 ms at 1e4 strains, 213 ms per db at fifty strains per species before)
    for (int d = 1; d < 64; ++d) {
        const bool dn = valid && lane - d >= start;
        if (!__any(dn)) break;
        const int ld = (lane - d) & 63;
        const uint32_t alo = __shfl(lo, ld), ahi = __shfl(hi, ld), aq = __shfl(q, ld);
        const bool below_first = alo < lo || (alo == lo && (ahi < hi || (ahi == hi && aq < q)));
        const unsigned long long mine_first = __ballot(dn && !below_first);      // bit l: lane l sorts before its partner l - d
        if (dn && below_first) ++rank;
        if (valid && lane + d < end && ((mine_first >> ((lane + d) & 63)) & 1ull)) ++rank;
    }
    if (valid) vis_pos[(uint64_t)g * 64 + start + rank] = q;   // every lane holds its value already: the stretch is rewritten in place
}

// The run table (upload time, depends on the graphs only): heads = positions whose node lies in another block than their
// predecessor's (or that start a walk); counted per block, scanned, then every head measures its run and files it.
__global__ void __launch_bounds__(256) run_count_kernel(TRIO_GRAPH_ARGS, const uint32_t *__restrict__ slow, const uint32_t *__restrict__ blk_base, uint32_t *__restrict__ blk_cnt) {
    TILE_LOOP(q, h, qend) {
        const uint32_t sp = hap_species[h], x = path_nodes[q];
        if (!slow[sp]) continue;                       // the visit table's species
        const bool head = q == path_off[h] || (path_nodes[q - 1] >> TRIO_BLK_SHIFT) != (x >> TRIO_BLK_SHIFT);
        if (head) atomicAdd(&blk_cnt[blk_base[sp] + (x >> TRIO_BLK_SHIFT)], 1u);
    }
}
__global__ void __launch_bounds__(256) run_fill_kernel(TRIO_GRAPH_ARGS, const uint32_t *__restrict__ slow, const uint32_t *__restrict__ blk_base, const uint32_t *__restrict__ blk_run_off,
                                                       uint32_t *__restrict__ cursor, uint4 *__restrict__ runs) {
    TILE_LOOP(q, h, qend) {
        const uint32_t sp = hap_species[h], x = path_nodes[q], bx = x >> TRIO_BLK_SHIFT;
        if (!slow[sp]) continue;
        const uint64_t qb = path_off[h];
        const bool head = q == qb || (path_nodes[q - 1] >> TRIO_BLK_SHIFT) != bx;
        if (!head) continue;
        uint64_t e = q + 1;
        while (e < qend && (path_nodes[e] >> TRIO_BLK_SHIFT) == bx) ++e;
        const uint32_t gb = blk_base[sp] + bx;
        runs[blk_run_off[gb] + atomicAdd(&cursor[gb], 1u)] = make_uint4((uint32_t)q, (uint32_t)(e - q), (uint32_t)qb, (uint32_t)qend);
    }
}

// ---- rows of the index ----------------------------------------------------------------------------------------------------------------
// A ROW is a unique window; its number is the place it is FILED at (round 5; rounds 1-4 numbered the rows in (species, hap, position) order,
// which cost a flag bit per path position, the ranks of those flags and a scattered store per row: 15.5 GB of traffic for 6 GB of payload at
// 1e4 strains).  Everything the step reads is indexed by that number: the lookup entry {smaller end, larger end} (the coverage pass finds a
// window under its MIDDLE node, whose record carries {first row, #rows}), the window's length (profile.rs:712), the haplotype that owns it,
// and the coverage pass's trio_bases.  The rows of a node are neighbours, sorted by their pair of ends -- a canonical order, the same on
// every build and on both routes below -- and the rows of a species are one block.

// the haplotype whose walk holds path position q: last h in [h0, h1) with path_off[h] <= q
__device__ __forceinline__ uint32_t hap_of_position(const uint64_t *__restrict__ path_off, uint32_t h0, uint32_t h1, uint32_t q) {
    uint32_t lo = h0, hi = h1;
    while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (path_off[mid] <= (uint64_t)q) lo = mid; else hi = mid; }
    return lo;
}
// what filing a row writes (dense stores in row order) -- KEYS: also the window start, from which the exporters make the
// (species, hap, position) order; FIRST (first build of a db): the rows per haplotype are counted (-> hap_trio_off)
struct RowOut {
    const uint32_t *node_len;
    const uint64_t *path_off, *hap_off;
    uint2 *ent;
    trio_len_t *len;
    uint16_t *hap;
    uint32_t *q;
    uint32_t *hap_cnt;
    __device__ __forceinline__ void put_len_hap(uint32_t row, uint32_t l, uint32_t h) const {
#if TRIO_LH_PACK
        len[row] = make_uint2(l, h);
#else
        len[row] = l; hap[row] = (uint16_t)h;
#endif
    }
};
// FIRST builds count the rows per haplotype.  One memory-side atomic per row on the ten counters of the species every wave of the GPU is filing at that
// moment took 104 ms at 1e4 strains (1.8e8 adds on 1e4 addresses, `r05_cfg4_kernel_stats`): a workgroup of the rows kernel counts in an LDS window of
// 1024 haplotypes from the species of its first group on (groups are in species order) and adds what it counted once, at its end.
constexpr uint32_t HAPCNT_WIN = 1024;
struct HapCount {
    uint32_t *lds;       // [HAPCNT_WIN] or null: straight to memory
    uint32_t base;       // global haplotype of lds[0]
    uint32_t *glob;
    __device__ __forceinline__ void add(uint32_t h) const {
        const uint32_t rel = h - base;
        if (lds && rel < HAPCNT_WIN) atomicAdd(&lds[rel], 1u); else atomicAdd(&glob[h], 1u);
    }
};
template <bool KEYS, bool FIRST>
__device__ __forceinline__ void row_file(const RowOut &o, uint32_t row, uint32_t q0, uint32_t lo, uint32_t hi, uint32_t mid, uint32_t sp, const HapCount &hc) {
    const uint32_t h0 = (uint32_t)o.hap_off[sp], h = hap_of_position(o.path_off, h0, (uint32_t)o.hap_off[sp + 1], q0);
    o.ent[row] = make_uint2(lo, hi);
    o.put_len_hap(row, o.node_len[lo] + o.node_len[mid] + o.node_len[hi], h - h0);
    if (KEYS) o.q[row] = q0;
    if (FIRST) hc.add(h);
}

// ---- PATH ROUTE: rows filed by a pass over the walks (species the visit table leaves to the node-block kernel; whole databases on the
// bucket path or under the options trio_path / trio_rows).  Its inputs are one flag bit per unique window start and the count of unique
// windows per node; a scan of the counts gives every node its block of rows (and its lookup head), the pass over the walks drops every unique
// window into its node's block in ARRIVAL order, and trio_canon_kernel then puts every block into the canonical order -- sorted by the pair
// of ends, which is distinct inside a node by the very definition of a unique window -- and files the rows.
// `only_slow` (mixed databases): only the tiles of the species left to the node-block kernel.
__global__ void __launch_bounds__(256) trio_lookup_kernel(TRIO_GRAPH_ARGS, const uint32_t *__restrict__ uniq_q, const uint32_t *__restrict__ trio_first,
                                                          uint32_t *__restrict__ cursor /* = the per-node counts; zero afterwards */, uint2 *__restrict__ trio_ent,
                                                          uint32_t *__restrict__ row_q, const uint32_t *__restrict__ only_slow) {
    constexpr int NR = PATH_TILE / 256;   // rounds of 256 consecutive positions
    __shared__ uint32_t s_wave[NR][4];
    const uint2 tile = tiles[blockIdx.x];
    if (tile.x == 0xFFFFFFFFu) return;   // filler tile
    const uint32_t h = tile.x;
    const uint64_t qend = path_off[h + 1], qt0 = path_off[h] + (uint64_t)tile.y * PATH_TILE;
    const uint32_t sidx = hap_species[h], nbase = node_base[sidx];
    if (only_slow && !only_slow[sidx]) return;                   // a species of the visit table
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // all rounds at once: the flags of the four rounds are loaded together, ONE barrier orders the wave counts, and the
    // gathers / writes of the unique windows of all rounds are in flight together
    uint32_t u[NR];
    unsigned long long bal[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const uint64_t q = qt0 + (uint64_t)r * 256 + threadIdx.x;
        u[r] = (q < qend) ? (uniq_q[q >> 5] >> (uint32_t)(q & 31ull)) & 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bal[r] = __ballot(u[r] != 0);
        if (lane == 0) s_wave[r][wave] = (uint32_t)__popcll(bal[r]);
    }
    __syncthreads();
    // The unique windows are a few per cent of the positions: they are compacted into an LDS list first, and the gathers / scatters of a
    // window then run on DENSE lanes
    __shared__ uint16_t s_list[PATH_TILE];
    uint32_t n_u = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t t = s_wave[r][w]; if (w < wave) woff += t; tot += t; }
        if (u[r]) s_list[n_u + woff + (uint32_t)__popcll(bal[r] & ((1ull << lane) - 1ull))] = (uint16_t)(r * 256 + (int)threadIdx.x);
        n_u += tot;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_u; t += 256) {
        const uint64_t q = qt0 + s_list[t];
        uint32_t g, a, b, c;
        window_of(q, qend, nbase, path_nodes, g, a, b, c);
        const uint32_t j = trio_first[g] + atomicSub(&cursor[g], 1u) - 1u;   // the node's own count, counted down: no cursor array to zero
        trio_ent[j] = make_uint2(nbase + a, nbase + c);                      // global node indices: the coverage pass works in them throughout
        row_q[j] = (uint32_t)q;
    }
}
// one thread per node that heads rows (path route): its block of rows sorted by (smaller end, larger end) -- insertion sort, a handful of rows;
// a hub of a thousand distinct neighbour pairs is a millisecond of one thread at load time -- and filed
template <bool KEYS, bool FIRST>
__global__ void __launch_bounds__(256) trio_canon_kernel(uint64_t V, uint32_t S, const uint32_t *__restrict__ node_base, const uint4 *__restrict__ node_rec,
                                                         const uint32_t *__restrict__ trio_first, const uint32_t *__restrict__ visited,
                                                         const uint32_t *__restrict__ only_slow, uint32_t *__restrict__ row_q, RowOut o) {
    const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    if (visited && !((visited[v >> 5] >> (uint32_t)(v & 31ull)) & 1u)) return;
    const uint32_t n = nr_rows(node_rec[v].y);
    if (n == 0) return;
    uint32_t lo = 0, hi = S;                                             // last s with node_base[s] <= v
    while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)node_base[mid] <= v) lo = mid; else hi = mid; }
    const uint32_t sp = lo;
    if (only_slow && !only_slow[sp]) return;                             // a species of the visit table: trio_rows_kernel files its rows
    const uint32_t f = trio_first[v];
    for (uint32_t i = 1; i < n; ++i) {
        const uint2 e = o.ent[f + i];
        const uint32_t q = row_q[f + i];
        uint32_t j = i;
        for (; j > 0; --j) {
            const uint2 p = o.ent[f + j - 1];
            if (p.x < e.x || (p.x == e.x && p.y <= e.y)) break;
            o.ent[f + j] = p; row_q[f + j] = row_q[f + j - 1];
        }
        if (j != i) { o.ent[f + j] = e; row_q[f + j] = q; }
    }
    for (uint32_t i = 0; i < n; ++i) {
        const uint2 e = o.ent[f + i];
        row_file<KEYS, FIRST>(o, f + i, row_q[f + i], e.x, e.y, (uint32_t)v, sp, HapCount{nullptr, 0u, o.hap_cnt});
    }
}

// scan of the per-node unique-window counts that also writes the lookup heads {first row, #rows} (CSR over the
// middle node) -- the prefix and its consumer in one launch
// `visited` (visit-table / node-block builds): a node without an interior visit is the middle of no window and no kernel of the
// build stores its count -- it reads as zero here instead of being zero-filled before every build (4V bytes)
struct TrioFirstLoad {
    const uint32_t *cnt, *visited;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
        if (visited && !((visited[i >> 5] >> (uint32_t)(i & 31ull)) & 1u)) return 0u;
        return cnt[i];
    }
};
// mixed databases: the lookup heads of the species left to the node-block kernel, filed BEHIND the rows of the visit table's species (row base
// = *u_fast, the total of the groups' counts); a node of a visit-table species reads as zero here
struct SlowFirstLoad {
    const uint32_t *cnt, *visited, *slow;
    const uint2 *tile_sp;
    const uint32_t *node_base;
    uint64_t V;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
        if (i >= V) return 0u;
        const uint2 t = tile_sp[i >> 11];
        uint32_t sp = t.x;
        while (sp < t.y && node_base[sp + 1] <= i) ++sp;
        if (!slow[sp] || !((visited[i >> 5] >> (uint32_t)(i & 31ull)) & 1u)) return 0u;
        return cnt[i];
    }
};
struct SlowFirstStore {
    uint32_t *first;
    uint4 *node_rec;
    uint64_t V;
    const uint32_t *u_fast;
    uint32_t *err;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t c) const {
        if (i < V && c) {                                  // (trio_first is written for the nodes that have rows: what the lookup pass reads)
            const uint32_t f = *u_fast + excl;
            first[i] = f;
            if (c >= NODE_REC_MAX_ROWS) atomicAdd(err, 1u);
            uint4 r = node_rec[i];
            const uint32_t y_new = nr_head(r.y, c, 0xFFu);
            if (r.y != y_new || r.w != f) { r.y = y_new; r.w = f; node_rec[i] = r; }
        }
    }
};
struct TrioFirstStore {
    uint32_t *first;
    uint4 *node_rec;   // the head {first row, #rows} rides in the node record the coverage kernel gathers anyway
    uint64_t V;
    uint32_t *err;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t c) const {
        first[i] = excl;
        // Only nodes that head unique windows carry a lookup head (8 % of them): which nodes those are and how many rows they head
        // is a function of the graphs alone, so every rebuild writes the same values -- a node without rows keeps the "0 rows" of its
        // upload-time record and is not touched (round 2 read and rewrote all V records per build: 32 bytes of traffic per node).
        if (i < V && c) {
            if (c >= NODE_REC_MAX_ROWS) atomicAdd(err, 1u);
            uint4 r = node_rec[i];
            const uint32_t y_new = nr_head(r.y, c, 0xFFu);   // this route does not compute the pair filter
            if (r.y != y_new || r.w != excl) { r.y = y_new; r.w = excl; node_rec[i] = r; }
        }
    }
};

// ---- FAST ROUTE: the rows of the species the visit table covers, filed from trio_visit_kernel<.., ROWS = true>'s records ----
// the first row of every group = the prefix of the groups' counts of unique visits, in three plain launches (tile sums, a scan of the sums
// by one workgroup, tile prefixes): a chained scan's workgroups spin on their predecessors, and beside the main stream's kernels of the step in
// flight that spinning stretched a 0.34-ms scan to 2.5 ms (and held the slots it spun in) -- the rebuild of the NEXT step runs beside the
// current step's row sort and LPs
struct GroupCountLoad { const unsigned long long *uq; __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return (uint32_t)__popcll(uq[i]); } };
struct PrefixStore { uint32_t *out; __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t) const { out[i] = excl; } };
constexpr int FR_WORDS = 16, FR_TILE = 256 * FR_WORDS;           // counts per thread and per workgroup
__global__ void __launch_bounds__(1024) tile_scan_kernel(uint32_t *__restrict__ sums, uint32_t n_tiles, uint32_t *__restrict__ total) {
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x, v = i < n_tiles ? sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_wave, &tot);
        if (i < n_tiles) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}
// (element i of a tile = stretch k, thread t: i = k * 256 + t -- coalesced loads and stores; a stretch's prefix = one DPP scan per wave
// + the four wave sums through LDS)
template <class Count, class Emit>
__device__ __forceinline__ void tile_prefix(uint64_t n, uint32_t start, Count count, Emit emit) {
    __shared__ uint32_t s_w[2][4];
    const uint64_t base = (uint64_t)blockIdx.x * FR_TILE;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = start;
#pragma unroll 4
    for (int k = 0; k < FR_WORDS; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        const uint32_t c = i < n ? count(i) : 0u;
        const uint32_t incl = wave_incl_scan_dpp(c);
        if (lane == 63) s_w[k & 1][wave] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t t = s_w[k & 1][w]; woff += w < (int)wave ? t : 0u; tot += t; }
        if (i < n) emit(i, run + woff + incl - c);
        run += tot;
    }
}
__global__ void __launch_bounds__(256) group_tile_sum_kernel(const unsigned long long *__restrict__ uq, uint64_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * FR_TILE;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < FR_WORDS; ++k) { const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x; if (i < n) c += (uint32_t)__popcll(uq[i]); }
    c = wave_reduce(c, [](uint32_t x, uint32_t y) { return x + y; });
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(256) group_tile_prefix_kernel(const unsigned long long *__restrict__ uq, uint64_t n, const uint32_t *__restrict__ sums, uint32_t *__restrict__ out) {
    tile_prefix(n, sums[blockIdx.x], [&](uint64_t i) { return (uint32_t)__popcll(uq[i]); }, [&](uint64_t i, uint32_t excl) { out[i] = excl; });
}

// the lookup head of a node = {its first row, the number of its rows}, written into the node record by the lane that holds the
// node's first unique window (records arrive in visit order: a node's windows are neighbours); `cnt` = rows of this node
__device__ __forceinline__ void trio_head_store(uint4 *__restrict__ node_rec, uint32_t v, uint32_t row, uint32_t cnt, uint32_t filter, uint32_t *__restrict__ err) {
    if (cnt >= NODE_REC_MAX_ROWS) atomicAdd(err, 1u);
    uint4 r = node_rec[v];
    const uint32_t y_new = nr_head(r.y, cnt, filter);
    if (r.y != y_new || r.w != row) { r.y = y_new; r.w = row; node_rec[v] = r; }   // (stored only where it is not there yet: see trio_rows_kernel)
}
// a group with more than VIS_REC unique visits (a stretch of private sequence; every group of a single-strain species): the whole wave
// reads the group's visits again, ranks the unique ones, and files them like the records
template <bool KEYS, bool FIRST>
__device__ __forceinline__ void trio_rows_group(uint32_t g, int lane, const unsigned long long *__restrict__ vis_uq, const uint32_t *__restrict__ gprefix,
                                                const uint32_t *__restrict__ vis_pos, const uint32_t *__restrict__ vis_nbase, const uint32_t *__restrict__ vis_sp,
                                                const uint32_t *__restrict__ path_nodes, uint4 *__restrict__ node_rec, const RowOut &o, uint32_t *__restrict__ err,
                                                const HapCount &hc) {
    const unsigned long long uq = vis_uq[g];
    const uint32_t nb = vis_nbase[g], sp = vis_sp[g], base = gprefix[g];
    const bool mine = (uq >> lane) & 1ull;
    uint4 rec = make_uint4(0u, 0u, 0u, 0xFFFFFFFFu);
    if (mine) {
        const uint32_t q = vis_pos[(uint64_t)g * 64 + lane];
        const U32x3 w = *reinterpret_cast<const U32x3 *>(path_nodes + (q - 1u));
        rec = make_uint4(q - 1u, nb + min(w.x, w.z), nb + max(w.x, w.z), nb + w.y);
    }
    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(uq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)uq, 0u));
    // first unique visit of its node: the unique lane below holds another node (the visits of a node are neighbours)
    const unsigned long long lower = uq & ((1ull << lane) - 1ull);
    const uint32_t prev_w = __shfl(rec.w, lower ? 63 - __builtin_clzll(lower) : lane);
    const bool first = mine && (!lower || prev_w != rec.w);
    const unsigned long long fm = __ballot(first);
    if (mine) row_file<KEYS, FIRST>(o, base + r, rec.x, rec.y, rec.z, rec.w, sp, hc);
    // the node's rows end at the next first lane; its pair filter = OR of the bits of its unique lanes (every first lane walks its span: a node's
    // unique visits, a handful; all lanes reach the shuffles)
    const uint32_t pbit = mine ? nr_pair_bit(rec.y, rec.z) : 0u;
    const unsigned long long nxt = fm & ~((2ull << lane) - 1ull);
    const unsigned long long span = uq & ~((1ull << lane) - 1ull) & (nxt ? (1ull << __builtin_ctzll(nxt)) - 1ull : ~0ull);
    uint32_t filt = 0u;
    unsigned long long sp_ = first ? span : 0ull;
    while (__any(sp_ != 0ull)) {
        const int l = sp_ ? __builtin_ctzll(sp_) : 0;
        const uint32_t ob = __shfl(pbit, l);
        if (sp_) { filt |= ob; sp_ &= sp_ - 1ull; }
    }
    if (first) trio_head_store(node_rec, rec.w, base + r, (uint32_t)__popcll(span), filt, err);
}
// EIGHT groups per batch, lane = (group, record): the records of a group that did not overflow (<= VIS_REC unique visits) are read as
// one coalesced kilobyte per batch; the row of record r of group g is the scan of the groups' counts + r.  A wave takes U batches at
// once, level by level -- counts, records, then the gathers every record depends on (three node lengths, its species' walk offsets, the node
// record its head goes into) -- so that U x the loads are in flight per wave.  Every store is dense in row order except the heads.
template <bool KEYS, bool FIRST, int U>
__global__ void __launch_bounds__(256) trio_rows_kernel(uint32_t NG, const unsigned long long *__restrict__ vis_uq, const uint32_t *__restrict__ gprefix,
                                                        const uint4 *__restrict__ vis_rec, const uint32_t *__restrict__ vis_nbase, const uint32_t *__restrict__ vis_sp,
                                                        const uint32_t *__restrict__ vis_pos, const uint32_t *__restrict__ path_nodes,
                                                        uint4 *__restrict__ node_rec, RowOut o, uint32_t *__restrict__ err, uint32_t xcd_chunks, uint32_t iters) {
    static_assert(VIS_REC == 8, "eight lanes per group");
    const int lane = threadIdx.x & 63;
    // FIRST builds: a workgroup takes `iters` consecutive chunks and counts the rows per haplotype in LDS (HapCount)
    __shared__ uint32_t s_hapcnt[FIRST ? HAPCNT_WIN : 1];
    HapCount hc{nullptr, 0u, o.hap_cnt};
    if (FIRST) {
        for (uint32_t i = threadIdx.x; i < HAPCNT_WIN; i += blockDim.x) s_hapcnt[i] = 0u;
        const uint32_t gfirst = blockIdx.x * iters * 32u * (uint32_t)U;
        hc.lds = s_hapcnt; hc.base = (uint32_t)o.hap_off[vis_sp[gfirst < NG ? gfirst : NG - 1u]];
        __syncthreads();
    }
    for (uint32_t it = 0; it < iters; ++it) {
    uint32_t blk = blockIdx.x * iters + it;          // xcd_chunks != 0 (rebuilds, iters == 1): every XCD files one contiguous eighth of the groups (see trio_visit_kernel)
    if (xcd_chunks) { blk = (blockIdx.x & 7u) * ((xcd_chunks + 7u) / 8u) + (blockIdx.x >> 3); if (blk >= xcd_chunks) break; }
    const uint32_t r = (uint32_t)lane & 7u;
    uint32_t g[U], cnt[U], row[U], sp[U];
    // ---- level 1: the groups' counts, first rows and species
#pragma unroll
    for (int u = 0; u < U; ++u) {
        g[u] = ((blk * 4u + (threadIdx.x >> 6)) * (uint32_t)U + (uint32_t)u) * 8u + ((uint32_t)lane >> 3);
        cnt[u] = 0; row[u] = 0; sp[u] = 0;
        if (g[u] < NG) { cnt[u] = (uint32_t)__popcll(vis_uq[g[u]]); row[u] = gprefix[g[u]] + r; sp[u] = vis_sp[g[u]]; }
    }
    // ---- level 2: the records
    uint4 rec[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        on[u] = cnt[u] <= (uint32_t)VIS_REC && r < cnt[u];                     // an overflowing group is taken whole, below
        rec[u] = make_uint4(0u, 0u, 0u, 0xFFFFFFFFu);
        if (on[u]) rec[u] = vis_rec[(uint64_t)g[u] * VIS_REC + r];
    }
    // the owner of a window = the haplotype whose walk holds its start.  The eight groups of a batch nearly always belong to ONE species: the
    // walk offsets of that species' haplotypes (up to 64) are loaded once, lane j holds offset j, and every lane counts the offsets at or
    // below its position by reading them lane after lane -- ALU work beside the record loads instead of a binary search of four dependent
    // loads behind them (the kernel waits for memory: every level of the chain shows).  Lanes of another species take the search.
    uint32_t sp0[U], h00[U], hs0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long live = __ballot(g[u] < NG);
        sp0[u] = (uint32_t)__builtin_amdgcn_readlane((int)sp[u], live ? __builtin_ctzll(live) : 0);
        h00[u] = (uint32_t)o.hap_off[sp0[u]]; hs0[u] = (uint32_t)o.hap_off[sp0[u] + 1] - h00[u];
    }
    // ---- level 3: what every record points at
    uint32_t len3[U], woff[U];
    uint4 nrv[U];
    bool first[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // first record of its node: the record below belongs to another node (or to another group)
        const uint32_t below = wave_shr1(rec[u].w, 0xFFFFFFFFu);
        first[u] = on[u] && (r == 0u || below != rec[u].w);
        len3[u] = 0; nrv[u] = make_uint4(0u, 0u, 0u, 0u);
        woff[u] = ((uint32_t)lane < hs0[u] && hs0[u] <= 64u) ? (uint32_t)o.path_off[h00[u] + (uint32_t)lane] : 0xFFFFFFFFu;   // P < 2^32
        if (on[u]) len3[u] = o.node_len[rec[u].y] + o.node_len[rec[u].w] + o.node_len[rec[u].z];
        if (first[u]) nrv[u] = node_rec[rec[u].w];
    }
    // ---- the rows, the heads
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long fm = __ballot(first[u]), om = __ballot(on[u]);
        uint32_t hl = 0;                                                     // owner within the species: offsets at or below the position, minus one
        if (hs0[u] <= 64u) {
            for (uint32_t j = 1; j < hs0[u]; ++j) hl += (uint32_t)__builtin_amdgcn_readlane((int)woff[u], (int)j) <= rec[u].x ? 1u : 0u;
        }
        if (on[u]) {
            uint32_t hb = h00[u];
            if (hs0[u] > 64u || sp[u] != sp0[u]) {                           // a species of more than 64 haplotypes, or not the batch's first species
                hb = (uint32_t)o.hap_off[sp[u]];
                hl = hap_of_position(o.path_off, hb, (uint32_t)o.hap_off[sp[u] + 1], rec[u].x) - hb;
            }
            o.ent[row[u]] = make_uint2(rec[u].y, rec[u].z);
            o.put_len_hap(row[u], len3[u], hl);
            if (KEYS) o.q[row[u]] = rec[u].x;
            if (FIRST) hc.add(hb + hl);
        }
        // the pair filter of a node = OR of its rows' bits: the rows of a node are neighbouring lanes (at most eight)
        const uint32_t pbit = on[u] ? nr_pair_bit(rec[u].y, rec[u].z) : 0u;
        uint32_t filt = pbit;
#pragma unroll
        for (int d = 1; d < 8; ++d) {
            const uint32_t ob = __shfl(pbit, (lane + d) & 63), ow = __shfl(rec[u].w, (lane + d) & 63);
            if (((lane & 7) + d) < 8 && ow == rec[u].w) filt |= ob;
        }
        if (first[u]) {
            // rows of the node: up to the next first record, or to the end of the group's records
            const unsigned long long grp = 0xFFull << (lane & ~7), stop = (fm | ~om) & grp & ~((2ull << lane) - 1ull);
            const int end = stop ? __builtin_ctzll(stop) : (lane & ~7) + 8;
            const uint32_t rows = (uint32_t)(end - lane);
            if (rows >= NODE_REC_MAX_ROWS) atomicAdd(err, 1u);
            // the head of a node is a function of the graphs alone: every rebuild computes it again, and STORES it only where the record does
            // not hold it yet (the first build of a db) -- a 16-byte store into a line of eight records dirties a 64-byte sector, and the heads
            // of 1e4 strains were 4.5 of the 7.4 GB this kernel wrote per build (`r05_pmc_trio_probe`)
            uint4 nr = nrv[u];
            const uint32_t y_new = nr_head(nr.y, rows, filt);
            if (nr.y != y_new || nr.w != row[u]) { nr.y = y_new; nr.w = row[u]; node_rec[rec[u].w] = nr; }
        }
    }
    // the groups of this wave with more unique visits than records, one after the other
#pragma unroll
    for (int u = 0; u < U; ++u) {
        unsigned long long ov = __ballot(r == 0u && cnt[u] > (uint32_t)VIS_REC);
        while (ov) {
            const int l = __builtin_ctzll(ov);
            ov &= ov - 1ull;
            trio_rows_group<KEYS, FIRST>(g[u] - ((uint32_t)lane >> 3) + ((uint32_t)l >> 3), lane, vis_uq, gprefix, vis_pos, vis_nbase, vis_sp, path_nodes, node_rec, o, err, hc);
        }
    }
    }   // iters
    if (FIRST) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < HAPCNT_WIN; i += blockDim.x) { const uint32_t c = s_hapcnt[i]; if (c) atomicAdd(&o.hap_cnt[hc.base + i], c); }
    }
}

// ---- REBUILDS of a db whose group offsets are known: uniqueness and filing in ONE pass over the visit table -------------------------------
// The first row of every group of 64 visits (gprefix) is a function of the graphs alone, like the group boundaries of the visit table themselves:
// the db's first build learns it (trio_visit_kernel<ROWS> -> prefix of the groups' counts -> trio_rows_kernel) and keeps it with the table.  Every
// later build -- the per-run rebuild of a resident step, profile.rs:2936 -- decides the uniqueness of every window again and files every row again,
// in the kernel that took the decision: no records through memory, no scan, no second kernel.  The offsets are VERIFIED on the way: a group whose
// count of unique visits is not what its neighbours' offsets say raises the error word (it comes back with the step's results).
// A group holds about five unique visits, so filing from the deciding lanes would run everything behind the decision at a twelfth of the lanes (first
// version: 14.6 ms at 1e4 strains against 5.6 + 7.0 for the two kernels -- the kernel is bound by VALU issue, 64-lane instructions per group).  Instead
// every wave QUEUES its unique windows in LDS -- consecutive groups of one species have consecutive rows -- and files the queue on dense lanes, lane =
// row, whenever the next group would not fit: coalesced stores of 64 consecutive rows, one pass over the species' walk offsets per ~12 groups.
struct FileQueue {
    uint4 rec[64];       // {window start, smaller end, larger end, middle} (global node indices), in visit order = row order
};
template <bool KEYS>
__device__ __forceinline__ void trio_file_flush(const FileQueue &qu, uint32_t cnt, uint32_t row0, uint32_t sp, int lane, uint4 *__restrict__ node_rec, const RowOut &o,
                                                uint32_t *__restrict__ err) {
    const bool on = (uint32_t)lane < cnt;
    const uint4 rec = on ? qu.rec[lane] : make_uint4(0u, 0u, 0u, 0xFFFFFFFFu);
    const uint32_t h0 = (uint32_t)o.hap_off[sp], hs = (uint32_t)o.hap_off[sp + 1] - h0;      // wave-uniform
    // the owner of a window = the haplotype whose walk holds its start: the species' walk offsets (up to 64) sit one per lane and every lane counts those
    // at or below its start (trio_rows_kernel)
    const uint32_t woff = ((uint32_t)lane < hs && hs <= 64u) ? (uint32_t)o.path_off[h0 + (uint32_t)lane] : 0xFFFFFFFFu;   // P < 2^32
    // first row of its node: the row below belongs to another node (a node's unique visits are neighbours, and groups -- hence queues -- hold whole nodes)
    const uint32_t below = wave_shr1(rec.w, 0xFFFFFFFFu);
    const bool first = on && (lane == 0 || below != rec.w);
    uint32_t len3 = 0;
    uint4 nrv = make_uint4(0u, 0u, 0u, 0u);
    if (on) len3 = o.node_len[rec.y] + o.node_len[rec.w] + o.node_len[rec.z];
    if (first) nrv = node_rec[rec.w];
    const unsigned long long fm = __ballot(first), om = __ballot(on);
    uint32_t hl = 0;
    if (hs <= 64u) { for (uint32_t j = 1; j < hs; ++j) hl += (uint32_t)__builtin_amdgcn_readlane((int)woff, (int)j) <= rec.x ? 1u : 0u; }
    const uint32_t row = row0 + (uint32_t)lane;
    if (on) {
        if (hs > 64u) hl = hap_of_position(o.path_off, h0, h0 + hs, rec.x) - h0;
        o.ent[row] = make_uint2(rec.y, rec.z);
        o.put_len_hap(row, len3, hl);
        if (KEYS) o.q[row] = rec.x;
    }
    // the node's rows end at the next first lane; its pair filter = OR of its rows' bits (every first lane walks its span: a handful of lanes)
    const uint32_t pbit = on ? nr_pair_bit(rec.y, rec.z) : 0u;
    const unsigned long long nxt = (fm | ~om) & ~((2ull << lane) - 1ull);
    const int end = nxt ? __builtin_ctzll(nxt) : 64;
    const unsigned long long span = first ? ((end == 64 ? ~0ull : (1ull << end) - 1ull) & ~((1ull << lane) - 1ull)) : 0ull;
    uint32_t filt = 0u;
    unsigned long long sp_ = span;
    while (__any(sp_ != 0ull)) {
        const int l = sp_ ? __builtin_ctzll(sp_) : 0;
        const uint32_t ob = __shfl(pbit, l);
        if (sp_) { filt |= ob; sp_ &= sp_ - 1ull; }
    }
    if (first) {
        const uint32_t rows = (uint32_t)(end - lane);
        if (rows >= NODE_REC_MAX_ROWS) atomicAdd(err, 1u);
        uint4 nr = nrv;
        const uint32_t y_new = nr_head(nr.y, rows, filt);
        if (nr.y != y_new || nr.w != row) { nr.y = y_new; nr.w = row; node_rec[rec.w] = nr; }   // (stored only where it is not there yet: trio_rows_kernel)
    }
}
template <int U, bool KEYS>
__global__ void __launch_bounds__(256) trio_file_kernel(uint32_t NG, uint32_t rounds, const uint32_t *__restrict__ vis_pos, const uint64_t *__restrict__ vis_head,
                                                        const uint32_t *__restrict__ vis_nbase, const uint32_t *__restrict__ vis_sp, const uint32_t *__restrict__ gprefix,
                                                        const uint32_t *__restrict__ path_nodes, uint4 *__restrict__ node_rec, RowOut o, uint32_t *__restrict__ err,
                                                        uint32_t xcd_chunks) {
    __shared__ FileQueue queues[4];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    FileQueue &qu = queues[wave];
    uint32_t blk = blockIdx.x;
    if (xcd_chunks) { blk = (blockIdx.x & 7u) * ((xcd_chunks + 7u) / 8u) + (blockIdx.x >> 3); if (blk >= xcd_chunks) blk = 0xFFFFFFu; }
    uint32_t g0 = blk == 0xFFFFFFu ? NG : (blk * 4u + wave) * ((uint32_t)U * rounds);      // this wave's U x rounds consecutive groups
    uint32_t q_cnt = 0, q_row0 = 0, q_sp = 0;                                              // the queue: entries, row of the first, their species (wave-uniform)
    for (uint32_t r = 0; r < rounds && g0 < NG; ++r, g0 += U) {
        uint32_t q[U], nb[U], sp[U], base[U], want[U];
        uint64_t heads[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = g0 + (uint32_t)u < NG ? g0 + (uint32_t)u : g0;   // wave-uniform
            q[u] = vis_pos[(uint64_t)g * 64 + lane];
            heads[u] = vis_head[g]; nb[u] = vis_nbase[g]; sp[u] = vis_sp[g];
            base[u] = gprefix[g]; want[u] = gprefix[g + 1] - base[u];
        }
        __builtin_amdgcn_sched_barrier(0);       // all U table loads leave before the first of them is waited for
        U32x3 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            valid[u] = g0 + (uint32_t)u < NG && q[u] != VIS_PAD;
            w[u] = *reinterpret_cast<const U32x3 *>(path_nodes + (valid[u] ? q[u] - 1u : 0u));   // an interior position: p - 1 and p + 1 exist
        }
        __builtin_amdgcn_sched_barrier(0);       // ... and all U gathers before the first decision
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // the decision: trio_visit_kernel's
            const uint32_t lo = min(w[u].x, w[u].z), hi = max(w[u].x, w[u].z);
            const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid[u]);
            const unsigned long long hd = heads[u] & vmask;
            const unsigned long long inb = vmask & ~hd;                      // lanes with a lane of their own stretch below them
            const uint32_t slo = wave_shr1z(lo), shi = wave_shr1z(hi);       // the pair of the lane below (DPP moves)
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(slo == lo && shi == hi) & inb;
            const unsigned long long bad = __builtin_amdgcn_ballot_w64(slo > lo || (slo == lo && shi > hi)) & inb;
            const unsigned long long dup = eq | (eq >> 1);                   // both partners are not unique
            const unsigned long long uq = vmask & ~dup;
            const uint32_t n_g = (uint32_t)__popcll(uq);                     // wave-uniform
            if (g0 + (uint32_t)u < NG && (bad || n_g != want[u]) && lane == 0) atomicAdd(err, 1u);   // table out of order / offsets that are not this table's
            if (n_g == 0u) continue;
            // the queue holds consecutive rows of one species: file it first where this group does not fit behind them
            if (q_cnt && (q_cnt + n_g > 64u || sp[u] != q_sp || base[u] != q_row0 + q_cnt)) {
                trio_file_flush<KEYS>(qu, q_cnt, q_row0, q_sp, lane, node_rec, o, err);
                q_cnt = 0;
            }
            if (q_cnt == 0u) { q_row0 = base[u]; q_sp = sp[u]; }
            if ((uq >> lane) & 1ull) {
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(uq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)uq, 0u));   // unique visits in the lanes below
                qu.rec[q_cnt + rk] = make_uint4(q[u] - 1u, nb[u] + lo, nb[u] + hi, nb[u] + w[u].y);
            }
            q_cnt += n_g;
        }
    }
    if (q_cnt) trio_file_flush<KEYS>(qu, q_cnt, q_row0, q_sp, lane, node_rec, o, err);
}

// ---- the export order: rows listed in (species, hap, position) order = ascending window start (the walks are one CSR over all haplotypes) ----
__global__ void __launch_bounds__(256) trio_iota_kernel(uint32_t n, uint32_t *__restrict__ v, const uint32_t *__restrict__ q, unsigned long long *__restrict__ key) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { v[i] = i; key[i] = q[i]; }
}
// export copies of the table in that order: canonical key (species-local), owner haplotype, length
__global__ void __launch_bounds__(256) trio_export_kernel(uint32_t n, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ q, const uint32_t *__restrict__ path_nodes,
                                                          const uint16_t *__restrict__ hap, const trio_len_t *__restrict__ len, uint32_t *__restrict__ abc_out,
                                                          uint32_t *__restrict__ hap_out, uint32_t *__restrict__ len_out) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const uint32_t row = perm[e], p = q[row];
    uint32_t a = path_nodes[p], b = path_nodes[p + 1], c = path_nodes[p + 2];
    if (a > c) { const uint32_t t = a; a = c; c = t; }                     // profile.rs:672-678
    if (abc_out) { abc_out[3ull * e] = a; abc_out[3ull * e + 1] = b; abc_out[3ull * e + 2] = c; }
#if TRIO_LH_PACK
    if (hap_out) hap_out[e] = len[row].y;
    if (len_out) len_out[e] = len[row].x;
#else
    if (hap_out) hap_out[e] = hap[row];
    if (len_out) len_out[e] = len[row];
#endif
}
__global__ void __launch_bounds__(256) gather_u64_kernel(uint32_t n, const uint32_t *__restrict__ perm, const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) dst[e] = src[perm[e]];
}

// The visit table (end of db upload).  Counting sort of the interior positions by their node: count -> which species stay
// with the node-block kernel -> greedy packing of every 256-node chunk into groups of 64 visits (one thread per chunk; a
// chunk starts on a group border, so the chunks pack independently) -> scan of the chunk sizes -> place -> fill -> sort.
int trio_visits_build(Ctx *ctx, Db *db) {
    db->trio_visit_ok = false;
    db->n_vgroups = 0;
    db->h_trio_slow.assign(db->S, 1);          // until shown otherwise every species is the node-block kernel's
    const bool force_block = ctx->cfg.trio_path == "block";   // every species through the node-block kernel (tests, measurements)
    PTX_HIP(ctx, db->d_trio_slow.alloc(db->S ? db->S : 1));
    PTX_HIP(ctx, hipMemsetAsync(db->d_trio_slow.p, 0, (db->S ? db->S : 1) * sizeof(uint32_t), ctx->stream));
    PTX_HIP(ctx, db->d_node_visited.alloc(db->V / 32 + 2));
    PTX_HIP(ctx, hipMemsetAsync(db->d_node_visited.p, 0, (db->V / 32 + 2) * sizeof(uint32_t), ctx->stream));
    auto all_slow = [&]() -> int {
        std::vector<uint32_t> ones(db->S ? db->S : 1, 1u);
        PTX_TRY(upload(ctx, db->d_trio_slow, ones.data(), ones.size()));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    if (db->P == 0 || db->P >= 0xFFFFFFFFull || db->V == 0 || db->S == 0) return all_slow();
    DevBuf<uint32_t> cnt, vslot, chunk_groups, chunk_gbase, scan_tmp, tot;
    PTX_HIP(ctx, cnt.alloc(db->V + 1));
    PTX_TRY(zero_fill(ctx, cnt.p, (db->V + 1) * sizeof(uint32_t)));
#define TRIO_GRAPH db->d_tiles.p, db->d_path_off.p, db->d_path_nodes.p, db->d_hap_species.p, db->d_node_base.p
    const dim3 tgrid((uint32_t)db->n_tiles);
    hipLaunchKernelGGL(visit_count_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, cnt.p);
    hipLaunchKernelGGL(visit_flags_kernel, dim3((uint32_t)((db->V + 256) / 256)), dim3(256), 0, ctx->stream, db->V, db->S, db->d_node_base.p, cnt.p,
                       db->d_node_visited.p, db->d_trio_slow.p);
    std::vector<uint32_t> slow(db->S);
    PTX_TRY(download(ctx, slow.data(), db->d_trio_slow.p, db->S));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint4> chunks;
    for (uint32_t s = 0; s < db->S; ++s) {
        if (force_block) slow[s] = 1u;
        if (slow[s]) continue;
        for (uint64_t v = db->h_node_off[s]; v < db->h_node_off[s + 1]; v += (1u << VIS_CHUNK_SHIFT))
            chunks.push_back(make_uint4((uint32_t)v, (uint32_t)std::min<uint64_t>(v + (1u << VIS_CHUNK_SHIFT), db->h_node_off[s + 1]), (uint32_t)db->h_node_off[s], s));
    }
    const uint32_t NC = (uint32_t)chunks.size();
    if (NC == 0) return all_slow();
    DevBuf<uint4> d_chunks;
    PTX_TRY(upload(ctx, d_chunks, chunks.data(), chunks.size()));
    PTX_HIP(ctx, chunk_groups.alloc(NC + 1)); PTX_HIP(ctx, chunk_gbase.alloc(NC + 1));
    PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(NC + 1))); PTX_HIP(ctx, tot.alloc(1));
    PTX_HIP(ctx, hipMemsetAsync(chunk_groups.p + NC, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(visit_pack_kernel<false>, dim3((NC + VP_CHUNKS - 1) / VP_CHUNKS), dim3(64), 0, ctx->stream, NC, d_chunks.p, cnt.p, chunk_groups.p, (const uint32_t *)nullptr,
                       (uint32_t *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    PTX_TRY(exclusive_scan_u32(ctx, chunk_groups.p, chunk_gbase.p, (uint64_t)NC + 1, scan_tmp.p, tot.p));
    uint32_t NG = 0;
    PTX_TRY(download(ctx, &NG, tot.p, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((uint64_t)NG * 64 >= 0xFFFFFFFFull) return all_slow();   // slots are 32-bit
    if (NG) {
        PTX_HIP(ctx, db->d_vis_pos.alloc((uint64_t)NG * 64)); PTX_HIP(ctx, db->d_vis_head.alloc(NG)); PTX_HIP(ctx, db->d_vis_nbase.alloc(NG)); PTX_HIP(ctx, db->d_vis_sp.alloc(NG));
        PTX_HIP(ctx, vslot.alloc(db->V));
        PTX_TRY(byte_fill(ctx, db->d_vis_pos.p, 0xFF, (uint64_t)NG * 64 * sizeof(uint32_t)));
        PTX_TRY(byte_fill(ctx, db->d_vis_head.p, 0, (uint64_t)NG * sizeof(uint64_t)));
        PTX_TRY(byte_fill(ctx, db->d_vis_nbase.p, 0, (uint64_t)NG * sizeof(uint32_t)));
        PTX_TRY(byte_fill(ctx, db->d_vis_sp.p, 0, (uint64_t)NG * sizeof(uint32_t)));
        PTX_TRY(upload(ctx, db->d_trio_slow, slow.data(), slow.size()));
        hipLaunchKernelGGL(visit_pack_kernel<true>, dim3((NC + VP_CHUNKS - 1) / VP_CHUNKS), dim3(64), 0, ctx->stream, NC, d_chunks.p, cnt.p, (uint32_t *)nullptr,
                           (const uint32_t *)chunk_gbase.p, vslot.p, reinterpret_cast<unsigned long long *>(db->d_vis_head.p), db->d_vis_nbase.p, db->d_vis_sp.p);
        hipLaunchKernelGGL(visit_fill_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, db->d_trio_slow.p, vslot.p, cnt.p, db->d_vis_pos.p);
        hipLaunchKernelGGL(visit_sort_kernel, dim3((NG + 3) / 4), dim3(256), 0, ctx->stream, NG, db->d_vis_pos.p, db->d_vis_head.p, db->d_path_nodes.p);
    } else PTX_TRY(upload(ctx, db->d_trio_slow, slow.data(), slow.size()));
#undef TRIO_GRAPH
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the temporaries (and `slow`, `chunks`) go out of scope
    for (uint32_t s = 0; s < db->S; ++s) db->h_trio_slow[s] = slow[s] ? 1 : 0;
    db->n_vgroups = NG;
    db->trio_visit_ok = true;
    return 0;
}

int trio_runs_build(Ctx *ctx, Db *db) {
    db->trio_block_ok = false;
    db->n_blocks = 0; db->n_runs = 0;
    if (db->P == 0 || db->P >= 0xFFFFFFFFull) return 0;
    std::vector<uint32_t> blk_base(db->S + 1, 0);
    for (uint32_t s = 0; s < db->S; ++s) {
        const uint64_t Vs = db->h_trio_slow[s] ? db->h_node_off[s + 1] - db->h_node_off[s] : 0;   // blocks only where the visit table leaves a species to this path
        if (Vs >= (1ull << 27)) return 0;             // the packed LDS key holds 27-bit local ids: such a db keeps the bucket path
        const uint64_t nb = (uint64_t)blk_base[s] + ((Vs + TRIO_BLK - 1) >> TRIO_BLK_SHIFT);
        if (nb >= 0x7FFFFFFFull) return 0;
        blk_base[s + 1] = (uint32_t)nb;
    }
    const uint32_t NB = blk_base[db->S];
    db->trio_block_ok = true;
    if (NB == 0) return 0;                            // every species goes through the visit table
    db->trio_block_ok = false;
    std::vector<uint32_t> blk_species(NB);
    for (uint32_t s = 0; s < db->S; ++s) std::fill(blk_species.begin() + blk_base[s], blk_species.begin() + blk_base[s + 1], s);
    PTX_TRY(upload(ctx, db->d_blk_base, blk_base.data(), db->S + 1));
    PTX_TRY(upload(ctx, db->d_blk_species, blk_species.data(), NB));
    DevBuf<uint32_t> cnt, scan_tmp, tot;
    PTX_HIP(ctx, cnt.alloc(2ull * (NB + 1)));
    PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(NB + 1)));
    PTX_HIP(ctx, tot.alloc(1));
    PTX_HIP(ctx, db->d_blk_run_off.alloc(NB + 1));
    PTX_HIP(ctx, hipMemsetAsync(cnt.p, 0, 2ull * (NB + 1) * sizeof(uint32_t), ctx->stream));
#define TRIO_GRAPH db->d_tiles.p, db->d_path_off.p, db->d_path_nodes.p, db->d_hap_species.p, db->d_node_base.p
    const dim3 tgrid((uint32_t)db->n_tiles);
    hipLaunchKernelGGL(run_count_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, db->d_trio_slow.p, db->d_blk_base.p, cnt.p);
    PTX_TRY(exclusive_scan_u32(ctx, cnt.p, db->d_blk_run_off.p, (uint64_t)NB + 1, scan_tmp.p, tot.p));
    uint32_t h_tot = 0;
    PTX_TRY(download(ctx, &h_tot, tot.p, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PTX_HIP(ctx, db->d_runs.alloc(h_tot ? h_tot : 1));
    hipLaunchKernelGGL(run_fill_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, db->d_trio_slow.p, db->d_blk_base.p, db->d_blk_run_off.p, cnt.p + (NB + 1), db->d_runs.p);
#undef TRIO_GRAPH
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the temporaries go out of scope
    {   // one record per block for the build kernel: {first run, end run, global first node, species-local first node}
        std::vector<uint32_t> run_off(NB + 1);
        PTX_TRY(download(ctx, run_off.data(), db->d_blk_run_off.p, (size_t)NB + 1));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint4> rec(NB + 1);
        for (uint32_t s = 0; s < db->S; ++s)
            for (uint32_t gb = blk_base[s]; gb < blk_base[s + 1]; ++gb) {
                const uint32_t n0 = (gb - blk_base[s]) << TRIO_BLK_SHIFT;
                const uint32_t nn = (uint32_t)std::min<uint64_t>(TRIO_BLK, db->h_node_off[s + 1] - db->h_node_off[s] - n0);
                rec[gb] = make_uint4(run_off[gb], run_off[gb + 1], (uint32_t)db->h_node_off[s] + n0, (gb - blk_base[s]) | (nn << 24));   // < 2^21 blocks per species (2^27 nodes)
            }
        rec[NB] = make_uint4(h_tot, h_tot, (uint32_t)db->V, 0u);
        PTX_TRY(upload(ctx, db->d_blk_rec, rec.data(), rec.size()));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    db->n_blocks = NB;
    db->n_runs = h_tot;
    db->trio_block_ok = true;
    return 0;
}

// The index of a db: uniqueness (visit table; node blocks or global buckets for what it does not cover), then the rows.
//   FAST route (species of the visit table): trio_visit_kernel<ROWS> -> prefix of the groups' counts -> trio_rows_kernel.
//   PATH route (the other species; the whole db under trio_path=bucket / trio_rows=path): flags + per-node counts -> scan of the counts
//   (heads) -> trio_lookup_kernel -> trio_canon_kernel.  In a mixed db the path route's rows follow the fast route's.
// with_keys: the window start of every row is kept as well (d_trio_q): what the exporters build the (species, hap, position) order from.
int trio_index_build(Ctx *ctx, Db *db, bool with_keys) {
    const uint64_t P = db->P, V = db->V;
    db->trio_keys_built = false;
    db->trio_perm_valid = false;
    const uint32_t H = (uint32_t)db->H, S = db->S;
    if (P >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "trio_index: %llu path steps exceed 32-bit positions", (unsigned long long)P);
    TrioScratch &ts = db->trio_scratch;
    // which uniqueness path: the visit table (default; species with a node of more than 64 visits: by node block), or through
    // global buckets for the whole db (a species of >= 2^27 nodes among those left to the node-block kernel, or forced)
    bool by_block = db->trio_block_ok && (db->trio_visit_ok || db->n_blocks);
    if (ctx->cfg.trio_path == "bucket") by_block = false;
    bool rows_by_visit = by_block && P && db->n_vgroups;
    if (ctx->cfg.trio_rows == "path") rows_by_visit = false;
    const bool path_route = P && (!rows_by_visit || db->n_blocks != 0);   // some (or all) rows are filed by the pass over the walks
    const bool mixed = rows_by_visit && path_route;
    // first build of a db -- or the first one that files the species in another order (the options trio_rows / trio_path changed between two
    // builds: tests): sizes, rows per haplotype and the chunk table of the per-haplotype statistics are (re)learnt
    const bool first_build = !db->trio_sizes_known || db->trio_layout_fast != rows_by_visit;
    // One arena for what the path route needs cleared: uniq bits (one per path position) | first_cnt [| cnt | cursor].  The visit-table /
    // node-block kernels STORE the count of every node that has a visit (the others read as zero through `d_node_visited`), the lookup
    // pass counts them back down to zero in place of a cursor array; cnt / cursor belong to the bucket path.  The fast route clears nothing.
    const size_t zbits = (P + 31) / 32 + 1;
    if (path_route) {
        const size_t zwords = zbits + (V + 1) + (by_block ? 0 : 2 * (V + 1));
        PTX_HIP(ctx, ts.zero_arena.alloc(zwords));
        ts.uniq_q.view(ts.zero_arena.p, zbits);
        ts.first_cnt.view(ts.zero_arena.p + zbits, V + 1);
        if (!by_block) { ts.cnt.view(ts.zero_arena.p + zbits + (V + 1), V + 1); ts.cursor.view(ts.zero_arena.p + zbits + 2 * (V + 1), V + 1); }
        if (!by_block) PTX_HIP(ctx, ts.bucket_off.alloc(V + 1));
        PTX_TRY(zero_fill(ctx, ts.zero_arena.p, (by_block ? zbits : zwords) * sizeof(uint32_t)));
        if (by_block) PTX_HIP(ctx, hipMemsetAsync(ts.first_cnt.p + V, 0, sizeof(uint32_t), ctx->stream));   // the closing entry of the count scan
        PTX_HIP(ctx, db->d_trio_first.alloc(V + 1));
    }
    PTX_HIP(ctx, ts.scan_tmp.alloc(16));
    PTX_HIP(ctx, ts.d_tot.alloc(4));
    PTX_HIP(ctx, hipMemsetAsync(ts.d_tot.p, 0, 4 * sizeof(uint32_t), ctx->stream));   // {-, rows of the fast route, error word of the build kernels, rows of the path route}
    // a rebuild of a db whose group offsets are known: uniqueness and filing of the visit table's species in one pass (trio_file_kernel)
    const bool fused = rows_by_visit && !first_build && ts.gprefix.p != nullptr && ts.gprefix_for == db->n_vgroups && !ctx->cfg.trio_two_pass;
    if (rows_by_visit && !fused) {
        PTX_HIP(ctx, ts.vis_uq.alloc(db->n_vgroups + 1)); PTX_HIP(ctx, ts.vis_rec.alloc((uint64_t)db->n_vgroups * VIS_REC));
        PTX_HIP(ctx, ts.gprefix.alloc(db->n_vgroups + 1));
        ts.gprefix_for = 0;
        PTX_HIP(ctx, hipMemsetAsync(ts.vis_uq.p + db->n_vgroups, 0, sizeof(uint64_t), ctx->stream));   // the closing entry of the count scan
    }
    if (first_build) {
        PTX_HIP(ctx, ts.hap_cnt.alloc(H + 1));
        PTX_HIP(ctx, hipMemsetAsync(ts.hap_cnt.p, 0, ((size_t)H + 1) * sizeof(uint32_t), ctx->stream));
    }
    PTX_HIP(ctx, db->d_hap_trio_off.alloc(H + 1));
#define TRIO_GRAPH db->d_tiles.p, db->d_path_off.p, db->d_path_nodes.p, db->d_hap_species.p, db->d_node_base.p
    const dim3 tgrid((uint32_t)db->n_tiles);
    // trio_xcd: bit 0 the visit kernel, bit 1 the rows kernel take their workgroups in XCD-contiguous chunks (measurements)
    const uint32_t trio_xcd = (uint32_t)ctx->cfg.trio_xcd;
    if (fused) {
        PTX_HIP(ctx, db->d_trio_ent.alloc(db->U_known + 1)); PTX_HIP(ctx, db->d_trio_len.alloc(db->U_known));   // (+ 1: the coverage pass loads entries in pairs)
#if !TRIO_LH_PACK
        PTX_HIP(ctx, db->d_trio_hap.alloc(db->U_known));
#endif
        if (with_keys) PTX_HIP(ctx, db->d_trio_q.alloc(db->U_known));
        const RowOut ro0{db->d_node_len.p, db->d_path_off.p, db->d_hap_off.p, db->d_trio_ent.p, db->d_trio_len.p, const_cast<uint16_t *>(TRIO_HAP_PTR(db)), db->d_trio_q.p, ts.hap_cnt.p};
        KTimer t(ctx, "trio_file_kernel");
        // U groups in flight x `rounds` rounds per wave (tf_u / tf_rounds pick another shape, for measurements).  EIGHT groups per wave, all in flight
        // at once, where the visit kernel of the first build takes 4 x 4: ms at 1e4 strains / at the fifty-strain share -- 8 x 1: 6.75 / 5.08, 4 x 2: 7.03 /
        // 5.33, 2 x 4: 7.23, 2 x 3: 7.35, 4 x 4: 8.65 / 5.90, 4 x 3: 8.64, 4 x 1: 8.72 / 7.14, 2 x 1: 11.2 / 9.1 (a wave that is filing its queue has no
        // loads in flight: short waves, many of them in turn -- but not so short that the queue is filed half empty)
        const uint32_t U = (uint32_t)ctx->cfg.tf_u, rounds = (uint32_t)std::max(1, ctx->cfg.tf_rounds);
#define TF_CHUNKS(UU) ((db->n_vgroups + 4u * UU * rounds - 1u) / (4u * UU * rounds))
#define TF_LAUNCH(UU, KK) hipLaunchKernelGGL((trio_file_kernel<UU, KK>), dim3((trio_xcd & 1u) ? ((TF_CHUNKS(UU) + 7u) / 8u) * 8u : TF_CHUNKS(UU)), dim3(256), 0, ctx->stream, db->n_vgroups, \
                                         rounds, db->d_vis_pos.p, db->d_vis_head.p, db->d_vis_nbase.p, db->d_vis_sp.p, (const uint32_t *)ts.gprefix.p, db->d_path_nodes.p, db->d_node_rec.p, ro0, \
                                         ts.d_tot.p + 2, (trio_xcd & 1u) ? TF_CHUNKS(UU) : 0u)
        if (with_keys) { if (U == 2) TF_LAUNCH(2, true); else if (U == 8) TF_LAUNCH(8, true); else TF_LAUNCH(4, true); }
        else { if (U == 2) TF_LAUNCH(2, false); else if (U == 8) TF_LAUNCH(8, false); else TF_LAUNCH(4, false); }
#undef TF_LAUNCH
#undef TF_CHUNKS
        // the rows of the fast route (the base of the path route's rows in a mixed db) = the closing entry of the offsets
        PTX_HIP(ctx, hipMemcpyAsync(ts.d_tot.p + 1, ts.gprefix.p + db->n_vgroups, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    }
    // ---- uniqueness
    if (P && by_block && db->n_vgroups && !fused) {
        KTimer t(ctx, "trio_visit_kernel");
        // every wave walks U x rounds consecutive groups of 64 visits (tv_u / tv_rounds pick another shape, for
        // measurements): consecutive groups visit consecutive nodes, whose walk entries share cache lines
        const uint32_t U = (uint32_t)ctx->cfg.tv_u, rounds = (uint32_t)std::max(1, ctx->cfg.tv_rounds);
        const uint32_t tv_ablate = ctx->cfg.tv_ablate;   // -DTV_ABLATE builds only
#define TV_CHUNKS(UU) ((db->n_vgroups + 4u * UU * rounds - 1u) / (4u * UU * rounds))
#define TV_LAUNCH(UU, RR) hipLaunchKernelGGL((trio_visit_kernel<UU, RR>), dim3((trio_xcd & 1u) ? ((TV_CHUNKS(UU) + 7u) / 8u) * 8u : TV_CHUNKS(UU)), dim3(256), 0, ctx->stream, db->n_vgroups, \
                                         rounds, db->d_vis_pos.p, db->d_vis_head.p, db->d_vis_nbase.p, db->d_path_nodes.p, ts.uniq_q.p, ts.first_cnt.p, ts.d_tot.p + 2, tv_ablate,  \
                                         reinterpret_cast<unsigned long long *>(ts.vis_uq.p), ts.vis_rec.p, (trio_xcd & 1u) ? TV_CHUNKS(UU) : 0u)
        if (rows_by_visit) { if (U == 2) TV_LAUNCH(2, true); else if (U == 8) TV_LAUNCH(8, true); else TV_LAUNCH(4, true); }
        else if (U == 2) TV_LAUNCH(2, false); else if (U == 8) TV_LAUNCH(8, false); else TV_LAUNCH(4, false);
#undef TV_LAUNCH
#undef TV_CHUNKS
    }
    if (P && by_block && db->n_blocks) {
        KTimer t(ctx, "trio_block_kernel");
        // LDS table slots per 64-node block (tb_slots = 512 | 256 | 128 picks another instantiation, for measurements): fewer
        // slots = more blocks resident per CU (the kernel is bound by the latency of each wave's dependent loads), more blocks
        // that need sub-passes
        const int slots = ctx->cfg.tb_slots;
#define TB_LAUNCH(N) hipLaunchKernelGGL(trio_block_kernel<N>, dim3(db->n_blocks), dim3(64), 0, ctx->stream, db->d_blk_rec.p, db->d_runs.p, \
                                        db->d_path_nodes.p, ts.uniq_q.p, ts.first_cnt.p, ts.d_tot.p + 2)
        if (slots == 512) TB_LAUNCH(512); else if (slots == 128) TB_LAUNCH(128); else TB_LAUNCH(256);
#undef TB_LAUNCH
    }
    if (P && !by_block) {
        PTX_HIP(ctx, ts.bucket.alloc(P));
        {
            KTimer t(ctx, "trio_count_kernel");
            hipLaunchKernelGGL(trio_count_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, ts.cnt.p);
        }
        PTX_TRY(exclusive_scan_u32(ctx, ts.cnt.p, ts.bucket_off.p, V + 1, ts.scan_tmp.p, ts.d_tot.p));
        {
            KTimer t(ctx, "trio_fill_kernel");
            hipLaunchKernelGGL(trio_fill_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, ts.bucket_off.p, ts.cursor.p, ts.bucket.p);
        }
        // number of windows: every hap with len >= 3 contributes len-2
        uint64_t n_win = 0;
        for (uint32_t h = 0; h < H; ++h) { uint64_t l = db->h_path_off[h + 1] - db->h_path_off[h]; if (l >= 3) n_win += l - 2; }
        if (n_win) {
            KTimer t(ctx, "trio_uniq_kernel");
            // mean bucket size decides: short buckets (few haplotypes per node) compare through shuffles, long ones hash
            bool hashed = n_win > 16 * V;   // measured: 7 windows per node -> shuffles 0.050 vs hash 0.058 ms; 34 per node -> 4.66 vs 1.69 ms
            if (ctx->cfg.uniq_hash >= 0) hashed = ctx->cfg.uniq_hash == 1;
            if (hashed)
                hipLaunchKernelGGL(trio_uniq_lds_kernel, dim3((uint32_t)((n_win + UNIQ_CH - 1) / UNIQ_CH)), dim3(256), 0, ctx->stream, n_win, (uint32_t)V,
                                   ts.bucket.p, ts.bucket_off.p, ts.uniq_q.p, ts.first_cnt.p);
            else
                hipLaunchKernelGGL(trio_uniq_kernel, dim3(grid_for(n_win, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, n_win, ts.bucket.p,
                                   ts.bucket_off.p, ts.uniq_q.p, ts.first_cnt.p);
        }
    }
    // ---- sizes: the first row of every group (fast route) and of every node (path route); on a db's first build the totals come back
    if (rows_by_visit && !fused) {
        if (ctx->cfg.flag_rank_chained)
            PTX_TRY(exclusive_scan_fn(ctx, GroupCountLoad{reinterpret_cast<const unsigned long long *>(ts.vis_uq.p)}, PrefixStore{ts.gprefix.p},
                                      (uint64_t)db->n_vgroups + 1, ts.d_tot.p + 1, "scan_chained_kernel<GroupCount>"));
        else {
            KTimer t(ctx, "group_tile_prefix_kernel");
            const uint64_t ng1 = (uint64_t)db->n_vgroups + 1;
            const uint32_t n_tiles = (uint32_t)((ng1 + FR_TILE - 1) / FR_TILE);
            PTX_HIP(ctx, ts.group_sums.alloc(n_tiles + 1));
            const unsigned long long *uq = reinterpret_cast<const unsigned long long *>(ts.vis_uq.p);
            hipLaunchKernelGGL(group_tile_sum_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, uq, ng1, ts.group_sums.p);
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, ts.group_sums.p, n_tiles, ts.d_tot.p + 1);
            hipLaunchKernelGGL(group_tile_prefix_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, uq, ng1, (const uint32_t *)ts.group_sums.p, ts.gprefix.p);
        }
    }
    if (path_route) {
        if (mixed)   // the species of the node-block kernel: heads behind the visit table's rows
            PTX_TRY(exclusive_scan_fn(ctx, SlowFirstLoad{ts.first_cnt.p, db->d_node_visited.p, db->d_trio_slow.p, db->d_emit_tile_sp.p, db->d_node_base.p, V},
                                      SlowFirstStore{db->d_trio_first.p, db->d_node_rec.p, V, ts.d_tot.p + 1, ts.d_tot.p + 2}, V, ts.d_tot.p + 3,
                                      "scan_chained_kernel<SlowFirst>"));
        else
            PTX_TRY(exclusive_scan_fn(ctx, TrioFirstLoad{ts.first_cnt.p, by_block ? db->d_node_visited.p : nullptr},
                                      TrioFirstStore{db->d_trio_first.p, db->d_node_rec.p, V, ts.d_tot.p + 2}, V + 1, ts.d_tot.p + 3, "scan_chained_kernel<TrioFirst>"));
    }
    uint32_t tot[4] = {0, 0, 0, 0};
    if (first_build) {   // U is a function of the graphs alone: a rebuild (pantax_hip_db_reset) reuses the size learnt by the first build
        PTX_TRY(download(ctx, tot, ts.d_tot.p, 4));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        db->U_known = (uint64_t)tot[1] + tot[3];
        if (db->U_known >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "trio_index: %llu unique trios exceed 32-bit rows", (unsigned long long)db->U_known);
    }
    const uint32_t Utot = (uint32_t)db->U_known;
    db->U = Utot;
    PTX_HIP(ctx, db->d_trio_ent.alloc((size_t)Utot + 1)); PTX_HIP(ctx, db->d_trio_len.alloc(Utot));   // (+ 1: the coverage pass loads entries in pairs)
#if !TRIO_LH_PACK
    PTX_HIP(ctx, db->d_trio_hap.alloc(Utot));
#endif
    if (with_keys) PTX_HIP(ctx, db->d_trio_q.alloc(Utot));
    if (path_route) PTX_HIP(ctx, ts.row_q.alloc(Utot));
    const RowOut ro{db->d_node_len.p, db->d_path_off.p, db->d_hap_off.p, db->d_trio_ent.p, db->d_trio_len.p, const_cast<uint16_t *>(TRIO_HAP_PTR(db)), db->d_trio_q.p, ts.hap_cnt.p};
    // ---- the rows
    if (rows_by_visit && !fused) {
        ts.gprefix_for = db->n_vgroups;                 // the offsets this launch files by stay with the table: later builds file in one pass (trio_file_kernel)
        KTimer t(ctx, "trio_rows_kernel");
        const uint32_t NG = db->n_vgroups;
        // a wave takes rows_u = 1, 2 or 4 batches of eight groups at once.  Two halve what the kernel waits for memory -- and the step got SLOWER
        // in round 4 (the current step's local sorts, which run beside it on the main stream, stretched from 1.1 to 3.8 ms).  Hence one.
        uint32_t RU = (uint32_t)ctx->cfg.rows_u;
        if (RU != 2 && RU != 4) RU = 1;
        const uint32_t rchunks = (NG + 32 * RU - 1) / (32 * RU);
        // a first build counts the rows per haplotype: 64 chunks per workgroup share one set of LDS counters (no XCD chunking there)
        const uint32_t iters = first_build ? 64u : 1u;
        const bool rxcd = (trio_xcd & 2u) && !first_build;
        const dim3 rgrid(first_build ? (rchunks + iters - 1) / iters : rxcd ? ((rchunks + 7u) / 8u) * 8u : rchunks);
#define ROWS_ARGS NG, reinterpret_cast<const unsigned long long *>(ts.vis_uq.p), ts.gprefix.p, ts.vis_rec.p, db->d_vis_nbase.p, db->d_vis_sp.p, db->d_vis_pos.p, \
                  db->d_path_nodes.p, db->d_node_rec.p, ro, ts.d_tot.p + 2, rxcd ? rchunks : 0u, iters
#define ROWS_LAUNCH(KK, FF, UU) hipLaunchKernelGGL((trio_rows_kernel<KK, FF, UU>), rgrid, dim3(256), 0, ctx->stream, ROWS_ARGS)
#define ROWS_PICK(KK, FF) { if (RU == 2) ROWS_LAUNCH(KK, FF, 2); else if (RU == 4) ROWS_LAUNCH(KK, FF, 4); else ROWS_LAUNCH(KK, FF, 1); }
        if (with_keys) { if (first_build) ROWS_PICK(true, true) else ROWS_PICK(true, false) }
        else { if (first_build) ROWS_PICK(false, true) else ROWS_PICK(false, false) }
#undef ROWS_PICK
#undef ROWS_LAUNCH
#undef ROWS_ARGS
    }
    if (path_route) {
        {
            KTimer t(ctx, "trio_lookup_kernel");
            hipLaunchKernelGGL(trio_lookup_kernel, tgrid, dim3(256), 0, ctx->stream, TRIO_GRAPH, (const uint32_t *)ts.uniq_q.p, (const uint32_t *)db->d_trio_first.p,
                               ts.first_cnt.p, db->d_trio_ent.p, ts.row_q.p, mixed ? (const uint32_t *)db->d_trio_slow.p : (const uint32_t *)nullptr);
        }
        KTimer t(ctx, "trio_canon_kernel");
        const dim3 cgrid((uint32_t)((V + 255) / 256));
#define CANON_LAUNCH(KK, FF) hipLaunchKernelGGL((trio_canon_kernel<KK, FF>), cgrid, dim3(256), 0, ctx->stream, V, S, (const uint32_t *)db->d_node_base.p,               \
                                                (const uint4 *)db->d_node_rec.p, (const uint32_t *)db->d_trio_first.p, by_block ? (const uint32_t *)db->d_node_visited.p \
                                                : (const uint32_t *)nullptr, mixed ? (const uint32_t *)db->d_trio_slow.p : (const uint32_t *)nullptr, ts.row_q.p, ro)
        if (with_keys) { if (first_build) CANON_LAUNCH(true, true); else CANON_LAUNCH(true, false); }
        else { if (first_build) CANON_LAUNCH(false, true); else CANON_LAUNCH(false, false); }
#undef CANON_LAUNCH
    }
#undef TRIO_GRAPH
    PTX_HIP(ctx, hipGetLastError());
    if (first_build) {
        // rows per haplotype -> hap_trio_off (the counts the first filter reads; the offsets of the export order), rows per species -> the
        // chunk table of the per-haplotype statistics; the error word of the build kernels is read HERE, behind all of them
        std::vector<uint32_t> hc(H + 1, 0);
        uint32_t err = 0;
        PTX_TRY(download(ctx, hc.data(), ts.hap_cnt.p, H));
        PTX_TRY(download(ctx, &err, ts.d_tot.p + 2, 1));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (err) return fail(ctx, PANTAX_HIP_E_LIMIT, "trio_index: %u problems in the build kernels: a node that heads 2^16 or more unique-trio rows, node blocks that could not be "
                                                      "resolved in LDS (trio_path=bucket), or visit groups out of order", err);
        db->h_hap_trio_off.assign(H + 1, 0);
        for (uint32_t h = 0; h < H; ++h) db->h_hap_trio_off[h + 1] = db->h_hap_trio_off[h] + hc[h];
        if (db->h_hap_trio_off[H] != db->U_known)
            return fail(ctx, PANTAX_HIP_E_STATE, "trio_index: %llu rows were filed but %llu counted", (unsigned long long)db->h_hap_trio_off[H], (unsigned long long)db->U_known);
        PTX_TRY(upload(ctx, db->d_hap_trio_off, db->h_hap_trio_off.data(), H + 1));
        // filing order of the species: those of the fast route first (in species order), then those of the path route
        db->h_sp_row_order.clear();
        for (int pass = 0; pass < 2; ++pass)
            for (uint32_t s = 0; s < S; ++s) {
                const bool fast = rows_by_visit && !db->h_trio_slow[s];
                if (fast == (pass == 0)) db->h_sp_row_order.push_back(s);
            }
        {
            uint64_t at = 0;
            std::vector<uint64_t> first(S, 0), cnt(S, 0);
            for (uint32_t s : db->h_sp_row_order) {
                cnt[s] = db->h_hap_trio_off[db->h_hap_off[s + 1]] - db->h_hap_trio_off[db->h_hap_off[s]];
                first[s] = at; at += cnt[s];
            }
            PTX_TRY(hap_stats_layout(ctx, db, first.data(), cnt.data()));
        }
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        db->trio_sizes_known = true;
        db->trio_layout_fast = rows_by_visit;
        // later builds file in one pass and need neither the records (16 B x 8 per group: 4.7 GB at 1e4 strains) nor the groups' ballots
        if (rows_by_visit && !ctx->cfg.trio_two_pass) { ts.vis_rec.release(); ts.vis_uq.release(); }
    }
    db->trio_built = true;
    db->trio_keys_built = with_keys;
    db->cov_done = false;
    return 0;
}

// the window start of every row is wanted (the exporters): rebuild with it unless it is there
int trio_keys_ensure(Ctx *ctx, Db *db) {
    if (db->trio_built && db->trio_keys_built) return 0;
    if (db->step_inflight) return fail(ctx, PANTAX_HIP_E_STATE, "trio tables: %d enqueued step(s) of this db have not been collected", db->step_inflight);
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream2));   // a step's rebuild on the side stream is over before the tables are replaced
    return trio_index_build(ctx, db, true);
}

// d_trio_perm[e] = the row of the e-th window in (species, hap, position) order: the rows sorted by their window start (the walks of all
// haplotypes are one CSR in that order).  Off the step's path: only pantax_hip_trio_get and the trio_bases of pantax_hip_node_coverage ask.
int trio_export_ensure(Ctx *ctx, Db *db) {
    PTX_TRY(trio_keys_ensure(ctx, db));
    if (db->trio_perm_valid) return 0;
    const uint64_t U = db->U;
    PTX_HIP(ctx, db->d_trio_perm.alloc(U ? U : 1));
    if (U) {
        DevBuf<uint64_t> ka, kb;
        DevBuf<uint32_t> vb, table, tmp;
        PTX_HIP(ctx, ka.alloc(U)); PTX_HIP(ctx, kb.alloc(U)); PTX_HIP(ctx, vb.alloc(U)); PTX_HIP(ctx, table.alloc(sort_table_elems(U))); PTX_HIP(ctx, tmp.alloc(16));
        hipLaunchKernelGGL(trio_iota_kernel, dim3((uint32_t)((U + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)U, db->d_trio_perm.p, (const uint32_t *)db->d_trio_q.p,
                           reinterpret_cast<unsigned long long *>(ka.p));
        SortBufs A, B;
        A.nw = B.nw = 1; A.k[0] = ka.p; B.k[0] = kb.p; A.v = db->d_trio_perm.p; B.v = vb.p;
        std::vector<SortPass> passes;
        add_passes(passes, 0, 0, bits_for(db->P ? db->P : 1));
        bool in_b = false;
        PTX_TRY(radix_sort(ctx, A, B, U, passes.data(), (int)passes.size(), table.p, tmp.p, &in_b, nullptr));
        if (in_b) PTX_HIP(ctx, hipMemcpyAsync(db->d_trio_perm.p, vb.p, U * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the sort's buffers go out of scope
    }
    db->trio_perm_valid = true;
    return 0;
}

// rows in export order: out[e] = src[row of e] (trio_bases of pantax_hip_node_coverage)
int trio_export_u64(Ctx *ctx, Db *db, const unsigned long long *d_src, unsigned long long *d_dst) {
    PTX_TRY(trio_export_ensure(ctx, db));
    if (db->U) hipLaunchKernelGGL(gather_u64_kernel, dim3((uint32_t)((db->U + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)db->U, (const uint32_t *)db->d_trio_perm.p, d_src, d_dst);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx

using namespace ptx;
extern "C" {

int pantax_hip_trio_index(pantax_hip_ctx *ctx, pantax_hip_db *db, uint64_t *n_unique_total_out) {
    if (!ctx || !db) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    if (!db->trio_built) {
        const bool rebuild = db->trio_sizes_known;
        PTX_TRY(trio_index_build(ctx, db));
        // a db's FIRST build reads the error word of its kernels itself; a rebuild (after pantax_hip_db_reset) leaves it for the pipelined step to
        // fetch with its results.  A stage caller consumes the index right away (trio_get, node_coverage): the word is read here, behind the build
        if (rebuild && db->trio_scratch.d_tot.p) {
            uint32_t err = 0;
            PTX_TRY(download(ctx, &err, db->trio_scratch.d_tot.p + 2, 1));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (err) {
                db->trio_built = false;
                return fail(ctx, PANTAX_HIP_E_LIMIT, "trio_index: %u problems in the rebuild's kernels: visit groups out of order / offsets that are not this table's, or a node "
                                                     "that heads 2^16 or more unique-trio rows", err);
            }
        }
    }
    if (n_unique_total_out) *n_unique_total_out = db->U;
    return 0;
}

int pantax_hip_trio_get(pantax_hip_ctx *ctx, const pantax_hip_db *cdb, uint32_t *abc_out, uint32_t *hap_out, int64_t *len_out,
                        uint64_t *hap_trio_off_out) {
    if (!ctx || !cdb) return PANTAX_HIP_E_INVALID;
    pantax_hip_db *db = const_cast<pantax_hip_db *>(cdb);
    if (!db->trio_built) return fail(ctx, PANTAX_HIP_E_STATE, "trio_get: call pantax_hip_trio_index first");
    PTX_ENTER(ctx);
    const uint64_t U = db->U;
    if ((abc_out || hap_out || len_out) && U) {
        // the table in (species, hap, position) order is an export: the rows are permuted on the device (the last build may have been a
        // step's, without the window starts: the same index is built again with them -- coverage results stay valid, a row keeps its number)
        const bool cov_done = db->cov_done;
        PTX_TRY(trio_export_ensure(ctx, db));
        db->cov_done = cov_done;
        DevBuf<uint32_t> d_abc, d_hap, d_len;
        if (abc_out) PTX_HIP(ctx, d_abc.alloc(3 * U));
        if (hap_out) PTX_HIP(ctx, d_hap.alloc(U));
        if (len_out) PTX_HIP(ctx, d_len.alloc(U));
        hipLaunchKernelGGL(trio_export_kernel, dim3((uint32_t)((U + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)U, (const uint32_t *)db->d_trio_perm.p,
                           (const uint32_t *)db->d_trio_q.p, (const uint32_t *)db->d_path_nodes.p, TRIO_HAP_PTR(db), (const trio_len_t *)db->d_trio_len.p,
                           abc_out ? d_abc.p : (uint32_t *)nullptr, hap_out ? d_hap.p : (uint32_t *)nullptr, len_out ? d_len.p : (uint32_t *)nullptr);
        PTX_HIP(ctx, hipGetLastError());
        std::vector<uint32_t> len32;
        if (abc_out) PTX_TRY(download(ctx, abc_out, d_abc.p, 3 * U));
        if (hap_out) PTX_TRY(download(ctx, hap_out, d_hap.p, U));
        if (len_out) { len32.resize(U); PTX_TRY(download(ctx, len32.data(), d_len.p, U)); }
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (len_out) for (uint64_t u = 0; u < U; ++u) len_out[u] = len32[u];
    }
    if (hap_trio_off_out) for (uint64_t h = 0; h <= db->H; ++h) hap_trio_off_out[h] = db->h_hap_trio_off[h];
    return 0;
}
}
