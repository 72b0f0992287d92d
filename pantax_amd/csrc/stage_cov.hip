// stage_cov.hip -- a8: the per-node coverage histogram (get_node_abundances, profile.rs:743-1026).
//
// Per read (semantics restated from the reference, line-cited below):
//   * local node = id - range_start (profile.rs:790 with start = range_start-1, :2886)
//   * one-node walk: target = pend-pstart; <0 => dropped (:821-827); bases += target (:828-829);
//     bitmap [pstart,pend) only if pstart<pend<=node_len (:832-841)
//   * otherwise: first node aligns node_len-pstart from pstart (:853-856; reference asserts
//     pstart<=node_len), interior nodes align fully (:860-862), the last aligns
//     max(target-seen,0) (:857-859); the bitmap is marked for every occurrence, clipped to the
//     node (:870-873); `seen` advances on every occurrence (:878) but bases are added once per
//     distinct node of the read (:879-882)
//   * every 3-window (a,b,c) is looked up in either orientation in the unique-trio table and adds
//     the read-local aligned lengths of its three nodes (:890-907)
// Outputs are integers and bit-exact: 64-bit atomic adds and 32-bit atomic ORs commute.
//
// Mapping: ONE THREAD PER WALK STEP.  A wave64 holds 64 consecutive steps (~8 neighbouring short
// reads); everything a step needs from its predecessors in the read comes from neighbouring lanes:
//   - `seen` (sum of aligned lengths before the last node): segmented wave scan over the lanes
//   - first-occurrence test of a node inside the read: shuffle compare against the earlier lanes
//   - (node, read_nodes_len) of steps i-1, i-2 for the 3-window: __shfl_up by 1 and 2
// Walks of <= 64 steps never straddle a wave (padded stream).  Longer walks (HiFi / ONT reads, hundreds of steps) get
// what lies in other waves from data prepared outside this kernel, all O(1) per step: a per-step "node occurred
// earlier in the walk" flag computed at upload (group_fill_long_kernel, LDS hash per walk), the sum of the node
// lengths before the last step from walk_sum_kernel, and plain loads for the two neighbours across a wave border.  The kernel
// is bound by the number of divergent (one cache line per lane) vector-memory instructions, so the
// tables it gathers from are packed into 16-byte records (one dwordx4 per lookup):
//   read_rec[r] = {first step, #steps, pstart, pend}      node_rec[v] = {bit_off (u64), len, -}
//   lookup head of node v (rides in node_rec) = {first row, #rows} of the unique windows whose MIDDLE is v;  trio_ent[j] = {smaller end, larger end} -- j IS the row (round 5: rows are numbered in filing order)
// A read that reaches this kernel was binned to its species, so every node id lies inside the
// species' id range (rcls.rs:253-257) and the index panic of profile.rs:849 cannot occur; an
// out-of-range id (inconsistent external binning) is counted as an abort per step instead.
//
// Algorithmic bytes per launch (the a8 row minus its popcount pass):
//   4T + 12R + 4V(node_len) + 8V(bases) + L/8 (bitmap) + 12*(T-2R) (trio probes)
// Layout: node arrays of all resident species are concatenated; a node's coverage bitmap starts
// at bit bit_off[v] of one global bit vector (1 bit per graph base instead of the reference's
// 1 byte, profile.rs:776-781).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "common.hpp"
#include "primitives.hpp"
#include "wave.hpp"
// bools are combined with & and | on purpose in the coverage kernels (no short-circuit control flow: every operand is a plain comparison or a vote)
#pragma clang diagnostic ignored "-Wbitwise-instead-of-logical"

namespace ptx {

constexpr int COV_BLOCK = 256;

// The test-before-set must see other CUs' ORs to be worth anything: the ORs execute below the
// per-CU L1 (which is never refreshed by them), so the probe is an agent-scope load (sc1: L1
// bypass, served by L2).  A stale 0 only costs a redundant OR; bits never clear, so it is safe.
__device__ __forceinline__ uint32_t bm_peek(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Bits [g0,g1) of the coverage bit vector are marked through the workgroup's LDS bit window (words
// [bw0, bw0 + COV_BWIN) of the global vector); words outside the window take the global test-then-OR path.  The window is ORed into memory once per workgroup.
constexpr uint32_t COV_BWIN = 2048;   // 32-bit words: 64 kbit of graph bases
constexpr int COV_WIN = 2048;         // nodes in the LDS window of `bases` (a multiple of 64 nodes from a multiple of 64: the full-node flags flush as ballots);
                                      // 2048 steps of 1e7 reads over 3.2e7 nodes span ~1000 nodes: 1024 overflowed on most chunks
// the LDS windows of the coverage kernel live at file scope: helpers that received them as (generic) pointer arguments
// made this compiler emit an illegal null check of the shared-memory aperture
// One dynamic LDS block per workgroup: [WIN u32 `bases` window][COV_BWIN u32 bit window][WIN u8 full-node flags: a step covered the
// whole node, its bits are not marked one by one (popcount_kernel takes the length)].  WIN is a launch parameter of the
// short-read kernel (a multiple of 256 nodes) and COV_WIN in the general one; helpers address the block by offsets.
extern __shared__ uint32_t s_cov[];
#define S_WIN(i) s_cov[(i)]
#define S_BM(bmo, i) s_cov[(bmo) + (i)]
#define S_FULL(bmo, i) reinterpret_cast<uint8_t *>(s_cov + (bmo) + COV_BWIN)[(i)]
__host__ __device__ constexpr size_t cov_lds_bytes(int win) { return (size_t)win * 4 + COV_BWIN * 4 + (size_t)win; }
__device__ __forceinline__ void lds_or(uint32_t *__restrict__ bm, uint32_t bmo, uint64_t bw0, uint32_t bwn, uint64_t w, uint32_t m) {
    const uint64_t off = w - bw0;     // unsigned wrap: words below the window are out of range too
    if (off < bwn) {
        if ((S_BM(bmo, off) & m) != m) atomicOr(&S_BM(bmo, off), m);
    } else if ((bm_peek(&bm[w]) & m) != m) atomicOr(&bm[w], m);
}
__device__ __forceinline__ void mark_range(uint32_t *__restrict__ bm, uint32_t bmo, uint64_t bw0, uint32_t bwn, uint64_t g0, uint64_t g1) {
    if (g1 <= g0) return;
    uint64_t w0 = g0 >> 5, w1 = (g1 - 1) >> 5;
    uint32_t m0 = 0xFFFFFFFFu << (g0 & 31);
    uint32_t m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
    if (w0 == w1) lds_or(bm, bmo, bw0, bwn, w0, m0 & m1);
    else {
        lds_or(bm, bmo, bw0, bwn, w0, m0);
        for (uint64_t w = w0 + 1; w < w1; ++w) lds_or(bm, bmo, bw0, bwn, w, 0xFFFFFFFFu);
        lds_or(bm, bmo, bw0, bwn, w1, m1);
    }
}
// the same for a range that lies inside the LDS bit window: 32-bit positions relative to the window, LDS only
__device__ __forceinline__ void win_or(uint32_t bmo, uint32_t w, uint32_t m) { if ((S_BM(bmo, w) & m) != m) atomicOr(&S_BM(bmo, w), m); }
__device__ __forceinline__ void mark_window(uint32_t bmo, uint32_t r0, uint32_t r1) {
    if (r1 <= r0) return;
    const uint32_t w0 = r0 >> 5, w1 = (r1 - 1) >> 5;
    const uint32_t m0 = 0xFFFFFFFFu << (r0 & 31), m1 = 0xFFFFFFFFu >> (31 - ((r1 - 1) & 31));
    if (w0 == w1) win_or(bmo, w0, m0 & m1);
    else {
        win_or(bmo, w0, m0);
        for (uint32_t w = w0 + 1; w < w1; ++w) win_or(bmo, w, 0xFFFFFFFFu);
        win_or(bmo, w1, m1);
    }
}

// Step codes (g_step_dup): walks of <= 64 steps carry the distance back to the first occurrence of the step's node in
// the walk (0 = none); longer walks carry STEP_LONG | (1 if the node occurred earlier in the walk).  Where the first
// occurrence sits matters only through "is it step 0" (profile.rs:853-856 vs :860-862), i.e. id == id of step 0.
constexpr uint32_t STEP_LONG = 0x80u;
// ... and, both kinds, STEP_START on the first step of a walk; pad steps carry STEP_PAD.  The slots of the grouped copy follow
// the stream (build_step_read lays the walks out in slot order), so a step's slot is not stored per step: group_slot[g] names
// the read that owns the first step of the 64-step group g, and every later walk start in the group advances it by one.
constexpr uint32_t STEP_START = 0x40u, STEP_PAD = 0xFFu, STEP_DIST = 0x3Fu;
// ballots / votes of a bool WITHOUT the detour through an int predicate (__ballot(int) costs a v_cndmask + v_cmp per call: the kernel is bound by VALU issue)
__device__ __forceinline__ unsigned long long ballot1(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool any1(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
__device__ __forceinline__ bool none1(bool b) { return __builtin_amdgcn_ballot_w64(b) == 0ull; }
__device__ __forceinline__ uint32_t slot_in_group(uint32_t group_first_slot, uint32_t code, int lane) {
    const unsigned long long starts = ballot1((code != STEP_PAD) & ((code & STEP_START) != 0u)) & ~1ull;   // lane 0's walk is group_first_slot itself
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u));   // starts in lower lanes
    return group_first_slot + below + (uint32_t)((starts >> lane) & 1ull);
}

// read_nodes_len of position j (never the last position) of a long walk: the length aligned at the node's FIRST
// occurrence in the read (profile.rs:879-882)
__device__ __forceinline__ uint32_t rl_from_memory(uint32_t j, uint32_t b, const uint32_t *__restrict__ node_id, const uint8_t *__restrict__ step_dup,
                                                   uint32_t delta, const uint4 *__restrict__ node_rec, uint32_t len0, uint32_t ps) {
    const uint32_t idj = node_id[b + j];
    if (j == 0 || ((step_dup[b + j] & 1u) && idj == node_id[b])) return len0 - ps;
    return node_rec[idj + delta].z;
}

// -DCOV_ABLATE builds (never the product library: make OUT=../lib_prof EXTRA=-DCOV_ABLATE, loaded through PANTAX_HIP_LIB) read
// PANTAX_COV_ABLATE: bit 0 no bit / flag marking, bit 1 no `bases`, bit 2 no unique-trio lookups -- wrong results, for timing only
#ifdef COV_ABLATE
#define ABL(bit) (ablate & (bit))
#else
#define ABL(bit) false
#endif
constexpr int COV_WIN_BACK = 128; // window starts (at least) this many nodes before the node of the chunk's first live step
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;

__device__ __forceinline__ void add_bases(unsigned long long *__restrict__ bases, uint32_t wlo, uint32_t win_n, uint32_t v, uint32_t aln) {
    const uint32_t off = v - wlo;   // unsigned wrap puts nodes below the window out of range too
    if (off < win_n && aln < (1u << 18)) atomicAdd(&S_WIN(off), aln);   // <= 8192 steps x 2^18 < 2^32
    else atomicAdd(&bases[v], (unsigned long long)aln);
}
// The full-node flag of node v (v = NO_FULL: none) -- called by ALL 64 lanes of a wave.  Inside the LDS window: plain byte stores of the
// same value (no atomic, nothing to lose).  Outside: the steps of a long walk are neighbouring nodes, so dozens of lanes would OR
// into the SAME 32-bit word -- memory-side atomics on one address run one after the other (1.0 of the kernel's 2.6 ms at cfg5's share);
// the lanes of one word combine their bits first (DPP reduction) and the first of them issues ONE atomic per distinct word.
constexpr uint32_t NO_FULL = 0xFFFFFFFFu;
__device__ __forceinline__ void mark_full_wave(uint32_t *__restrict__ full, uint32_t wlo, uint32_t win_n, uint32_t v) {
    const uint32_t off = v - wlo;
    const bool have = v != NO_FULL;
    if (have && off < win_n) S_FULL(COV_WIN, off) = 1;
    const bool out = have && off >= win_n;
    unsigned long long todo = __ballot(out);
    const uint32_t w = v >> 5, bit = 1u << (v & 31);
    const int lane = threadIdx.x & 63;
    while (todo) {                                 // (wave-uniform) one round per distinct word: two to four for a stretch of a walk
        const int leader = __builtin_ctzll(todo);
        const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, leader);
        const bool mine = out && w == wl;
        const uint32_t orv = wave_reduce(mine ? bit : 0u, [](uint32_t x, uint32_t y) { return x | y; });
        if (lane == leader) atomicOr(&full[wl], orv);   // no test-before-set: the probe is a dependent round trip, the OR is fire-and-forget
        todo &= ~__ballot(mine);
    }
}

// ---------------------------------------------------------------------------------------------
// The short-read kernel: every 64-step group whose walks all have <= 64 steps (no STEP_LONG code: all of a short-read
// sample).  Such walks never straddle a group, so a wave holds whole reads and NOTHING of a step's read lives outside the
// wave: no border lanes, no walk sums, no per-step slot array.  Written for instruction count -- the kernel is bound by VALU /
// SALU issue at full occupancy, not by bytes (round 2: 337 VALU + 270 SALU wave-instructions per 64 steps, most of the SALU
// from exec-mask branches): loads are unconditional with a safe index on dead lanes, per-lane cases are selects, and only
// the LDS / memory updates sit under a mask.  Groups that hold a step of a longer walk are left to coverage_step_kernel.
// Levels: {code, node id} + group_slot (scalar) -> {read record 16 B, slot record 8 B} -> {node record 16 B, active byte}
// -> two unique-trio entries.
// Workgroup = one ITEM of the read layout (build_step_read): up to COV_ITEM_GROUPS consecutive groups whose reads all START inside one
// block of 2048 node ids.  (Round 3 gave every workgroup a fixed number of groups: where a species is thinly covered -- most species of a
// Dirichlet-distributed sample -- 4096 steps span more nodes than the LDS windows hold, and every update outside them is a memory-side
// atomic: 3.3 of the kernel's 10.6 ms at 1e4 strains.  By node block the windows cover the block whatever the depth, and a deeply
// covered block is simply cut into more items.)
constexpr uint32_t COV_ITEM_GROUPS = 128;     // (64 until the end of round 5: a block of 2048 ids holds ~80 groups at 1e8 reads, cut as 64 + 17; whole blocks as ONE item: 5.2 -> 5.0 ms)
struct __attribute__((packed, aligned(4))) EntPair { uint32_t a, b, c, d; };     // two neighbouring lookup entries {smaller end, larger end}
constexpr int COV_BLK_SHIFT = 11;
//
// LONG (round 6): the same select-only body for the groups that hold steps of walks of MORE than 64 steps (HiFi / ONT reads; round 5 sent them through
// coverage_step_kernel, whose per-lane branches for wave-straddling walks -- dependent loads under an exec mask for the two neighbours across a wave
// border, for the first node of the walk, for the walk sums -- made it the kernel furthest from its roofline: 0.16 of peak with no wasted traffic).
// What a step of such a walk needs from OTHER waves is fetched unconditionally and wave-uniformly:
//   * the two steps in front of the wave (ids, step codes, node records): addresses that depend on the group alone -> scalar loads, issued with the
//     level they belong to (ids / codes with the stream, node records with the node gather), never a dependent load behind a per-lane test;
//   * per slot {length of the walk's first node, sum of the node lengths before the last step} (walk_sum_kernel) and the id of the walk's first
//     step: gathers with the read record / the node record, for every lane (the lanes of one walk share the address);
// and every per-lane case is a select.  only_long != 0: groups without a step of a longer walk are the plain instantiation's.
template <int WIN>
__device__ __forceinline__ void mark_full_out_wave(uint32_t *__restrict__ full, bool out, uint32_t v) {
    // the steps of a long walk are neighbouring nodes: dozens of lanes would OR into the SAME word, and memory-side atomics on one address run one
    // after the other -- the lanes of a word combine their bits first and ONE of them issues the atomic (wave-uniform loop, called by all 64 lanes)
    unsigned long long todo = ballot1(out);
    const uint32_t w = v >> 5, bit = 1u << (v & 31);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, leader);
        const bool mine = out & (w == wl);
        const uint32_t orv = wave_reduce(mine ? bit : 0u, [](uint32_t x, uint32_t y) { return x | y; });
        if (lane == leader) atomicOr(&full[wl], orv);
        todo &= ~ballot1(mine);
    }
}
template <bool WITH_TRIO, int U, int PASSES, int WIN, bool LONG = false>
__global__ void __launch_bounds__(COV_BLOCK) coverage_fast_kernel(
    const uint2 *__restrict__ items, const uint32_t *__restrict__ group_slot, const uint4 *__restrict__ read_rec, const uint2 *__restrict__ slot_rec,
    const uint32_t *__restrict__ node_id, const uint8_t *__restrict__ step_code, const uint8_t *__restrict__ active,
    const uint4 *__restrict__ node_rec, const uint64_t *__restrict__ bit_off, uint64_t V, unsigned long long *__restrict__ bases,
    uint32_t *__restrict__ bitmap, uint32_t *__restrict__ full, const uint2 *__restrict__ trio_ent, unsigned long long *__restrict__ trio_bases,
    unsigned long long *__restrict__ n_abort, uint32_t ablate, int blk_shift,
    const uint32_t *__restrict__ long_sum = nullptr, const uint32_t *__restrict__ long_len0 = nullptr, uint32_t only_long = 0u,
    uint32_t chunk_groups = 0u, uint32_t total_groups = 0u, uint32_t win_back = 0u, const uint32_t *__restrict__ item_sel = nullptr) {
    constexpr int WAVES = COV_BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this workgroup's groups [g0, n_groups): an item of the read layout -- or, LONG, a plain cut of the stream: a long walk begins anywhere in a group,
    // so hardly any group starts with the first step of a read and the layout's items are arbitrary cuts anyway; the windows then begin `win_back` nodes
    // in front of the first live step's node (a reverse-strand walk runs DOWN from its first node, the key the stream is ordered by)
    uint32_t g0, n_groups;
    if constexpr (LONG) { g0 = blockIdx.x * chunk_groups; n_groups = min(g0 + chunk_groups, total_groups); }
    else { const uint2 item = items[item_sel ? item_sel[blockIdx.x] : blockIdx.x]; g0 = item.x; n_groups = item.y; }   // (item_sel: the items a db of SOME of the species launches)
    for (int i = threadIdx.x; i < (int)(cov_lds_bytes(WIN) / 4); i += COV_BLOCK) s_cov[i] = 0;      // the three windows, one block
    // window base: the node of the first step of the first group that has a live one (workgroup-uniform scalar loads)
    uint32_t wlo = 0, win_n = 0, mark_n = 0, bit0_lo = 0, bwn = 0;
    uint64_t bw0 = 0;
#pragma unroll 1
    for (uint32_t g = g0; g < n_groups && win_n == 0; ++g) {
        const uint32_t gs = group_slot[g];
        if (gs == NO_SLOT) continue;
        const uint2 sr0 = slot_rec[gs];
        if ((int)sr0.x < 0 || !active[sr0.x]) continue;
        // the window starts at the item's NODE BLOCK, not at its first read: the reads of a bucket of the layout (512 ids wide at 3e8 nodes)
        // are in no particular order, so later reads of the item may start hundreds of nodes in front of the first one -- every read of
        // the item starts inside the block (build_step_read), and ids and node indices run in step inside a species
        const uint32_t id0 = node_id[(uint64_t)g * 64], v0 = id0 + sr0.y, in_blk = id0 & ((1u << blk_shift) - 1u);
        const uint32_t back = LONG ? win_back : in_blk + 64u;
        wlo = (v0 > back ? v0 - back : 0u) & ~63u;
        win_n = WIN;
        const uint64_t b_lo = bit_off[wlo], b_hi = bit_off[min((uint64_t)wlo + WIN, V)];
        bw0 = b_lo >> 5;
        bit0_lo = (uint32_t)(bw0 << 5);
        bwn = COV_BWIN;
        // every node of the window has its bits inside the LDS bit window: a partial range is marked in 32-bit positions relative to it
        mark_n = (b_hi - (bw0 << 5) <= (uint64_t)COV_BWIN * 32) ? (uint32_t)min((uint64_t)WIN, V - wlo) : 0u;
    }
    // the barrier orders the LDS zero-fill only (a workgroup fence on the local address space): the window probes above are
    // still in flight and are first needed by the updates of the first pass, behind that pass's own three levels of loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    // Level 1 of round r + 1 is requested at the top of round r (round 5): the stream loads are the one level that comes from HBM every time, and the
    // kernel waits for its four dependent levels two thirds of the time -- three more registers per group in flight (64 in all: still eight waves per
    // SIMD) take the first level off the chain: 5.70 -> 5.21 ms at 1e8 reads, 0.62 -> 0.57 at 1e7 (same box, alternating builds).  -DCOV_NO_PREFETCH:
    // the round-4 loop, for measurements.
#ifndef COV_NO_PREFETCH
    uint32_t n_code[U], n_id[U], n_gs[U];
    {
        const uint32_t gw0 = g0 + (uint32_t)(wave * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = gw0 + (uint32_t)u < n_groups ? gw0 + (uint32_t)u : (gw0 < n_groups ? gw0 : g0);
            const uint64_t t = (uint64_t)g * 64 + lane;
            n_code[u] = step_code[t]; n_id[u] = node_id[t]; n_gs[u] = group_slot[g];
        }
    }
#endif
    // The plain instantiation takes TWO levels off the chain (round 6): the stream loads run two rounds ahead and the read / slot records of the coming
    // round are requested at the top of this one -- nine more registers per group in flight, 5.30 -> 5.13 ms at 1e8 reads (-DCOV_NO_PF2: one level)
#if !defined(COV_NO_PF2) && !defined(COV_NO_PREFETCH)
    constexpr bool PF2 = !LONG;
#else
    constexpr bool PF2 = false;
#endif
    uint32_t c_code[U], c_id[U], c_gs[U];      // PF2: this round's level 1 ...
    uint4 c_rr[U];                             // ... and level 2, requested a round ago
    uint2 c_sr[U];
    bool c_run[U];                             // ... and whether the group is this kernel's at all (wave-uniform: decided once, when its records are requested)
#pragma unroll
    for (int u = 0; u < U; ++u) c_run[u] = false;
    if constexpr (PF2) {
        const uint32_t gw0 = g0 + (uint32_t)(wave * U), gw1 = gw0 + (uint32_t)(WAVES * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c_code[u] = n_code[u]; c_id[u] = n_id[u]; c_gs[u] = n_gs[u];           // round 0 (requested above)
            const uint32_t g = gw1 + (uint32_t)u < n_groups ? gw1 + (uint32_t)u : (gw0 < n_groups ? gw0 : g0);
            const uint64_t t = (uint64_t)g * 64 + lane;
            n_code[u] = step_code[t]; n_id[u] = node_id[t]; n_gs[u] = group_slot[g];   // round 1
            const bool pad0 = c_code[u] == STEP_PAD;
            const bool run0 = (gw0 + (uint32_t)u < n_groups) & any1(!pad0) & none1(!pad0 & ((c_code[u] & STEP_LONG) != 0u));
            const uint32_t sl = slot_in_group(c_gs[u], c_code[u], lane);
            const uint32_t slot = (pad0 | !run0) ? (c_gs[u] == NO_SLOT ? 0u : c_gs[u]) : sl;
            c_rr[u] = read_rec[slot]; c_sr[u] = slot_rec[slot];
            c_run[u] = run0;
        }
    }
#pragma unroll 1
    for (int pass = 0;; ++pass) {
        const uint32_t gw = g0 + (uint32_t)((pass * WAVES + wave) * U);     // this wave's U consecutive groups
        if (gw >= n_groups) break;
        // ---- level 1
        uint32_t code[U], id[U], gs[U];
        bool run[U];                                                          // wave-uniform: the group is this kernel's
        uint4 rr[U];
        uint2 sr[U];
#ifndef COV_NO_PREFETCH
        if constexpr (PF2) {
            const uint32_t gn = gw + (uint32_t)(WAVES * U), g2 = gn + (uint32_t)(WAVES * U);   // the coming round's groups, and the one behind it
            uint32_t m_code[U], m_id[U], m_gs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                run[u] = c_run[u];
                code[u] = c_code[u]; id[u] = c_id[u]; gs[u] = c_gs[u]; rr[u] = c_rr[u]; sr[u] = c_sr[u];
                const uint32_t g = g2 + (uint32_t)u < n_groups ? g2 + (uint32_t)u : gw;
                const uint64_t t = (uint64_t)g * 64 + lane;
                m_code[u] = step_code[t]; m_id[u] = node_id[t]; m_gs[u] = group_slot[g];
                // level 2 of the coming round (its level 1 was requested a round ago)
                const bool padn = n_code[u] == STEP_PAD;
                const bool runn = (gn + (uint32_t)u < n_groups) & any1(!padn) & none1(!padn & ((n_code[u] & STEP_LONG) != 0u));
                const uint32_t sl = slot_in_group(n_gs[u], n_code[u], lane);
                const uint32_t slot = (padn | !runn) ? (n_gs[u] == NO_SLOT ? 0u : n_gs[u]) : sl;
                c_rr[u] = read_rec[slot]; c_sr[u] = slot_rec[slot];
                c_run[u] = runn;
                c_code[u] = n_code[u]; c_id[u] = n_id[u]; c_gs[u] = n_gs[u];
                n_code[u] = m_code[u]; n_id[u] = m_id[u]; n_gs[u] = m_gs[u];
            }
        } else
        {
            const uint32_t gn = gw + (uint32_t)(WAVES * U);                    // the coming round's groups
#pragma unroll
            for (int u = 0; u < U; ++u) {
                run[u] = gw + (uint32_t)u < n_groups;
                code[u] = n_code[u]; id[u] = n_id[u]; gs[u] = n_gs[u];
                const uint32_t g = gn + (uint32_t)u < n_groups ? gn + (uint32_t)u : gw;
                const uint64_t t = (uint64_t)g * 64 + lane;
                n_code[u] = step_code[t]; n_id[u] = node_id[t]; n_gs[u] = group_slot[g];
            }
        }
#else
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = gw + (uint32_t)u;
            run[u] = g < n_groups;
            const uint64_t t = (uint64_t)(run[u] ? g : gw) * 64 + lane;
            code[u] = step_code[t]; id[u] = node_id[t];
            gs[u] = group_slot[run[u] ? g : gw];
        }
#endif
        // ---- level 2 (dead lanes read the group's first record: in range, and on a line that is fetched anyway)
        bool pad[U];
        uint32_t ll0[U], lsum[U];                                             // LONG: first node length / sum of the lengths before the last step, by slot
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pad[u] = code[u] == STEP_PAD;
            if constexpr (!PF2) {
                const bool has_long = any1(!pad[u] & ((code[u] & STEP_LONG) != 0u));
                // nothing here / a group of the OTHER instantiation (steps of a longer walk: LONG's; none: the plain one's, unless LONG takes every group)
                run[u] = run[u] & any1(!pad[u]) & (LONG ? (has_long | (only_long == 0u)) : !has_long);
            }
            const uint32_t sl = slot_in_group(gs[u], code[u], lane);
            const uint32_t slot = (pad[u] | !run[u]) ? (gs[u] == NO_SLOT ? 0u : gs[u]) : sl;
            if constexpr (!PF2) { rr[u] = read_rec[slot]; sr[u] = slot_rec[slot]; }     // (PF2: requested a round ago)
            if constexpr (LONG) { ll0[u] = long_len0[slot]; lsum[u] = long_sum[slot]; }
        }
        // LONG: the two steps in front of the wave's group (wave-uniform addresses: position of the group - 1, - 2; the dword of step codes in front of it)
        uint32_t h_id1[U], h_id2[U], h_codes[U];
        if constexpr (LONG) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t gu = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gw + (uint32_t)u));
                const uint64_t t0 = (uint64_t)(run[u] && gu != 0u ? gu : 1u) * 64;     // (group 0 has nothing in front of it: any in-range address)
                h_id1[u] = node_id[t0 - 1]; h_id2[u] = node_id[t0 - 2];
                h_codes[u] = *reinterpret_cast<const uint32_t *>(step_code + t0 - 4);
            }
        }
        // ---- level 3
        uint4 nr[U];
        uint32_t v[U], act[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = run[u] & !pad[u] & ((int)sr[u].x >= 0);
            v[u] = ok[u] ? id[u] + sr[u].y : wlo;
            nr[u] = node_rec[v[u]];
            act[u] = active[ok[u] ? sr[u].x : 0u];      // never null here (the launcher passes all-ones when no species is deselected): an unconditional load
                                                        // travels beside the node record; under a pointer test the compiler waited for it first
        }
        // LONG: id of the walk's first step (one address per walk) and the node records of the two steps in front of the wave (wave-uniform: the
        // walk of lane 0, when it began one / two steps or more before this group)
        uint32_t idf[U], i0_[U], hv1[U], hv2[U];
        uint4 nrh1[U], nrh2[U];
        if constexpr (LONG) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idf[u] = node_id[rr[u].x];
                const uint32_t gbase = (gw + (uint32_t)u) * 64u;
                const bool ok0 = __builtin_amdgcn_readfirstlane((int)ok[u]) != 0;
                i0_[u] = ok0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(gbase - rr[u].x)) : 0u;      // lane 0's position in its walk
                const uint32_t delta0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sr[u].y);
                hv1[u] = i0_[u] >= 1u ? h_id1[u] + delta0 : wlo;
                hv2[u] = i0_[u] >= 2u ? h_id2[u] + delta0 : wlo;
                nrh1[u] = node_rec[hv1[u]]; nrh2[u] = node_rec[hv2[u]];
            }
        }
        // ---- level 4: the unique-trio entries of the window (i-2, i-1, i), requested as soon as the head is known
        uint32_t i_[U], nl[U], len0[U], nh[U], hx[U], tlo[U], thi[U];
        uint2 e0[U], e1[U];
        bool live[U], single[U], dead_read[U], cross[U];
        int dist[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t gbase = (gw + (uint32_t)u) * 64u;
            i_[u] = gbase + (uint32_t)lane - rr[u].x;                         // position in the walk (T_pad < 2^32)
            ok[u] = ok[u] & (act[u] != 0u);
            nl[u] = ok[u] ? nr[u].z : 0u;
            if constexpr (LONG) {
                dist[u] = (int)min(ok[u] ? i_[u] : 0u, (uint32_t)lane);       // earlier steps of my walk held by lower lanes
                cross[u] = ok[u] & (i_[u] > (uint32_t)lane);                  // the walk began before this wave
                const uint32_t in_wave = __shfl(nl[u], lane - dist[u]);
                len0[u] = cross[u] ? ll0[u] : in_wave;
            } else
            len0[u] = __shfl(nl[u], lane - (int)i_[u]);                       // length of the walk's first node: the lane of step 0 (live lanes)
            uint32_t v2 = wave_shr1(wave_shr1(v[u]));
            uint32_t hw1 = wave_shr1(nr[u].w), hy1 = wave_shr1(nr[u].y);         // the lookup head of the window's MIDDLE node: the lane below
            if constexpr (LONG) {                                             // across the wave border: the records fetched for the steps in front of the group
                v2 = lane == 0 ? hv2[u] : lane == 1 ? hv1[u] : v2;
                hw1 = lane == 0 ? nrh1[u].w : hw1; hy1 = lane == 0 ? nrh1[u].y : hy1;
            }
            single[u] = rr[u].y == 1u;
            dead_read[u] = !single[u] & (rr[u].z > len0[u]);                  // assert :854 -> the whole read contributes nothing
            live[u] = ok[u] & !dead_read[u] & !(single[u] & (rr[u].w < rr[u].z));   // :821-827
            nh[u] = 0; hx[u] = 0; tlo[u] = 0; thi[u] = 0;
            e0[u] = make_uint2(0u, 0u); e1[u] = e0[u];
            if (WITH_TRIO && !ABL(4u)) {
                // canonical window (min end, middle, max end): the rows are filed under the MIDDLE node, keyed by the two ends
                hx[u] = hw1;
                tlo[u] = min(v[u], v2); thi[u] = max(v[u], v2);
                // the node's pair filter: a window whose bit is clear is not among its rows -- nothing is fetched for it
                nh[u] = (live[u] & (i_[u] >= 2u) & ((nr_filter(hy1) & nr_pair_bit(tlo[u], thi[u])) != 0u)) ? nr_rows(hy1) : 0u;
                // the head's first TWO entries in one 16-byte load (entries are 8 bytes since round 5; the pair is dword-aligned, and the array has one
                // entry of slack behind its last row)
                const EntPair ep = *reinterpret_cast<const EntPair *>(trio_ent + (nh[u] ? hx[u] : 0u));
                e0[u] = make_uint2(ep.a, ep.b); e1[u] = make_uint2(ep.c, ep.d);
            }
        }
        // ---- per group: aligned lengths and the updates
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = rr[u].y, ps = rr[u].z, pe = rr[u].w, i = i_[u];
            if (any1(ok[u] & dead_read[u] & (i == 0u))) { if (ok[u] & dead_read[u] & (i == 0u)) atomicAdd(n_abort, 1ull); }
            // `seen` before this step = wave prefix sum of the walk's aligned lengths minus its value at the walk's first lane
            const uint32_t contrib = (live[u] & !single[u]) ? (i == 0u ? nl[u] - ps : nl[u]) : 0u;
            const uint32_t pexcl = wave_incl_scan_dpp(contrib) - contrib;
            uint32_t seen = pexcl - __shfl(pexcl, lane - (LONG ? dist[u] : (int)i));
            if constexpr (LONG) seen = cross[u] ? lsum[u] - ps : seen;        // (used by a walk's last step only) all steps but the last: walk_sum_kernel
            const uint32_t tgt = pe - ps;                                     // target (profile.rs:800) where it is not negative
            uint32_t aln = nl[u];                                             // :860-862
            if (i + 1u == k) aln = ((pe >= ps) & (tgt > seen)) ? tgt - seen : 0u;   // :857-859 max(target - seen, 0)
            if (i == 0u) aln = single[u] ? tgt : nl[u] - ps;                  // :853-856, :828
            const uint32_t sidx = i == 0u ? ps : 0u;
            uint32_t hi = sidx + aln;
            if (hi > nl[u]) hi = nl[u];                                       // :871
            const bool markable = live[u] & (hi > sidx) & !(single[u] & !((ps < pe) & (pe <= nl[u])));   // :832
            const uint32_t dupd = code[u] & STEP_DIST;                        // distance back to the node's first occurrence in the walk (0: this is it)
            bool earlier = dupd != 0u, at_step0 = dupd == i;                  // the node occurred earlier in the walk / its first occurrence is step 0
            if constexpr (LONG) {                                             // a longer walk's code: STEP_LONG | occurred earlier
                const bool lc = (code[u] & STEP_LONG) != 0u;
                const uint32_t id_first = cross[u] ? idf[u] : __shfl(id[u], lane - dist[u]);
                earlier = lc ? (code[u] & 1u) != 0u : earlier;
                at_step0 = lc ? id[u] == id_first : at_step0;
            }
            const uint32_t rl = !live[u] ? 0u : !earlier ? aln : (at_step0 ? len0[u] - ps : nl[u]);   // read_nodes_len :879-882
            const uint32_t off = v[u] - wlo;                                  // unsigned wrap: nodes below the window are out of range too
            const bool inw = off < win_n;
            if (live[u] & !earlier & (aln != 0u) && !ABL(2u)) {               // :881 / :828
                if (inw & (aln < (1u << 18))) atomicAdd(&S_WIN(off), aln);
                else atomicAdd(&bases[v[u]], (unsigned long long)aln);
            }
            const bool whole = markable & (sidx == 0u) & (hi == nl[u]);       // the whole node: one flag
            if constexpr (LONG) { if (!ABL(1u) && !ABL(8u)) mark_full_out_wave<WIN>(full, whole & !inw, v[u]); }
            if (markable && !ABL(1u)) {
                if (whole) {
                    if (inw) S_FULL(WIN, off) = 1; else if (!LONG && !ABL(8u)) atomicOr(&full[v[u] >> 5], 1u << (v[u] & 31));
                } else if (off < mark_n) {
                    const uint32_t rel = nr[u].x - bit0_lo;
                    mark_window(WIN, rel + sidx, rel + hi);
                } else if (!ABL(8u)) {
                    const uint64_t bo = nr_bit_off(nr[u]);
                    mark_range(bitmap, WIN, bw0, bwn, bo + sidx, bo + hi);
                }
            }
            if (WITH_TRIO && !ABL(4u)) {                                      // :890-907
                uint32_t rl1 = wave_shr1(rl), rl2 = wave_shr1(rl1);
                if constexpr (LONG) {
                    // read_nodes_len of the two steps in front of the wave (never a walk's last step): the length aligned at the node's FIRST occurrence
                    // in the read -- wave-uniform values of lane 0's walk (lane 1 uses them only when it continues that walk)
                    const uint32_t idf0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)idf[u]), l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)len0[u]),
                                   ps0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ps);
                    const uint32_t c1 = h_codes[u] >> 24, c2 = (h_codes[u] >> 16) & 0xFFu;
                    const uint32_t rlh1 = ((i0_[u] == 1u) | (((c1 & 1u) != 0u) & (h_id1[u] == idf0))) ? l0 - ps0 : nrh1[u].z;
                    const uint32_t rlh2 = ((i0_[u] == 2u) | (((c2 & 1u) != 0u) & (h_id2[u] == idf0))) ? l0 - ps0 : nrh2[u].z;
                    rl2 = lane == 0 ? rlh2 : lane == 1 ? rlh1 : rl2;
                    rl1 = lane == 0 ? rlh1 : rl1;
                }
                // a row IS its lookup entry (round 5): the index of the entry that matches (rows are 32-bit; NO_ROW: none)
                constexpr uint32_t NO_ROW = 0xFFFFFFFFu;
                const bool m0 = (nh[u] != 0u) & (e0[u].x == tlo[u]) & (e0[u].y == thi[u]);
                const bool m1 = (nh[u] > 1u) & (e1[u].x == tlo[u]) & (e1[u].y == thi[u]);
                uint32_t row = m0 ? hx[u] : m1 ? hx[u] + 1u : NO_ROW;
                if (any1(!(m0 | m1) & (nh[u] > 2u))) {
                    if (!(m0 | m1) & (nh[u] > 2u))
                        for (uint32_t j = 2; j < nh[u]; ++j) {
                            const uint2 e = trio_ent[hx[u] + j];
                            if (e.x == tlo[u] && e.y == thi[u]) { row = hx[u] + j; break; }
                        }
                }
                const uint32_t sum = rl2 + rl1 + rl;                             // three node lengths: far below 2^32
                if ((row != NO_ROW) & (sum != 0u)) atomicAdd(&trio_bases[row], (unsigned long long)sum);
            }
        }
    }
    __syncthreads();
    if (win_n) {
        for (int i = threadIdx.x; i < WIN; i += COV_BLOCK) {
            const uint32_t c = S_WIN(i);
            if (c) atomicAdd(&bases[wlo + i], (unsigned long long)c);
            const unsigned long long fb = __ballot(S_FULL(WIN, i) != 0);
            if (fb && (lane & 31) == 0 && !ABL(32u)) {
                const uint32_t m = (uint32_t)(fb >> (lane & 32));
                if (m) atomicOr(&full[(wlo + i) >> 5], m);
            }
        }
        if (!ABL(16u))
        for (uint32_t i = threadIdx.x; i < bwn; i += COV_BLOCK) {
            const uint32_t m = S_BM(WIN, i);
            if (m) atomicOr(&bitmap[bw0 + i], m);
        }
    }
}

// Steps arrive grouped by the locus of their read's first node (build_step_read below), so a workgroup's
// chunk of consecutive steps lands in a narrow node window: `bases` is accumulated in an LDS
// window of COV_WIN nodes (32-bit LDS atomics) and flushed with one 64-bit global atomic per touched
// node -- the LDS-staged segmented reduction of the scatter.  Nodes outside the window (or oversized
// lengths) fall back to the global atomic; the result is identical either way.
//
// The kernel is bound by instruction issue and by the latency of its chain of dependent gathers, so
//   * the chain is three levels: {slot, node id, step code} (stream) -> {read record (16 B), slot record (8 B)} -> {node record}
//     (-> a unique-trio entry where the node has any).  The slot record {species, node base - first id} is written by the
//     binning pass; a binned walk lies inside its species' range and db_upload makes the range span exactly the graph, so a
//     step places its node with ONE add and no range test; the node record carries the lookup head of the unique-trio index (first row, #rows)
//     next to bit offset and length, and a 3-window takes the head of its middle node from the lane below:
//     ONE divergent 16-byte gather per step.  Everything is in global node indices (the lookup entries too).
//   * a step that covers its whole node (every interior step of a read: profile.rs:860-862 with :870-873) sets ONE flag for
//     the node instead of marking its bits word by word (popcount_kernel then takes the node's length); only the partial
//     ranges -- first and last step of a read -- are marked in the bit window, in 32-bit positions relative to the window.
//   * every wave works on U groups of 64 steps at once: the loads of one level are issued for all U groups before the
//     first of them is waited for (U x the memory-level parallelism per wave; registers permitting).
// PASSES such rounds share one set of LDS windows (zeroing and flushing them is per workgroup).  Workgroups are handed to
// the XCDs round-robin by the dispatcher; XCD_MAP makes every XCD walk ONE contiguous eighth of the stream, so neighbouring
// chunks -- which share the node records and bitmap lines at their seam -- meet in the same L2.
template <bool WITH_TRIO, int U, int PASSES>
__global__ void __launch_bounds__(COV_BLOCK) coverage_step_kernel(
    uint64_t T, const uint32_t *__restrict__ group_slot, const uint4 *__restrict__ read_rec, const uint2 *__restrict__ slot_rec,
    const uint32_t *__restrict__ node_id, const uint8_t *__restrict__ step_dup, const uint8_t *__restrict__ active,
    const uint4 *__restrict__ node_rec, unsigned long long *__restrict__ bases, uint32_t *__restrict__ bitmap, uint32_t *__restrict__ full,
    const uint2 *__restrict__ trio_ent, unsigned long long *__restrict__ trio_bases, unsigned long long *__restrict__ n_abort,
    const uint32_t *__restrict__ long_sum, const uint32_t *__restrict__ long_len0, uint32_t n_chunks, uint32_t xcd_map, uint32_t ablate,
    uint32_t only_long /* 1: groups without a step of a longer walk belong to coverage_fast_kernel */) {
    constexpr int CHUNK = COV_BLOCK * U * PASSES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t chunk = blockIdx.x;
    if (xcd_map) {   // blockIdx -> XCD is round-robin over 8: XCD x takes chunks [x * per, (x + 1) * per)
        const uint32_t per = (n_chunks + 7) / 8;
        chunk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per || chunk >= n_chunks) return;
    }
    const uint64_t chunk_b = (uint64_t)chunk * CHUNK;
    uint64_t chunk_e = chunk_b + CHUNK;
    if (chunk_e > T) chunk_e = T;
    for (int i = threadIdx.x; i < (int)(cov_lds_bytes(COV_WIN) / 4); i += COV_BLOCK) s_cov[i] = 0;
    // window base: the node of the first live step among a few probes of the chunk.  Every thread computes it
    // (workgroup-uniform addresses), so nobody waits on a broadcast and the probes overlap the first gathers.
    uint32_t wlo = 0, win_n = 0;
    uint64_t bw0 = 0, bit0 = 0;
    uint32_t bwn = 0;
#pragma unroll
    for (int c = 0; c < CHUNK / COV_BLOCK; ++c) {
        const uint64_t tc = chunk_b + (uint64_t)c * COV_BLOCK;
        if (win_n == 0 && tc < chunk_e) {
            const uint32_t slot = group_slot[tc >> 6];          // tc is a multiple of 64: the read that owns the group's first step
            if (slot != NO_SLOT) {
                const uint2 sr0 = slot_rec[slot];
                if ((int)sr0.x >= 0 && !(active && !active[sr0.x])) {
                    const uint32_t v0 = node_id[tc] + sr0.y;
                    wlo = (v0 > (uint32_t)COV_WIN_BACK ? v0 - COV_WIN_BACK : 0u) & ~63u;
                    win_n = COV_WIN;
                    bw0 = nr_bit_off(node_rec[wlo]) >> 5;        // bit window starts at the window's first node
                    bit0 = bw0 << 5;
                    bwn = COV_BWIN;
                }
            }
        }
    }
    __syncthreads();
    // the stream loads of round r + 1 are requested at the top of round r, as in the short-read kernel (-DCOV_NO_PREFETCH: the round-4 loop)
#ifndef COV_NO_PREFETCH
    uint32_t n_id[U], n_dupc[U], n_gs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t t = chunk_b + (uint64_t)(wave * U + u) * 64 + lane;
        const uint64_t tc = t < chunk_e ? t : chunk_b;              // (in range; dead lanes of a round are masked by `ok`)
        n_id[u] = node_id[tc]; n_dupc[u] = step_dup[tc]; n_gs[u] = group_slot[tc >> 6];
    }
#endif
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const uint64_t wbase = chunk_b + (uint64_t)((pass * (COV_BLOCK / 64) + wave) * U) * 64;   // this wave's U x 64 consecutive steps
        if (wbase >= chunk_e) break;
        // ---- level 1: the stream
        uint32_t slot[U], id[U], dupc[U], ti[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t t = wbase + (uint64_t)u * 64 + lane;
            ok[u] = t < chunk_e;                                     // whole groups: T_pad and the chunk size are multiples of 64
            slot[u] = NO_SLOT; id[u] = 0; dupc[u] = STEP_PAD;
            ti[u] = (uint32_t)t;                                     // T_pad < 2^32 (build_step_read)
#ifndef COV_NO_PREFETCH
            uint32_t gs_now = NO_SLOT;
            if (ok[u]) { id[u] = n_id[u]; dupc[u] = n_dupc[u]; gs_now = n_gs[u]; }
            {
                const uint64_t tn = t + (uint64_t)COV_BLOCK * U;    // the same lane's step in the coming round
                const uint64_t tc = (pass + 1 < PASSES && tn < chunk_e) ? tn : chunk_b;
                n_id[u] = node_id[tc]; n_dupc[u] = step_dup[tc]; n_gs[u] = group_slot[tc >> 6];
            }
            const uint32_t gs = gs_now;
#else
            if (ok[u]) { id[u] = node_id[t]; dupc[u] = step_dup[t]; }
            const uint32_t gs = ok[u] ? group_slot[t >> 6] : NO_SLOT;
#endif
            const uint32_t sl = slot_in_group(gs, dupc[u], lane);
            ok[u] = ok[u] && dupc[u] != STEP_PAD;
            if (only_long && !__any(ok[u] && (dupc[u] & STEP_LONG))) ok[u] = false;   // a short-read group: the other kernel's
            if (ok[u]) slot[u] = sl;
        }
        // ---- level 2: per-read records
        uint4 rr[U];
        uint2 sr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rr[u] = make_uint4(0u, 0u, 0u, 0u); sr[u] = make_uint2(0xFFFFFFFFu, 0u);
            if (ok[u]) { rr[u] = read_rec[slot[u]]; sr[u] = slot_rec[slot[u]]; }
        }
        // ---- level 3: the node record (issued before the species' `active` flag is known: a wasted gather at worst)
        uint4 nr[U];
        uint32_t v[U], act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = ok[u] && (int)sr[u].x >= 0;                       // "U" / dropped rows
            nr[u] = make_uint4(0u, 0u, 0u, 0u); v[u] = 0; act[u] = 1u;
            if (ok[u]) {
                v[u] = id[u] + sr[u].y;
                nr[u] = node_rec[v[u]];
                if (active) act[u] = active[sr[u].x];
            }
        }
        // ---- shuffles, the trio lookup head and the first trio entry (level 4), for all groups
        uint32_t v1[U], v2[U], len0[U], tlo[U], thi[U];
        uint2 th[U];
        uint2 e0[U], e1[U];   // the first TWO lookup entries of the head: with one, 95 % of the waves held a lane whose window was
                              // the node's second entry (8 % of the visits meet a head of two or more) and paid another dependent gather
        int dist[U];
        bool cross[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u] && !act[u]) ok[u] = false;                      // unselected species
            if (!ok[u]) { v[u] = 0; nr[u] = make_uint4(0u, 0u, 0u, 0u); }
            const uint32_t b = rr[u].x;
            const uint32_t i = ok[u] ? ti[u] - b : 0u;
            dist[u] = ok[u] ? (int)min(i, (uint32_t)lane) : 0;        // earlier steps of my read held by lower lanes
            cross[u] = ok[u] && (int)i > lane;                        // the walk began before this wave (more than 64 steps)
            // neighbours one and two lanes down: DPP wave shifts (VALU), not LDS-crossbar shuffles
            v1[u] = wave_shr1(v[u]); v2[u] = wave_shr1(v1[u]);
            const uint32_t tf1 = wave_shr1(nr[u].w), ty1 = wave_shr1(nr[u].y);
            th[u] = make_uint2(0u, 0u); tlo[u] = 0; thi[u] = 0; e0[u] = make_uint2(0u, 0u); e1[u] = make_uint2(0u, 0u);
            if (WITH_TRIO && !ABL(4u) && ok[u] && i >= 2) {
                if (lane < 1) v1[u] = node_id[b + i - 1] + sr[u].y;
                if (lane < 2) v2[u] = node_id[b + i - 2] + sr[u].y;
                // canonical window (min end, middle, max end); the lookup rows are filed under the MIDDLE node (the lane below), keyed by the two ends
                tlo[u] = min(v[u], v2[u]); thi[u] = max(v[u], v2[u]);
                uint32_t hy = ty1;
                if (lane >= 1) th[u].x = tf1;
                else { const uint4 r1 = node_rec[v1[u]]; th[u].x = r1.w; hy = r1.y; }   // wave border of a long walk
                th[u].y = (nr_filter(hy) & nr_pair_bit(tlo[u], thi[u])) ? nr_rows(hy) : 0u;   // the pair filter: nothing is fetched for a window whose bit is clear
                if (th[u].y) {
                    const EntPair ep = *reinterpret_cast<const EntPair *>(trio_ent + th[u].x);   // two entries, one load (one entry of slack behind the last row)
                    e0[u] = make_uint2(ep.a, ep.b); e1[u] = make_uint2(ep.c, ep.d);
                }
            }
            // first node length: from the lane that holds step b, else (long walk) noted by walk_sum_kernel
            const uint32_t nl_src = __shfl(nr[u].z, lane - dist[u]);
            len0[u] = nr[u].z;
            if (ok[u] && i > 0) len0[u] = !cross[u] ? nl_src : long_len0[slot[u]];
        }
        // ---- per group: aligned lengths, bitmap, bases, trio bases
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t b = rr[u].x, k = rr[u].y, ps = rr[u].z, pe = rr[u].w;
            // positions, node lengths and aligned lengths are 32-bit quantities (the packed layout carries u32 columns and a
            // walk cannot align 4 Gbp): only `target` needs a sign.
            const uint32_t i = ok[u] ? ti[u] - b : 0u;
            const uint64_t bo = nr_bit_off(nr[u]);
            const uint32_t nl = nr[u].z;
            const uint64_t rel = bo - bit0;                           // the node inside the LDS bit window: 32-bit relative positions
            const bool in_win = rel < (uint64_t)bwn * 32 && nl <= (uint64_t)bwn * 32 - rel;   // (a node that starts below the window wraps to a huge rel)
            bool live = ok[u];
            uint32_t mfull = NO_FULL;                                 // the node this step covers whole (its flag is set below, by the whole wave)
            const long long target = (long long)pe - (long long)ps;   // profile.rs:800
            if (live && k == 1) {                                     // :811
                if (target >= 0) {                                    // :821-827
                    if (target && !ABL(2u)) add_bases(bases, wlo, win_n, v[u], (uint32_t)target);
                    if (ps < pe && pe <= nl && !ABL(1u)) {            // :832
                        if (ps == 0 && pe == nl) { if (!ABL(8u)) mfull = v[u]; }
                        else if (ABL(16u)) {}
                        else if (in_win) mark_window(COV_WIN, (uint32_t)rel + ps, (uint32_t)rel + pe);
                        else mark_range(bitmap, COV_WIN, bw0, bwn, bo + ps, bo + pe);
                    }
                }
                live = false;
            }
            if (live && ps > len0[u]) {                               // assert :854 -> whole read contributes nothing
                if (i == 0) atomicAdd(n_abort, 1ull);
                live = false;
            }
            // ---- `seen` before this step = sum of the aligned lengths of steps 0..i-1 of MY read: a plain wave prefix sum (DPP)
            // minus its value at the lane that holds step 0 -- the steps of a read sit in consecutive lanes (mod 2^32 like the adds)
            const uint32_t contrib = live ? (i == 0 ? nl - ps : nl) : 0u;
            const int dst = dist[u];
            const uint32_t pexcl = wave_incl_scan_dpp(contrib) - contrib;
            const uint32_t seen_in_wave = pexcl - __shfl(pexcl, lane - dst);
            // ---- first occurrence of this node in the read (:879): decided at upload time (step codes above)
            const uint32_t id0 = __shfl(id[u], lane - dst);           // id of step 0 when the walk starts in this wave
            uint32_t rl = 0;
            if (live) {
                int jf = -1;                                          // -1: first occurrence; 0: the node of step 0; 1: another earlier step
                if (dupc[u] & STEP_LONG) { if (dupc[u] & 1u) jf = (id[u] == (cross[u] ? node_id[b] : id0)) ? 0 : 1; }
                else if (dupc[u] & STEP_DIST) jf = (int)i - (int)(dupc[u] & STEP_DIST);
                uint32_t aln, sidx;
                if (i == 0) { aln = nl - ps; sidx = ps; }             // :853-856
                else if (i == k - 1) {                                // :857-859
                    uint32_t seen = seen_in_wave;
                    if (cross[u]) seen = long_sum[slot[u]] - ps;      // all steps but the last, from walk_sum_kernel
                    aln = target > (long long)seen ? (uint32_t)(target - (long long)seen) : 0u;   // max(target - seen, 0)
                    sidx = 0;
                } else { aln = nl; sidx = 0; }                        // :860-862
                uint32_t hi = sidx + aln;
                if (hi > nl) hi = nl;                                 // :871
                if (ABL(1u)) {}
                else if (sidx == 0 && hi == nl) { if (nl && !ABL(8u)) mfull = v[u]; }
                else if (ABL(16u)) {}
                else if (in_win) mark_window(COV_WIN, (uint32_t)rel + sidx, (uint32_t)rel + hi);
                else mark_range(bitmap, COV_WIN, bw0, bwn, bo + sidx, bo + hi);
                if (jf < 0) {
                    rl = aln;
                    if (aln && !ABL(2u)) add_bases(bases, wlo, win_n, v[u], aln);     // :881
                } else rl = (jf == 0) ? (len0[u] - ps) : nl;
            }
            mark_full_wave(full, wlo, win_n, mfull);
            if (WITH_TRIO) {                                          // :890-907
                uint32_t rl1 = wave_shr1(rl), rl2 = wave_shr1(wave_shr1(rl));
                if (live && i >= 2 && !ABL(4u)) {
                    if (lane < 1) rl1 = rl_from_memory(i - 1, b, node_id, step_dup, sr[u].y, node_rec, len0[u], ps);
                    if (lane < 2) rl2 = rl_from_memory(i - 2, b, node_id, step_dup, sr[u].y, node_rec, len0[u], ps);
                    long long row = -1;                                      // a row IS its lookup entry: the index of the entry that matches
                    if (th[u].y) {
                        if (e0[u].x == tlo[u] && e0[u].y == thi[u]) row = (long long)th[u].x;
                        else if (th[u].y > 1 && e1[u].x == tlo[u] && e1[u].y == thi[u]) row = (long long)th[u].x + 1;
                        else
                            for (uint32_t j = 2; j < th[u].y; ++j) {
                                const uint2 e = trio_ent[th[u].x + j];
                                if (e.x == tlo[u] && e.y == thi[u]) { row = (long long)th[u].x + j; break; }
                            }
                    }
                    if (row >= 0) {
                        const unsigned long long sum = (unsigned long long)rl2 + rl1 + rl;
                        if (sum) atomicAdd(&trio_bases[row], sum);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (win_n) {
        for (int i = threadIdx.x; i < COV_WIN; i += COV_BLOCK) {
            const uint32_t c = S_WIN(i);
            if (c) atomicAdd(&bases[wlo + i], (unsigned long long)c);
            // full-node flags: the window starts at a multiple of 64 nodes, so a wave's ballot is two whole words of the flag vector
            const unsigned long long fb = __ballot(S_FULL(COV_WIN, i) != 0);
            if (fb && (lane & 31) == 0) {
                const uint32_t m = (uint32_t)(fb >> (lane & 32));
                if (m) atomicOr(&full[(wlo + i) >> 5], m);
            }
        }
    }
    for (uint32_t i = threadIdx.x; i < bwn; i += COV_BLOCK) {
        const uint32_t m = S_BM(COV_WIN, i);
        if (m) atomicOr(&bitmap[bw0 + i], m);     // nothing waits for these (a probe first would be a dependent round trip per word)
    }
}

// Walks of more than 64 steps: the last step needs `seen` = everything aligned before it (:857-859), which lives in
// other waves.  One cheap pass over the steps of long walks adds the node lengths of all steps but the last into
// long_sum[slot] (one atomic per wave and walk); launched only when the upload saw such walks.
template <int WS_U, bool EAGER = false>   // EAGER (most walks are long): the node ids and group slots are requested WITH the step codes, not behind the test for a long step
__global__ void __launch_bounds__(256) walk_sum_kernel(uint64_t T, const uint32_t *__restrict__ group_slot, const uint8_t *__restrict__ step_dup,
                                                       const uint4 *__restrict__ read_rec, const uint2 *__restrict__ slot_rec,
                                                       const uint32_t *__restrict__ node_id, const uint32_t *__restrict__ node_len,
                                                       uint32_t *__restrict__ long_sum, uint32_t *__restrict__ long_len0) {
    // WS_U groups of 64 steps per wave and round, the loads of a level issued for all of them before the first is used (the
    // pass is a chain of three dependent levels: {code, slot, id} -> {slot record, read record} -> node length, a 4-byte
    // gather from the plain length array, not the 16-byte node record)
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * 256 * WS_U;
    for (uint64_t base = ((uint64_t)blockIdx.x * 256 + (threadIdx.x - lane)) * WS_U; base < T; base += stride) {
        uint32_t code[WS_U], slot[WS_U], id[WS_U], nl[WS_U], gs_e[WS_U], id_e[WS_U];
        uint64_t t[WS_U];
        bool any = false;
#pragma unroll
        for (int u = 0; u < WS_U; ++u) {
            t[u] = base + (uint64_t)u * 64 + lane;
            code[u] = t[u] < T ? step_dup[t[u]] : STEP_PAD;
            if constexpr (EAGER) { gs_e[u] = t[u] < T ? group_slot[t[u] >> 6] : NO_SLOT; id_e[u] = t[u] < T ? node_id[t[u]] : 0u; }
        }
#pragma unroll
        for (int u = 0; u < WS_U; ++u) any = any || (code[u] != STEP_PAD && (code[u] & STEP_LONG));
        if (!__any(any)) continue;
#pragma unroll
        for (int u = 0; u < WS_U; ++u) {
            slot[u] = NO_SLOT; id[u] = 0;
            uint32_t gs;
            if constexpr (EAGER) gs = gs_e[u]; else gs = t[u] < T ? group_slot[t[u] >> 6] : NO_SLOT;
            const uint32_t sl = slot_in_group(gs, code[u], lane);
            if (code[u] != STEP_PAD && (code[u] & STEP_LONG)) { slot[u] = sl; if constexpr (EAGER) id[u] = id_e[u]; else id[u] = node_id[t[u]]; }
        }
        uint2 sr[WS_U];
        uint4 rr[WS_U];
#pragma unroll
        for (int u = 0; u < WS_U; ++u) {
            sr[u] = make_uint2(0xFFFFFFFFu, 0u); rr[u] = make_uint4(0u, 0u, 0u, 0u);
            if (slot[u] != NO_SLOT) { sr[u] = slot_rec[slot[u]]; rr[u] = read_rec[slot[u]]; }
        }
#pragma unroll
        for (int u = 0; u < WS_U; ++u) {
            nl[u] = 0;
            const bool counted = slot[u] != NO_SLOT && (int)sr[u].x >= 0 && (uint32_t)(t[u] - rr[u].x) + 1 < rr[u].y;   // not the last step
            if (counted) nl[u] = node_len[id[u] + sr[u].y];
        }
#pragma unroll
        for (int u = 0; u < WS_U; ++u) {
            if (!__any(slot[u] != NO_SLOT)) continue;
            if (slot[u] != NO_SLOT && (int)sr[u].x >= 0 && (uint32_t)(t[u] - rr[u].x) + 1 < rr[u].y && t[u] == rr[u].x)
                long_len0[slot[u]] = nl[u];                          // length of the walk's first node, for the lanes of later waves
            // segmented sum over runs of equal slot (a walk's steps are contiguous), one atomic per run
            const uint32_t prev = __shfl_up(slot[u], 1);
            const bool head = lane == 0 || prev != slot[u];
            const unsigned long long heads = __ballot(head);
            const int start = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));   // lane of my run's head
            // (one DPP prefix sum of the wave minus its value in front of the run's head -- 64 lengths stay far below 2^32 --; a segmented shuffle scan of six
            // bpermutes stood here until round 6)
            const uint32_t sc = wave_incl_scan_dpp(nl[u]);
            const uint32_t front = __shfl(sc, start > 0 ? start - 1 : 0);
            const uint32_t incl = sc - (start > 0 ? front : 0u);
            const unsigned long long after = heads & ~((2ull << lane) - 1ull);
            const bool tail = after ? (lane + 1 == __builtin_ctzll(after)) : lane == 63;
            if (tail && slot[u] != NO_SLOT && incl) atomicAdd(&long_sum[slot[u]], incl);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Resident layout of the packed reads: grouped by the locus of their first node and padded so that
// a walk of <= 64 steps never straddles a 64-step boundary.  Key = first node id >> shift (ids are
// globally ordered by species and position, sort_range.rs:25-33).  Counting sort of the reads into
// slots (histogram -> scan -> scatter), then one thread per bucket lays its walks out (start moved to
// the next multiple of 64 when the walk would straddle one; bucket sizes rounded up to 64), a scan of
// the bucket sizes, and the fill.  Done once per upload: it depends on the reads only, not on the
// binning.  Slot order inside a bucket is arbitrary; every output of the path is an order-independent
// integer sum, so results stay bit-exact.  Reads with an empty walk own no slot (profile.rs:794-796).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) group_count_kernel(uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
                                                          int shift, uint32_t *__restrict__ cnt_r) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256) {
        const uint32_t b = step_off[r], k = step_off[r + 1] - b;
        if (k) atomicAdd(&cnt_r[node_id[b] >> shift], 1u);
    }
}
__global__ void __launch_bounds__(256) group_slot_kernel(uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
                                                         const uint32_t *__restrict__ pstart, const uint32_t *__restrict__ pend, const uint32_t *__restrict__ qlen,
                                                         const uint8_t *__restrict__ mapq, int shift, const uint32_t *__restrict__ base_r, uint32_t *__restrict__ cur_r,
                                                         uint32_t *__restrict__ slot_of, uint4 *__restrict__ g_read_rec, uint2 *__restrict__ g_qm,
                                                         uint32_t *__restrict__ n_long) {
    // The per-read columns are read HERE, in file order (coalesced), and leave as the slot's two records -- {first step of the walk in
    // the source columns (group_fill_kernel replaces it by the walk's place in the grouped stream), #steps, pstart, pend} and {read
    // length, MAPQ}: the fill pass then reads one coalesced record per slot.  (Round 3 kept {#steps, read} per slot and let the fill
    // pass gather six columns at random: 46 GB fetched for 4 GB of payload at 1e8 reads.)
    uint32_t mine = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256) {
        const uint32_t b = step_off[r], k = step_off[r + 1] - b;
        uint32_t slot = NO_SLOT;
        if (k) {
            const uint32_t key = node_id[b] >> shift;
            slot = base_r[key] + atomicAdd(&cur_r[key], 1u);
            g_read_rec[slot] = make_uint4(b, k, pstart[r], pend[r]);
            g_qm[slot] = make_uint2(qlen[r], (uint32_t)mapq[r]);    // slot-order copies for the binning pass
            mine += k > 64 ? 1u : 0u;
        }
        slot_of[r] = slot;
    }
    if (__any(mine != 0)) {
        mine = wave_reduce(mine, [](uint32_t x, uint32_t y) { return x + y; });
        if ((threadIdx.x & 63) == 0) atomicAdd(n_long, mine);
    }
}
// One thread lays out a UNIT of 2^g consecutive buckets; only units are rounded up to 64 steps.  (Rounding every 32-node
// bucket cost 32 pad steps per bucket on average: with ten reads per bucket -- 1e7 reads over 3.2e7 nodes -- the padded
// stream was 1.45 x the walk steps, and the coverage kernel spends a lane on every pad.)
// A thread lays out ONE unit (its reads one after the other: a walk of <= 64 steps never straddles a 64-step border).  A workgroup of 64 threads takes 64
// consecutive units -- their reads are one stretch of slots --, loads the reads' step counts into LDS coalesced, lets every thread walk its unit there,
// and writes the places back coalesced.  (The first version had every thread read its reads' 16-byte records from memory, far from its neighbours':
// 10.7 GB of sector traffic for 1.6 GB of records at 1e8 reads, 4.8 ms.)  A stretch of more reads than the LDS holds takes the plain loop.
constexpr uint32_t GL_UNITS = 64, GL_CAP = 24576;
__global__ void __launch_bounds__(64) group_layout_kernel(uint32_t NB, int g, const uint32_t *__restrict__ base_r /*[NB+1]*/,
                                                          const uint4 *__restrict__ g_read_rec, uint32_t *__restrict__ slot_rel,
                                                          uint32_t *__restrict__ size_s) {
    __shared__ uint32_t s_k[GL_CAP];
    const uint32_t NU = (NB + (1u << g) - 1) >> g;
    const uint32_t u0 = blockIdx.x * GL_UNITS, key = u0 + threadIdx.x;
    const uint32_t kb = min(NB, u0 << g), ke = min(NB, (u0 + GL_UNITS) << g);
    const uint32_t s_begin = base_r[kb], n_wg = base_r[ke] - s_begin;                    // (workgroup-uniform)
    const bool staged = n_wg <= GL_CAP;
    if (staged) {
        for (uint32_t i = threadIdx.x; i < n_wg; i += GL_UNITS) s_k[i] = g_read_rec[s_begin + i].y;
        __syncthreads();
    }
    if (key < NU) {
        const uint32_t k0 = key << g, k1 = min(NB, (key + 1) << g);
        uint32_t pos = 0;
        for (uint32_t s = base_r[k0], e = base_r[k1]; s < e; ++s) {
            const uint32_t k = staged ? s_k[s - s_begin] : g_read_rec[s].y;
            if (k <= 64 && (pos & 63) + k > 64) pos = (pos + 63) & ~63u;
            if (staged) s_k[s - s_begin] = pos; else slot_rel[s] = pos;
            pos += k;
        }
        size_s[key] = (pos + 63) & ~63u;
    }
    if (staged) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_wg; i += GL_UNITS) slot_rel[s_begin + i] = s_k[i];
    }
}
// In SLOT order, one WAVE per 64 consecutive slots: the lanes first file their slot's records (coalesced), then hand the steps
// of the 64 walks out flat over the wave -- lane = step of the output stream, which the slots follow in order, so node ids and
// step codes are written as dense runs and the short source walks are gathered.  (Thread per read in file order scattered
// single dwords and bytes over the whole stream: 72 GB written for 4 GB of payload at 1e8 reads, 46 ms; thread per slot with
// a private loop over its steps still re-read every walk k times for the first-occurrence codes: 41 ms.)  The code of a step
// -- distance back to the first occurrence of its node in the walk -- comes from the lanes below (and, where a walk began in
// the round before, from that round's ids).
__global__ void __launch_bounds__(256) group_fill_kernel(uint32_t n_slots, const uint32_t *__restrict__ node_id, int shift,
                                                         const uint32_t *__restrict__ base_s, const uint32_t *__restrict__ slot_rel, uint4 *__restrict__ g_read_rec,
                                                         uint32_t *__restrict__ g_node_id, uint32_t *__restrict__ g_group_slot,
                                                         uint8_t *__restrict__ g_step_dup) {
    __shared__ uint32_t s_excl[4][65], s_b[4][64], s_sb[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_waves = (n_slots + 63) / 64;
    for (uint32_t w = blockIdx.x * 4 + wave; w < n_waves; w += gridDim.x * 4) {
        const uint32_t slot = w * 64 + lane;
        uint32_t b = 0, k = 0, sb = 0;
        if (slot < n_slots) {
            uint4 rec = g_read_rec[slot];                            // {first step in the source columns, #steps, pstart, pend}: group_slot_kernel
            b = rec.x; k = rec.y;
            sb = base_s[node_id[b] >> shift] + slot_rel[slot];
            rec.x = sb;
            g_read_rec[slot] = rec;
            if (k > 64) k = 0;                                       // laid out by group_fill_long_kernel, one workgroup per walk
            else if ((sb & 63u) == 0u) g_group_slot[sb >> 6] = slot; // a walk of <= 64 steps lies inside one 64-step group
        }
        const uint32_t incl = wave_incl_scan_dpp(k);
        const uint32_t total = __shfl(incl, 63);
        s_excl[wave][lane] = incl - k; s_b[wave][lane] = b; s_sb[wave][lane] = sb;
        if (lane == 0) s_excl[wave][64] = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        uint32_t prev_id = 0;
        for (uint32_t f0 = 0; f0 < total; f0 += 64) {
            const uint32_t f = f0 + (uint32_t)lane;
            const bool on = f < total;
            uint32_t o = 0;                                          // owner: the last slot whose first flat step is <= f (walks of 0 steps own none)
            if (on) {
                uint32_t lo = 0, hi = 63;
                while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (s_excl[wave][mid] <= f) lo = mid; else hi = mid - 1; }
                o = lo;
            }
            const uint32_t i = on ? f - s_excl[wave][o] : 0u;
            const uint32_t id = on ? node_id[s_b[wave][o] + i] : 0u;
            // first occurrence of my node among the i earlier steps of my walk: they sit in the lanes below, or in the round before
            uint32_t dup = 0;
            const uint32_t imax = wave_reduce(i, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            for (uint32_t d = 1; d <= imax; ++d) {
                const uint32_t cur = __shfl(id, (lane - (int)d) & 63), old = __shfl(prev_id, (lane - (int)d) & 63);
                const uint32_t other = (int)d <= lane ? cur : old;
                if (d <= i && other == id) dup = d;                   // the largest such distance = the first occurrence
            }
            if (on) {
                const uint32_t dst = s_sb[wave][o] + i;
                g_node_id[dst] = id;
                g_step_dup[dst] = (uint8_t)(dup | (i == 0 ? STEP_START : 0u));
            }
            prev_id = id;
        }
        __builtin_amdgcn_wave_barrier();                             // the LDS rows are reused by this wave's next 64 slots
    }
}

// Walks of more than 64 steps: one workgroup copies the walk (coalesced) and decides for every step whether its node
// occurred earlier in the walk -- an LDS hash of (node id -> smallest position) for walks of up to LONG_HASH/2 steps,
// a plain scan of the earlier steps above that.
constexpr uint32_t LONG_HASH = 8192;
__global__ void __launch_bounds__(256) group_fill_long_kernel(uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
                                                              int shift, const uint32_t *__restrict__ base_s, const uint32_t *__restrict__ slot_of,
                                                              const uint32_t *__restrict__ slot_rel, uint32_t *__restrict__ g_node_id,
                                                              uint32_t *__restrict__ g_group_slot, uint8_t *__restrict__ g_step_dup) {
    __shared__ uint32_t h_key[LONG_HASH], h_pos[LONG_HASH];
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    for (uint64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const uint32_t b = step_off[r], k = step_off[r + 1] - b;
        if (k <= 64) continue;
        const uint32_t slot = slot_of[r];
        const uint32_t sb = base_s[node_id[b] >> shift] + slot_rel[slot];
        for (uint32_t i = threadIdx.x; i < k; i += 256) {
            g_node_id[sb + i] = node_id[b + i];
            if (((sb + i) & 63u) == 0u) g_group_slot[(sb + i) >> 6] = slot;   // every group this walk's steps begin
        }
        if (k <= LONG_HASH / 2) {
            for (uint32_t i = threadIdx.x; i < LONG_HASH; i += 256) { h_key[i] = EMPTY; h_pos[i] = EMPTY; }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < k; i += 256) {
                const uint32_t id = node_id[b + i];
                uint32_t h = (id * 2654435761u) >> 19;             // 13 bits
                for (;;) {
                    const uint32_t old = atomicCAS(&h_key[h], EMPTY, id);
                    if (old == EMPTY || old == id) { atomicMin(&h_pos[h], i); break; }
                    h = (h + 1) & (LONG_HASH - 1);
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < k; i += 256) {
                const uint32_t id = node_id[b + i];
                uint32_t h = (id * 2654435761u) >> 19;
                while (h_key[h] != id) h = (h + 1) & (LONG_HASH - 1);
                g_step_dup[sb + i] = (uint8_t)(STEP_LONG | (h_pos[h] < i ? 1u : 0u) | (i == 0 ? STEP_START : 0u));
            }
            __syncthreads();
        } else {
            for (uint32_t i = threadIdx.x; i < k; i += 256) {
                const uint32_t id = node_id[b + i];
                uint32_t dup = 0;
                for (uint32_t j = 0; j < i; ++j) if (node_id[b + j] == id) { dup = 1; break; }
                g_step_dup[sb + i] = (uint8_t)(STEP_LONG | dup | (i == 0 ? STEP_START : 0u));
            }
        }
    }
}

// first group of every node block: the smallest group whose first step is the first step of a read that starts in the block (a group
// that continues a longer walk, or holds only pads, starts no read)
__global__ void __launch_bounds__(256) group_block_kernel(uint32_t n_groups, const uint32_t *__restrict__ group_slot, const uint32_t *__restrict__ g_node_id,
                                                          const uint8_t *__restrict__ step_code, int bshift, uint32_t *__restrict__ first_g) {
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < n_groups; g += gridDim.x * 256) {
        if (group_slot[g] == NO_SLOT) continue;
        const uint32_t code = step_code[(uint64_t)g * 64];
        if (code == STEP_PAD || !(code & STEP_START)) continue;
        atomicMin(&first_g[g_node_id[(uint64_t)g * 64] >> bshift], g);
    }
}

int build_step_read(Ctx *ctx, Reads *rd, uint32_t max_node_id) {
    static std::atomic<uint64_t> next_layout{1};
    rd->layout_id = next_layout.fetch_add(1);       // (what a db's list of work items is made for)
    rd->T_pad = 0;
    rd->n_long = 0;
    rd->n_slots = 0;
    rd->n_items = 0;
    rd->g_flags_valid = false;
    rd->species_valid = false;
    PTX_HIP(ctx, rd->d_slot_of.alloc(rd->R ? rd->R : 1));
    PTX_HIP(ctx, rd->d_g_slot_rec.alloc(rd->R ? rd->R : 1));
    PTX_HIP(ctx, rd->d_g_qm.alloc(rd->R ? rd->R : 1));
    if (rd->R == 0) return 0;
    if (rd->T == 0) {
        PTX_HIP(ctx, hipMemsetAsync(rd->d_slot_of.p, 0xFF, rd->R * sizeof(uint32_t), ctx->stream));
        return 0;
    }
    int shift = 5;
    // buckets of 32 node ids up to 5e8 ids (round 4; 2^20 buckets before: 512-id buckets at 3e8 ids, whose reads are in no particular order --
    // the coverage pass gathers node records along the stream, and neighbours in the stream should be neighbours in the graph)
    int bucket_cap_bits = 24;
    if (ctx->cfg.group_bucket_bits) bucket_cap_bits = std::max(10, std::min(26, ctx->cfg.group_bucket_bits));
    while (((uint64_t)max_node_id >> shift) + 1 > (1ull << bucket_cap_bits)) ++shift;
    const uint32_t NB = (uint32_t)(max_node_id >> shift) + 1;
    DevBuf<uint32_t> cnt, scan_tmp, slot_rel;
    PTX_HIP(ctx, cnt.alloc(4ull * (NB + 1) + 8));
    uint32_t *cnt_r = cnt.p, *base_r = cnt_r + (NB + 1), *size_s = base_r + (NB + 1), *base_s = size_s + (NB + 1);
    PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(NB + 1)));
    PTX_HIP(ctx, slot_rel.alloc(rd->R));
    PTX_HIP(ctx, rd->d_g_read_rec.alloc(rd->R));
    PTX_TRY(zero_fill(ctx, cnt_r, (NB + 1) * sizeof(uint32_t)));
    int gridR = grid_for(rd->R, 256, ctx->n_cu * 8);
    hipLaunchKernelGGL(group_count_kernel, dim3(gridR), dim3(256), 0, ctx->stream, rd->R, rd->d_step_off.p, rd->d_node_id.p, shift, cnt_r);
    PTX_TRY(exclusive_scan_u32(ctx, cnt_r, base_r, NB + 1, scan_tmp.p, nullptr));
    PTX_TRY(zero_fill(ctx, cnt_r, (NB + 1) * sizeof(uint32_t)));   // reused as cursors
    uint32_t *d_total = (uint32_t *)ctx->d_scalars.p, *d_n_long = d_total + 1;
    PTX_HIP(ctx, hipMemsetAsync(d_n_long, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(group_slot_kernel, dim3(gridR), dim3(256), 0, ctx->stream, rd->R, rd->d_step_off.p, rd->d_node_id.p, rd->d_pstart.p, rd->d_pend.p,
                       rd->d_qlen.p, rd->d_mapq.p, shift, base_r, cnt_r, rd->d_slot_of.p, rd->d_g_read_rec.p, rd->d_g_qm.p, d_n_long);
    // layout units: 2^g buckets each, about 2048 walk steps per unit (the rounding of a unit to 64 steps then costs ~1.5 %)
    int g = 0;
    while (g < 12 && ((double)rd->T / (double)NB) * (double)(1u << g) < 2048.0) ++g;
    const uint32_t NU = (NB + (1u << g) - 1) >> g;
    hipLaunchKernelGGL(group_layout_kernel, dim3((NU + GL_UNITS - 1) / GL_UNITS), dim3(GL_UNITS), 0, ctx->stream, NB, g, base_r, rd->d_g_read_rec.p, slot_rel.p, size_s);
    PTX_TRY(exclusive_scan_u32(ctx, size_s, base_s, NU, scan_tmp.p, d_total));
    const int ushift = shift + g;   // unit of a read = its first node id >> ushift
    uint32_t h_tot[2] = {0, 0}, h_slots = 0;
    PTX_TRY(download(ctx, h_tot, d_total, 2));
    PTX_TRY(download(ctx, &h_slots, base_r + NB, 1));   // reads that own a slot (non-empty walk)
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t h_total = h_tot[0];
    rd->n_long = h_tot[1];
    rd->n_slots = h_slots;
    if ((uint64_t)h_total < rd->T) return fail(ctx, PANTAX_HIP_E_LIMIT, "reads_upload: padded step stream exceeds 32-bit positions");
    rd->T_pad = h_total;
    PTX_HIP(ctx, rd->d_g_node_id.alloc(rd->T_pad)); PTX_HIP(ctx, rd->d_g_group_slot.alloc(rd->T_pad / 64 + 1)); PTX_HIP(ctx, rd->d_g_step_dup.alloc(rd->T_pad));
    PTX_TRY(byte_fill(ctx, rd->d_g_node_id.p, 0, rd->T_pad * sizeof(uint32_t)));
    PTX_TRY(byte_fill(ctx, rd->d_g_group_slot.p, 0xFF, (rd->T_pad / 64 + 1) * sizeof(uint32_t)));
    PTX_TRY(byte_fill(ctx, rd->d_g_step_dup.p, 0xFF, rd->T_pad));                                            // STEP_PAD
    if (rd->n_slots)
        hipLaunchKernelGGL(group_fill_kernel, dim3(grid_for(rd->n_slots, 256, ctx->n_cu * 16)), dim3(256), 0, ctx->stream, rd->n_slots, rd->d_node_id.p, ushift,
                           base_s, slot_rel.p, rd->d_g_read_rec.p, rd->d_g_node_id.p, rd->d_g_group_slot.p, rd->d_g_step_dup.p);
    if (rd->n_long) {
        const uint32_t gridL = (uint32_t)std::min<uint64_t>(rd->R, (uint64_t)ctx->n_cu * 64);
        hipLaunchKernelGGL(group_fill_long_kernel, dim3(gridL), dim3(256), 0, ctx->stream, rd->R, rd->d_step_off.p, rd->d_node_id.p, ushift, base_s,
                           rd->d_slot_of.p, slot_rel.p, rd->d_g_node_id.p, rd->d_g_group_slot.p, rd->d_g_step_dup.p);
        PTX_HIP(ctx, rd->d_long_sum.alloc(rd->R));
        PTX_HIP(ctx, rd->d_long_len0.alloc(rd->R));
    }
    PTX_HIP(ctx, hipGetLastError());
    // work items of the short-read coverage kernel: the groups cut at the borders of 2048-id node blocks (the stream is in the order of
    // the reads' first nodes, bucket by bucket), a block's groups cut into items of COV_ITEM_GROUPS
    {
        const uint32_t n_groups = (uint32_t)(rd->T_pad / 64);
        const int bshift = std::max(COV_BLK_SHIFT, shift);
        const uint32_t NBLK = (uint32_t)(max_node_id >> bshift) + 1;
        DevBuf<uint32_t> first_g;
        PTX_HIP(ctx, first_g.alloc(NBLK + 1));
        PTX_HIP(ctx, hipMemsetAsync(first_g.p, 0xFF, ((size_t)NBLK + 1) * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(group_block_kernel, dim3(grid_for(n_groups, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, n_groups, rd->d_g_group_slot.p, rd->d_g_node_id.p,
                           rd->d_g_step_dup.p, bshift, first_g.p);
        std::vector<uint32_t> fg((size_t)NBLK + 1);
        PTX_TRY(download(ctx, fg.data(), first_g.p, (size_t)NBLK + 1));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        fg[NBLK] = n_groups;
        bool monotone = true;
        for (uint32_t b = NBLK; b-- > 0;) { if (fg[b] == 0xFFFFFFFFu) fg[b] = fg[b + 1]; else if (fg[b] > fg[b + 1]) monotone = false; }
        fg[0] = 0;                                                  // groups in front of the first live one (pads) belong to the first block
        std::vector<uint2> items;
        rd->h_item_block.clear();
        const uint32_t cap = ctx->cfg.cov_item_groups > 0 ? (uint32_t)ctx->cfg.cov_item_groups : COV_ITEM_GROUPS;
        if (monotone)
            for (uint32_t b = 0; b < NBLK; ++b)
            {   // a block's groups in EQUAL items of at most `cap` groups (81 groups: 41 + 40, not 64 + 17 -- a workgroup zeroes and flushes its LDS windows once per item)
                const uint32_t n = fg[b + 1] - fg[b];
                if (!n) continue;
                const uint32_t k = (n + cap - 1) / cap, per = (n + k - 1) / k;
                for (uint32_t g = fg[b]; g < fg[b + 1]; g += per) { items.push_back(make_uint2(g, std::min(fg[b + 1], g + per))); rd->h_item_block.push_back(b); }
            }
        else   // cannot happen with the counting sort above; never silent: plain cuts of the stream
            for (uint32_t g = 0; g < n_groups; g += COV_ITEM_GROUPS) items.push_back(make_uint2(g, std::min(n_groups, g + COV_ITEM_GROUPS)));
        rd->n_items = (uint32_t)items.size();
        rd->item_blk_shift = monotone ? bshift : 0;
        PTX_TRY(upload(ctx, rd->d_g_items, items.data(), items.size()));
    }
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // temporaries are released on return
    return 0;
}

int reads_group(Ctx *ctx, Reads *rd) {
    if (rd->grouped) return 0;
    PTX_TRY(build_step_read(ctx, rd, rd->max_node_id));
    rd->grouped = true;
    rd->binned = false;          // the species of a read now live in its slot record: the next binning pass writes them
    return 0;
}

// node_base_cov[v] = number of covered bases (profile.rs:844/874, :1018-1023)
// a node some step covered whole carries a flag instead of marked bits (coverage_step_kernel): its count is its length
__global__ void __launch_bounds__(256) popcount_kernel(uint64_t V, const uint64_t *__restrict__ bit_off, const uint32_t *__restrict__ full,
                                                       const uint32_t *__restrict__ bitmap, uint32_t *__restrict__ cov) {
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (uint64_t)gridDim.x * 256) {
        uint64_t g0 = bit_off[v], g1 = bit_off[v + 1];
        uint32_t c = 0;
        if ((full[v >> 5] >> (uint32_t)(v & 31)) & 1u) c = (uint32_t)(g1 - g0);
        else if (g1 > g0) {
            uint64_t w0 = g0 >> 5, w1 = (g1 - 1) >> 5;
            uint32_t m0 = 0xFFFFFFFFu << (g0 & 31);
            uint32_t m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
            if (w0 == w1) c = __popc(bitmap[w0] & m0 & m1);
            else {
                c = __popc(bitmap[w0] & m0);
                for (uint64_t w = w0 + 1; w < w1; ++w) c += __popc(bitmap[w]);
                c += __popc(bitmap[w1] & m1);
            }
        }
        cov[v] = c;
    }
}

// The same counts for graphs of LONG nodes (a single-genome species is a chain of 1024-bp chunks, build_eq1.rs:26-36: 32 words of the bit vector per
// node): the per-thread loop above walks 64 different cache lines per iteration (11 ms at the reference-DB shape).  Here a wave takes 64 consecutive nodes,
// reads the words of the whole stretch coalesced, keeps the number of set bits in front of every word in LDS (a DPP prefix sum per 64 words) and takes a
// node's count as the difference of that prefix at its two ends -- node_cov_stats_kernel<.., LONGN>'s scheme (stage_lad.hip) for the stage call.
constexpr uint32_t PCL_WORDS = 2304;   // words of one stretch the prefix holds; a longer stretch takes the per-lane loop
__global__ void __launch_bounds__(256) popcount_long_kernel(uint64_t V, const uint64_t *__restrict__ bit_off, const uint32_t *__restrict__ full,
                                                            const uint32_t *__restrict__ bitmap, uint32_t *__restrict__ cov) {
    __shared__ __attribute__((aligned(16))) uint32_t s_prefix[4 * PCL_WORDS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *pw = s_prefix + wave * PCL_WORDS;
    const uint64_t n_str = (V + 63) / 64;
    for (uint64_t st = (uint64_t)blockIdx.x * 4 + wave; st < n_str; st += (uint64_t)gridDim.x * 4) {
        const uint64_t v = st * 64 + lane;
        const bool in = v < V;
        const uint64_t g0 = in ? bit_off[v] : 0ull, g1 = in ? bit_off[v + 1] : 0ull;
        const uint64_t b0 = __shfl(g0, 0);                                             // (lane 0 is always inside)
        const uint32_t last = (uint32_t)min((uint64_t)63, V - 1 - st * 64);
        const uint64_t b1 = __shfl(g1, (int)last);
        // (from the 16-byte boundary at or below the stretch's first word: four words per lane and load; what the last load reads beyond the stretch is
        // inside the arena -- the flags follow the bit vector -- and never looked up)
        const uint64_t wa = (b0 >> 5) & ~3ull, nw = b1 > b0 ? ((b1 - 1) >> 5) - wa + 1 : 0;
        const bool coop = nw <= (uint64_t)PCL_WORDS && __builtin_amdgcn_ballot_w64(in && g1 - g0 > 64ull) != 0ull;   // (wave-uniform)
        if (coop) {
            uint32_t carry = 0;
            for (uint32_t k = 0; k < (uint32_t)nw; k += 256) {
                const uint32_t i = k + 4u * lane;
                const uint4 x = i < (uint32_t)nw ? *reinterpret_cast<const uint4 *>(bitmap + wa + i) : make_uint4(0u, 0u, 0u, 0u);
                const uint32_t p0 = (uint32_t)__popc(x.x), p1 = p0 + (uint32_t)__popc(x.y), p2 = p1 + (uint32_t)__popc(x.z), p3 = p2 + (uint32_t)__popc(x.w);
                const uint32_t incl = wave_incl_scan_dpp(p3);
                const uint32_t base = carry + incl - p3;                                 // set bits in front of this lane's four words
                if (i < (uint32_t)nw) *reinterpret_cast<uint4 *>(pw + i) = make_uint4(base, base + p0, base + p1, base + p2);
                carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
        uint32_t c = 0;
        if (in && g1 > g0) {
            const uint64_t w0 = g0 >> 5, w1 = (g1 - 1) >> 5;
            const uint32_t m0 = 0xFFFFFFFFu << (g0 & 31), m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
            if ((full[v >> 5] >> (uint32_t)(v & 31)) & 1u) c = (uint32_t)(g1 - g0);
            else if (coop) c = (pw[w1 - wa] + (uint32_t)__popc(bitmap[w1] & m1)) - (pw[w0 - wa] + (uint32_t)__popc(bitmap[w0] & ~m0));
            else if (w0 == w1) c = __popc(bitmap[w0] & m0 & m1);
            else {
                c = __popc(bitmap[w0] & m0);
                for (uint64_t w = w0 + 1; w < w1; ++w) c += __popc(bitmap[w]);
                c += __popc(bitmap[w1] & m1);
            }
        }
        if (coop) __builtin_amdgcn_wave_barrier();                                      // (the prefix is rewritten by the next stretch)
        if (in) cov[v] = c;
    }
}

__global__ void __launch_bounds__(256) count_nonzero_words_kernel(const uint32_t *__restrict__ p, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) c += p[i] != 0u;
    if (c) atomicAdd(out, c);
}

// The part of the coverage pass that depends on the binning only -- zero-filling the result arena and the walk sums of
// long reads: the resident step issues it while the trio index is still being built on the side stream.
int coverage_prepare(Ctx *ctx, Db *db, Reads *rd, bool with_trio) {
    const uint64_t words = (db->L + 31) / 32 + 1;
    const uint64_t U = with_trio ? db->U : 0;
    // one arena, one memset: [bases V u64][trio_bases U u64][abort u64][bitmap words u32][full-node flags: 1 bit per node, padded by a window]
    const uint64_t fwords = (db->V + 4096 + 63) / 32 + 2;     // padded by the largest LDS window
    const size_t off_trio = db->V * 8, off_abort = off_trio + (U ? U : 1) * 8, off_bm = (off_abort + 8 + 15) & ~(size_t)15 /* (16-byte loads of the bit vector) */, off_full = off_bm + words * 4,
                 total = off_full + fwords * 4;
    PTX_HIP(ctx, db->d_cov_arena.alloc(total));
    uint8_t *base = db->d_cov_arena.p;
    db->d_bases.view(base, db->V);
    db->d_trio_bases.view(base + off_trio, U ? U : 1);
    db->d_abort = reinterpret_cast<unsigned long long *>(base + off_abort);
    db->d_bitmap.view(base + off_bm, words);
    db->d_full.view(base + off_full, fwords);
    PTX_HIP(ctx, db->d_cov.alloc(db->V));
    // the resident step's last readers left the arena zeroed (cov_arena_clean) unless its layout or place changed since: only the abort counter is reset
    const uint64_t sig = (uint64_t)(uintptr_t)base ^ ((uint64_t)total * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)off_bm << 1) ^ ((uint64_t)off_full << 2) ^ (uint64_t)off_trio;
    db->cov_arena_total = total;
    if (db->cov_clean_pending) {   // a fill on the side stream (coverage_arena_clean_async): over before anything here touches the arena
        PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream, db->ev_cov_clean, 0));
        db->cov_clean_pending = false;
    }
    if (db->cov_arena_clean && db->cov_arena_sig == sig) {
        if (ctx->cfg.cov_arena_verify) {
            DevBuf<unsigned long long> d_nz;
            PTX_HIP(ctx, d_nz.alloc(1));
            PTX_HIP(ctx, hipMemsetAsync(d_nz.p, 0, 8, ctx->stream));
            PTX_HIP(ctx, hipMemsetAsync(db->d_abort, 0, 8, ctx->stream));
            hipLaunchKernelGGL(count_nonzero_words_kernel, dim3(grid_for(total / 4, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, reinterpret_cast<const uint32_t *>(base), (uint64_t)(total / 4), d_nz.p);
            unsigned long long nz = 0;
            PTX_TRY(download(ctx, &nz, d_nz.p, 1));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (nz) return fail(ctx, PANTAX_HIP_E_STATE, "coverage_prepare: %llu non-zero words in an arena its last readers should have left clean", nz);
        }
        PTX_HIP(ctx, hipMemsetAsync(db->d_abort, 0, 8, ctx->stream));
    } else {
        KTimer t(ctx, "zero_fill_kernel");   // (the arena's fill is a kernel of the step like any other: timed with them)
        PTX_TRY(zero_fill(ctx, base, total));
    }
    db->cov_arena_clean = false;   // the coming pass writes it
    db->cov_arena_sig = sig;
    if (rd->R && rd->T_pad && rd->n_long && rd->long_sums_db != 0 && rd->long_sums_db == db->uid) {
        // the binning pass of these reads against THIS db took the walk sums on its way (bin_slots_kernel, round 6)
    } else if (rd->R && rd->T_pad && rd->n_long) {
        PTX_HIP(ctx, hipMemsetAsync(rd->d_long_sum.p, 0, rd->R * sizeof(uint32_t), ctx->stream));
        KTimer t(ctx, "walk_sum_kernel");
        const bool eager = (uint64_t)rd->n_long * 2 > rd->n_slots;      // most walks are long: one dependent level less (0.363 -> 0.333 ms at the cfg5 share)
        hipLaunchKernelGGL((eager ? walk_sum_kernel<4, true> : walk_sum_kernel<4, false>), dim3(grid_for(rd->T_pad / 4 + 1, 256, ctx->n_cu * 16)), dim3(256), 0, ctx->stream, rd->T_pad, rd->d_g_group_slot.p,
                           rd->d_g_step_dup.p, rd->d_g_read_rec.p, rd->d_g_slot_rec.p, rd->d_g_node_id.p, db->d_node_len.p, rd->d_long_sum.p,
                           rd->d_long_len0.p);
    }
    PTX_HIP(ctx, hipGetLastError());
    db->cov_prepared = true;
    return 0;
}

// The resident step's zero fill of the coverage arena, taken off the main stream (round 6): enqueued on the side stream behind everything the main stream
// holds so far (the arena's last readers among it), it runs beside what the main stream gets next -- strain_enqueue calls it in front of the LPs, one
// workgroup per species, which leave most of the memory system idle (behind the node statistics, i.e. beside the row sort's bandwidth-bound passes: no
// gain; in two bursts, one beside the sort's sampling kernels: less gain; behind the LPs: a third of the gain; at the start of the NEXT step beside its binning
// pass, which waits for gathers at 0.3 of the HBM rate: the same 0.3 ms as here -- a streaming fill slows a latency-bound neighbour by what it takes).  The next coverage_prepare waits for
// ev_cov_clean.
int coverage_arena_clean_async(Ctx *ctx, Db *db) {
    if (!db->d_cov_arena.p || !db->cov_arena_total || !ctx->stream2) return 0;
    uint8_t *f_ptr = db->d_cov_arena.p;
    const size_t f_n = db->cov_arena_total;
    if (!db->ev_cov_read) PTX_HIP(ctx, hipEventCreateWithFlags(&db->ev_cov_read, hipEventDisableTiming));
    if (!db->ev_cov_clean) PTX_HIP(ctx, hipEventCreateWithFlags(&db->ev_cov_clean, hipEventDisableTiming));
    PTX_HIP(ctx, hipEventRecord(db->ev_cov_read, ctx->stream));
    PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, db->ev_cov_read, 0));
    hipStream_t main_stream = ctx->stream;
    ctx->stream_main = main_stream; ctx->stream = ctx->stream2;
    int rc;
    rc = zero_fill(ctx, f_ptr, f_n);
    const hipError_t e = hipEventRecord(db->ev_cov_clean, ctx->stream2);
    ctx->stream = main_stream; ctx->stream_main = nullptr;
    if (rc != 0) return rc;
    PTX_HIP(ctx, e);
    db->cov_done = false;          // the arena no longer holds a coverage result
    db->cov_arena_clean = true; db->cov_clean_pending = true;
    return 0;
}

int coverage_launch(Ctx *ctx, Db *db, Reads *rd, const uint8_t *d_active, bool with_trio, bool defer_count) {
    if (!rd->grouped) return fail(ctx, PANTAX_HIP_E_STATE, "node_coverage: these reads are a slice kept as plain columns (to be routed to their owner), not resident reads");
    const bool trace = ctx->cfg.trace && !defer_count;     // (the stage call of the file seam: where its milliseconds go)
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[node_coverage]        %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (!db->cov_prepared) PTX_TRY(coverage_prepare(ctx, db, rd, with_trio));
    lap("arena + zero fill");
    db->cov_prepared = false;
    db->trio_free_valid = false;   // a reader of the unique-trio tables goes onto the stream: the event of an earlier strain step no longer covers them
    unsigned long long *d_abort = db->d_abort;
    if (rd->R && rd->T_pad) {
        const uint32_t xcd_map = (uint32_t)ctx->cfg.cov_xcd;   // 1: every XCD walks one contiguous eighth of the stream (measured slower: 1.42 vs 1.30 ms at cfg3)
        const uint32_t ablate = ctx->cfg.cov_ablate;           // -DCOV_ABLATE builds only
        const bool cov_general = ctx->cfg.cov_general;
        const bool trio = with_trio && db->U;
        const uint8_t *d_act_fast = d_active;
        if (!d_act_fast) {   // the short-read kernel loads the flag unconditionally: all ones when no species is deselected
            if (db->d_ones.n < db->S) { PTX_HIP(ctx, db->d_ones.alloc(db->S)); PTX_HIP(ctx, hipMemsetAsync(db->d_ones.p, 1, db->S, ctx->stream)); }
            d_act_fast = db->d_ones.p;
        }
        // walks of <= 64 steps: the short-read kernel, one wave per 64-step group, PASSES groups per wave and workgroup (the LDS
        // windows are zeroed and flushed once per workgroup).  Skipped when every walk is longer.
        if (rd->n_long < rd->n_slots && rd->n_items && !cov_general) {
            KTimer t(ctx, "coverage_fast_kernel");
            // groups in flight per wave, rounds per workgroup, nodes in the LDS window: 2 x 4 groups (2048 steps); 2 x 8 (4096 steps) on streams of 2^28
            // steps and more, where a workgroup's start-up chain costs more (measured: 0.711 vs 0.768 ms at 8e7 steps, 11.7 vs 9.8 ms at 8e8).  The window:
            // 2304 nodes since round 5 -- an item's reads start inside one block of 2048 ids and the window begins 64..127 nodes in front of it, so 2304
            // hold every short read; the 3072 of rounds 3-4 cost 23.5 KB of LDS per workgroup = SIX waves per SIMD where the registers allow eight (19.7 KB:
            // eight).  The kernel waits for its gathers two thirds of the time: 6.46 -> 5.77 ms at 1e8 reads, 0.708 -> 0.639 at 1e7 (2048 nodes: 7.3 / 0.77,
            // the reads at a block's end fall off the window)
            int fshape = rd->T_pad >= (1ull << 28) ? 2823 : 2423;
            if (ctx->cfg.covf_shape > 0) fshape = ctx->cfg.covf_shape;
            // Only the items whose node block meets the id range of one of the db's species hold reads of its species (round 6: the file seam runs a selection
            // group by group over the same resident reads -- a read that starts outside every range is "U" for this db, and streaming its steps only to
            // find that out cost a group of a quarter of the species 6 ms where the whole selection as one db took 8).  Every read of an item starts inside
            // the item's block.  The groups of the seam are contiguous in the order of the species TABLE (by abundance), not of the ids: the items are
            // picked species by species into a list (a few hundred KB), not as one range.
            uint32_t n_sel = rd->n_items;
            const uint32_t *d_sel = nullptr;
            if (db->item_sel_layout == rd->layout_id && rd->layout_id != 0) {        // the list made for these reads' layout by an earlier pass of this db
                n_sel = db->item_sel_n;
                d_sel = db->item_sel_on ? db->d_item_sel.p : nullptr;
            } else if (rd->h_item_block.size() == rd->n_items && rd->item_blk_shift > 0 && db->S && !db->h_range_start.empty()) {
                std::vector<std::pair<uint32_t, uint32_t>> rg;       // item ranges of the species, then merged
                for (uint32_t s2 = 0; s2 < db->S; ++s2) {
                    const uint32_t b_lo = (uint32_t)std::max<int64_t>(db->h_range_start[s2], 0) >> rd->item_blk_shift;
                    const uint32_t b_hi = (uint32_t)std::min<int64_t>(std::max<int64_t>(db->h_range_end[s2], 0), 0xFFFFFFFFll) >> rd->item_blk_shift;
                    uint32_t i_lo = (uint32_t)(std::lower_bound(rd->h_item_block.begin(), rd->h_item_block.end(), b_lo) - rd->h_item_block.begin());
                    // An item carries the block of the FIRST read of its groups; the last group of the item in front may run on into this block (a group is
                    // 64 steps of consecutive reads, and where reads are sparse a layout unit spans several blocks): its reads of block b_lo are this
                    // species' too.  One item back is enough -- the next group already begins with a read of b_lo and opens an item of that block.
                    if (i_lo > 0) --i_lo;
                    const uint32_t i_hi = (uint32_t)(std::upper_bound(rd->h_item_block.begin(), rd->h_item_block.end(), b_hi) - rd->h_item_block.begin());
                    if (i_hi > i_lo) rg.emplace_back(i_lo, i_hi);
                }
                std::sort(rg.begin(), rg.end());
                std::vector<uint32_t> sel;
                uint32_t done = 0;
                for (const auto &r : rg) for (uint32_t i = std::max(r.first, done); i < r.second; ++i) { sel.push_back(i); done = i + 1; }
                if (sel.size() + sel.size() / 8 < rd->n_items) {     // (worth the indirection)
                    n_sel = (uint32_t)sel.size();
                    if (n_sel) { PTX_TRY(upload(ctx, db->d_item_sel, sel.data(), sel.size())); d_sel = db->d_item_sel.p; }
                }
                db->item_sel_layout = rd->layout_id; db->item_sel_n = n_sel; db->item_sel_on = d_sel != nullptr || n_sel == 0;
            }
            if (trace) std::fprintf(stderr, "[node_coverage]        %u of %u items launched\n", n_sel, rd->n_items);
#define COVF_ARGS rd->d_g_items.p, rd->d_g_group_slot.p, rd->d_g_read_rec.p, rd->d_g_slot_rec.p, rd->d_g_node_id.p, rd->d_g_step_dup.p, d_act_fast, db->d_node_rec.p, \
                  db->d_bit_off.p, db->V, db->d_bases.p, db->d_bitmap.p, db->d_full.p, db->d_trio_ent.p, db->d_trio_bases.p, d_abort, ablate, rd->item_blk_shift, \
                  (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, 0u, 0u, 0u, d_sel
#define COVF_LAUNCH(UU, PP, WW)                                                                                                             \
            {                                                                                                                            \
                const int grid = (int)n_sel;                                                                                             \
                if (grid <= 0) {}                                                                                                        \
                else if (trio) hipLaunchKernelGGL((coverage_fast_kernel<true, UU, PP, WW>), dim3(grid), dim3(COV_BLOCK), cov_lds_bytes(WW), ctx->stream, COVF_ARGS); \
                else hipLaunchKernelGGL((coverage_fast_kernel<false, UU, PP, WW>), dim3(grid), dim3(COV_BLOCK), cov_lds_bytes(WW), ctx->stream, COVF_ARGS);  \
            }
            switch (fshape) {                                              // <U><PASSES><window / 1024>
                case 182: COVF_LAUNCH(1, 8, 2048) break;
                case 242: COVF_LAUNCH(2, 4, 2048) break;
                case 282: COVF_LAUNCH(2, 8, 2048) break;
                case 283: COVF_LAUNCH(2, 8, 3072) break;
                case 284: COVF_LAUNCH(2, 8, 4096) break;
                case 243: COVF_LAUNCH(2, 4, 3072) break;
                case 2823: COVF_LAUNCH(2, 8, 2304) break;                  // <U><PASSES><window / 256 as two digits - 70>: windows between 2048 and 3072 nodes
                case 2825: COVF_LAUNCH(2, 8, 2560) break;
                case 2423: COVF_LAUNCH(2, 4, 2304) break;
                case 2425: COVF_LAUNCH(2, 4, 2560) break;
                case 1823: COVF_LAUNCH(1, 8, 2304) break;
                case 4423: COVF_LAUNCH(4, 4, 2304) break;
                case 442: COVF_LAUNCH(4, 4, 2048) break;
                case 443: COVF_LAUNCH(4, 4, 3072) break;
                default: COVF_LAUNCH(2, 4, 2048) break;
            }
#undef COVF_LAUNCH
#undef COVF_ARGS
        }
        // groups that hold steps of longer walks (HiFi / ONT reads): the general kernel.  U groups of 64 steps in flight per wave,
        // PASSES rounds per workgroup (PANTAX_COV_SHAPE=<U><PASSES> picks another instantiation, for measurements);
        // PANTAX_COV_GENERAL=1 sends every group through it (measurements, and the tests force it).
        const bool long_by_step = ctx->cfg.cov_long == "step";   // round 5's kernel for the groups of longer walks (measurements, and the tests compare the two)
        if ((rd->n_long || cov_general) && !long_by_step) {
            // round 6: the select-only body (coverage_fast_kernel<.., LONG>) over plain cuts of the stream.  covl_shape = <U><groups per workgroup / 8><window /
            // 1024 nodes><nodes in front of the first step / 256>: default 2 groups in flight per wave, 16 groups (1024 steps, ~2 HiFi reads) per workgroup,
            // a 3072-node window that begins 1024 nodes in front of the first live step (reverse-strand walks run down from their first node)
            KTimer t(ctx, "coverage_long_kernel");
            if (!rd->d_long_sum.p) { PTX_HIP(ctx, rd->d_long_sum.alloc(rd->R)); PTX_HIP(ctx, rd->d_long_len0.alloc(rd->R)); }   // (cov_general over short reads: never read)
            const uint32_t only_long = cov_general ? 0u : 1u;
            int shape = ctx->cfg.covl_shape > 0 ? ctx->cfg.covl_shape : 2834;
            // (four digits <U><G><W><B>, or five <U><GG><W><B> for more than 72 groups per workgroup)
            const int su = shape >= 10000 ? shape / 10000 : shape / 1000, sg = shape >= 10000 ? shape / 100 % 100 : shape / 100 % 10, sw = shape / 10 % 10, sb = shape % 10;
            const uint32_t chunk_groups = (uint32_t)std::max(1, sg) * 8u, total_groups = (uint32_t)(rd->T_pad / 64), win_back = (uint32_t)sb * 256u;
            const int grid = (int)((total_groups + chunk_groups - 1) / chunk_groups);
#define COVL_ARGS (const uint2 *)nullptr, rd->d_g_group_slot.p, rd->d_g_read_rec.p, rd->d_g_slot_rec.p, rd->d_g_node_id.p, rd->d_g_step_dup.p, d_act_fast, db->d_node_rec.p, \
                  db->d_bit_off.p, db->V, db->d_bases.p, db->d_bitmap.p, db->d_full.p, db->d_trio_ent.p, db->d_trio_bases.p, d_abort, ablate, 0, \
                  (const uint32_t *)rd->d_long_sum.p, (const uint32_t *)rd->d_long_len0.p, only_long, chunk_groups, total_groups, win_back
#define COVL_LAUNCH(UU, WW)                                                                                                                  \
            {                                                                                                                                \
                if (trio) hipLaunchKernelGGL((coverage_fast_kernel<true, UU, 1, WW, true>), dim3(grid), dim3(COV_BLOCK), cov_lds_bytes(WW), ctx->stream, COVL_ARGS); \
                else hipLaunchKernelGGL((coverage_fast_kernel<false, UU, 1, WW, true>), dim3(grid), dim3(COV_BLOCK), cov_lds_bytes(WW), ctx->stream, COVL_ARGS);  \
            }
            if (grid > 0) switch (su * 10 + sw) {
                case 12: COVL_LAUNCH(1, 2048) break;
                case 13: COVL_LAUNCH(1, 3072) break;
                case 14: COVL_LAUNCH(1, 4096) break;
                case 22: COVL_LAUNCH(2, 2048) break;
                case 24: COVL_LAUNCH(2, 4096) break;
                default: COVL_LAUNCH(2, 3072) break;
            }
#undef COVL_LAUNCH
#undef COVL_ARGS
        }
        if ((rd->n_long || cov_general) && long_by_step) {
            KTimer t(ctx, "coverage_step_kernel");
            const uint32_t only_long = cov_general ? 0u : 1u;
            int shape = rd->T_pad >= (1ull << 25) ? 18 : 14;
            if (ctx->cfg.cov_shape > 0) shape = ctx->cfg.cov_shape;
#define COVS_ARGS rd->T_pad, rd->d_g_group_slot.p, rd->d_g_read_rec.p, rd->d_g_slot_rec.p, rd->d_g_node_id.p, rd->d_g_step_dup.p, d_active, \
                  db->d_node_rec.p, db->d_bases.p, db->d_bitmap.p, db->d_full.p, db->d_trio_ent.p, db->d_trio_bases.p, d_abort, rd->d_long_sum.p, \
                  rd->d_long_len0.p, n_chunks, xcd_map, ablate, only_long
#define COVS_LAUNCH(UU, PP)                                                                                                                  \
            {                                                                                                                                \
                const uint32_t n_chunks = (uint32_t)((rd->T_pad + (uint64_t)COV_BLOCK * UU * PP - 1) / ((uint64_t)COV_BLOCK * UU * PP));     \
                const int grid = xcd_map ? (int)(((n_chunks + 7) / 8) * 8) : (int)n_chunks;                                                  \
                if (trio) hipLaunchKernelGGL((coverage_step_kernel<true, UU, PP>), dim3(grid), dim3(COV_BLOCK), cov_lds_bytes(COV_WIN), ctx->stream, COVS_ARGS);  \
                else hipLaunchKernelGGL((coverage_step_kernel<false, UU, PP>), dim3(grid), dim3(COV_BLOCK), cov_lds_bytes(COV_WIN), ctx->stream, COVS_ARGS);      \
            }
            switch (shape) {
                case 22: COVS_LAUNCH(2, 2) break;
                case 21: COVS_LAUNCH(2, 1) break;
                case 41: COVS_LAUNCH(4, 1) break;
                case 42: COVS_LAUNCH(4, 2) break;
                case 18: COVS_LAUNCH(1, 8) break;
                case 14: COVS_LAUNCH(1, 4) break;
                default: COVS_LAUNCH(1, 4) break;
            }
#undef COVS_LAUNCH
#undef COVS_ARGS
        }
    }
    PTX_HIP(ctx, hipGetLastError());
    lap("coverage kernels");
    db->cov_count_pending = defer_count && db->V != 0;   // the resident step: node_stats_launch counts the covered bases in its own pass
    if (db->V && !defer_count) {
        KTimer t(ctx, "popcount_kernel");
        if (db->L / db->V >= (uint64_t)ctx->cfg.ncs_prefix_min && !ctx->cfg.ncs_no_prefix)     // long nodes on average: counts from a per-stretch prefix in LDS
            hipLaunchKernelGGL(popcount_long_kernel, dim3(grid_for((db->V + 63) / 64, 4, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, db->V,
                               db->d_bit_off.p, db->d_full.p, db->d_bitmap.p, db->d_cov.p);
        else
        hipLaunchKernelGGL(popcount_kernel, dim3(grid_for(db->V, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, db->V,
                           db->d_bit_off.p, db->d_full.p, db->d_bitmap.p, db->d_cov.p);
    }
    PTX_HIP(ctx, hipGetLastError());
    lap("popcount");
    db->cov_done = true;
    return 0;
}

}  // namespace ptx
