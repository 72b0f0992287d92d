// stage_bin.hip -- a2 + a3: read -> species binning and per-species counters.
//
// Reference: process_single_read_simple rcls.rs:237-258 (min/max node id of the walk, first
// range in file order with start <= min && max <= end, else "U"; empty walk => (-1,-1) => "U")
// and the counters of equal_length_read_cls / non_equal_length_read_cls profile.rs:208-297
// (read_count, sum(read_len), #(3<=mapq<=60), #(mapq==60)) over reads with species != "U".
//
// HBM-bound streaming kernel: algorithmic bytes = 4T (node ids) + 4(R+1) (offsets) + 4R (qlen)
// + R (mapq) + 4R (species out).  One thread per read (short-read walks are ~6 steps; adjacent
// threads read adjacent id runs so the wave's loads stay within a few cache lines); the species
// counters are staged in an LDS histogram and flushed with one 64-bit atomic per touched bin.
#include <algorithm>
#include "common.hpp"
#include "wave.hpp"

namespace ptx {

constexpr int BIN_BLOCK = 256;
constexpr int BIN_LDS_SPECIES = 1024;
constexpr int BIN_REPL = 32;

template <bool SORTED>
__device__ __forceinline__ int find_species(uint32_t mn, uint32_t mx, const uint32_t *__restrict__ rs,
                                            const uint32_t *__restrict__ re, const uint32_t *__restrict__ ridx, int S) {
    if (SORTED) {
        // ranges sorted by start and pairwise disjoint: at most one range can contain [mn,mx],
        // so "first match in file order" == "the match"
        int lo = 0, hi = S;  // last i with rs[i] <= mn
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (rs[mid] <= mn) lo = mid + 1; else hi = mid;
        }
        int i = lo - 1;
        if (i >= 0 && mx <= re[i]) return (int)ridx[i];
        return -1;
    } else {
        for (int i = 0; i < S; ++i)  // file order, first match (rcls.rs:253-257)
            if (mn >= rs[i] && mx <= re[i]) return (int)ridx[i];
        return -1;
    }
}

template <bool SORTED, bool LDS_HIST>
__global__ void __launch_bounds__(BIN_BLOCK) bin_reads_kernel(
    uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
    const uint32_t *__restrict__ qlen, const uint8_t *__restrict__ mapq, const uint32_t *__restrict__ rs,
    const uint32_t *__restrict__ re, const uint32_t *__restrict__ ridx, int S, int32_t *__restrict__ species_out,
    unsigned long long *__restrict__ counters_rep /* [BIN_REPL][4][S]: read_count, base_sum, less_multi, uniq_count */) {
    // every workgroup ends with a handful of global atomics on the same few words; spreading the workgroups
    // over BIN_REPL replicas keeps that tail from serialising (same-address atomics cost ~12 ns each)
    unsigned long long *__restrict__ counters = counters_rep + (size_t)(blockIdx.x % BIN_REPL) * 4 * S;
    __shared__ unsigned int s_cnt[LDS_HIST ? 3 * BIN_LDS_SPECIES : 1];
    __shared__ unsigned long long s_base[LDS_HIST ? BIN_LDS_SPECIES : 1];
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < 3 * S; i += BIN_BLOCK) s_cnt[i] = 0;
        for (int i = threadIdx.x; i < S; i += BIN_BLOCK) s_base[i] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    // whole waves iterate together (lanes past R stay in the loop, masked) so wave-level aggregation is legal
    for (uint64_t base = (uint64_t)blockIdx.x * BIN_BLOCK + (threadIdx.x - lane); base < R; base += (uint64_t)gridDim.x * BIN_BLOCK) {
        const uint64_t r = base + lane;
        int sp = -1;
        uint32_t q = 0;
        bool lm = false, uq = false;
        uint32_t b = 0, e = 0;
        if (r < R) { b = step_off[r]; e = step_off[r + 1]; }
        uint32_t mn = 0xFFFFFFFFu, mx = 0;
        // walks of more than 64 steps (long reads): the wave's four 16-lane rows each scan one such walk at a time (one
        // 64-byte line per row and iteration), the row's min / max come from four DPP steps
        const bool long_walk = e - b > 64u;
        const int row = lane >> 4, rl = lane & 15;
        for (unsigned long long todo = __ballot(long_walk); todo;) {
            int src = -1;
            unsigned long long t = todo;
            for (int w_ = 0; w_ <= row && t; ++w_) { src = (w_ == row) ? __ffsll((long long)t) - 1 : -1; t &= t - 1; }   // my row's walk: the row-th pending lane
            for (int w_ = 0; w_ < 4 && todo; ++w_) todo &= todo - 1;
            const uint32_t bb = __shfl(b, src < 0 ? 0 : src), ee = __shfl(e, src < 0 ? 0 : src);
            uint32_t m1 = 0xFFFFFFFFu, m2 = 0;
            if (src >= 0)
                for (uint32_t i = bb + rl; i < ee; i += 16) { const uint32_t v = node_id[i]; m1 = min(m1, v); m2 = max(m2, v); }
            m1 = row_reduce(m1, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
            m2 = row_reduce(m2, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            // the owner lane may sit in another row: fetch the result from the row that scanned its walk
            for (int rr = 0; rr < 4; ++rr) {
                const int owner = __shfl(src, rr * 16);
                const uint32_t a1 = __shfl(m1, rr * 16), a2 = __shfl(m2, rr * 16);
                if (owner >= 0 && lane == owner) { mn = a1; mx = a2; }
            }
        }
        if (r < R) {
            if (e > b) {
                if (!long_walk)
                    for (uint32_t i = b; i < e; ++i) {
                        uint32_t v = node_id[i];
                        mn = min(mn, v);
                        mx = max(mx, v);
                    }
                sp = find_species<SORTED>(mn, mx, rs, re, ridx, S);
            }
            species_out[r] = sp;
            if (sp >= 0) {
                q = qlen[r];
                uint32_t m = mapq[r];
                lm = (m >= 3 && m <= 60);
                uq = (m == 60);
            }
        }
        // counters: when every binned lane of the wave has the same species (the usual case inside one
        // species' reads) one lane adds the wave totals; 64 same-address LDS atomics would serialise
        const unsigned long long have = __ballot(sp >= 0);
        if (have == 0) continue;
        const int sp0 = __shfl(sp, __ffsll((long long)have) - 1);
        const bool uniform = __all(sp < 0 || sp == sp0);
        if (uniform) {
            unsigned long long qs = q;   // q is 0 on lanes without a species
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) qs += __shfl_down(qs, off);
            const unsigned int c = __popcll(have), l = __popcll(__ballot(lm)), u = __popcll(__ballot(uq));
            if (lane == 0) {
                if (LDS_HIST) {
                    atomicAdd(&s_cnt[sp0], c); atomicAdd(&s_base[sp0], qs);
                    if (l) atomicAdd(&s_cnt[S + sp0], l);
                    if (u) atomicAdd(&s_cnt[2 * S + sp0], u);
                } else {
                    atomicAdd(&counters[sp0], (unsigned long long)c); atomicAdd(&counters[S + sp0], qs);
                    if (l) atomicAdd(&counters[2 * S + sp0], (unsigned long long)l);
                    if (u) atomicAdd(&counters[3 * S + sp0], (unsigned long long)u);
                }
            }
        } else if (sp >= 0) {
            if (LDS_HIST) {
                atomicAdd(&s_cnt[sp], 1u);
                atomicAdd(&s_base[sp], (unsigned long long)q);
                if (lm) atomicAdd(&s_cnt[S + sp], 1u);
                if (uq) atomicAdd(&s_cnt[2 * S + sp], 1u);
            } else {
                atomicAdd(&counters[sp], 1ull);
                atomicAdd(&counters[S + sp], (unsigned long long)q);
                if (lm) atomicAdd(&counters[2 * S + sp], 1ull);
                if (uq) atomicAdd(&counters[3 * S + sp], 1ull);
            }
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < S; i += BIN_BLOCK) {
            unsigned int c = s_cnt[i];
            if (c) {
                atomicAdd(&counters[i], (unsigned long long)c);
                atomicAdd(&counters[S + i], s_base[i]);
                unsigned int l = s_cnt[S + i], u = s_cnt[2 * S + i];
                if (l) atomicAdd(&counters[2 * S + i], (unsigned long long)l);
                if (u) atomicAdd(&counters[3 * S + i], (unsigned long long)u);
            }
        }
    }
}

// The same pass for up to BIN_LDS_SPECIES species with EVERYTHING a read needs about the species in LDS (dynamic, sized by S:
// 44 bytes per species): the ranges for the search, {first id, node base, node count} for the slot record, the counter
// histogram.  The kernel is bound by its chain of dependent loads (PMC: 94 % of wave time waiting, 33 M VALU instructions
// for 1e7 reads): with the tables in global memory a read cost ~17 dependent round trips (offsets, ~8 walk loads one after
// the other, 7 search levels, 3 species-table loads); here it is the offsets, one batch of up to 8 walk loads issued
// together (a second batch for walks of 9..64 steps), LDS searches, and the per-read columns, which are requested before
// the species is known.
template <bool SORTED>
__global__ void __launch_bounds__(BIN_BLOCK) bin_reads_lds_kernel(
    uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
    const uint32_t *__restrict__ qlen, const uint8_t *__restrict__ mapq, const uint32_t *__restrict__ rs,
    const uint32_t *__restrict__ re, const uint32_t *__restrict__ ridx, int S, int32_t *__restrict__ species_out,
    unsigned long long *__restrict__ counters_rep) {
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long *s_base = s_dyn;                                   // [S]
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_dyn + S);    // [3S]
    uint32_t *s_rs = s_cnt + 3 * S, *s_re = s_rs + S, *s_ridx = s_re + S; // [S] each, in search order
    unsigned long long *__restrict__ counters = counters_rep + (size_t)(blockIdx.x % BIN_REPL) * 4 * S;
    for (int i = threadIdx.x; i < S; i += BIN_BLOCK) {
        s_base[i] = 0; s_cnt[i] = 0; s_cnt[S + i] = 0; s_cnt[2 * S + i] = 0;
        s_rs[i] = rs[i]; s_re[i] = re[i]; s_ridx[i] = ridx[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * BIN_BLOCK + (threadIdx.x - lane); base < R; base += (uint64_t)gridDim.x * BIN_BLOCK) {
        const uint64_t r = base + lane;
        int sp = -1;
        uint32_t b = 0, e = 0, q = 0, m = 255u;
        if (r < R) {
            b = step_off[r]; e = step_off[r + 1];
            q = qlen[r]; m = mapq[r];                                     // requested now, used once the species is known
        }
        uint32_t mn = 0xFFFFFFFFu, mx = 0;
        const uint32_t k = e - b;
        const bool long_walk = k > 64u;
        // walks of more than 64 steps (long reads): the wave's four 16-lane rows each scan one such walk at a time
        const int row = lane >> 4, rl = lane & 15;
        for (unsigned long long todo = __ballot(long_walk); todo;) {
            int src = -1;
            unsigned long long t = todo;
            for (int w_ = 0; w_ <= row && t; ++w_) { src = (w_ == row) ? __ffsll((long long)t) - 1 : -1; t &= t - 1; }
            for (int w_ = 0; w_ < 4 && todo; ++w_) todo &= todo - 1;
            const uint32_t bb = __shfl(b, src < 0 ? 0 : src), ee = __shfl(e, src < 0 ? 0 : src);
            uint32_t m1 = 0xFFFFFFFFu, m2 = 0;
            if (src >= 0)
                for (uint32_t i = bb + rl; i < ee; i += 16) { const uint32_t v = node_id[i]; m1 = min(m1, v); m2 = max(m2, v); }
            m1 = row_reduce(m1, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
            m2 = row_reduce(m2, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            for (int rr = 0; rr < 4; ++rr) {
                const int owner = __shfl(src, rr * 16);
                const uint32_t a1 = __shfl(m1, rr * 16), a2 = __shfl(m2, rr * 16);
                if (owner >= 0 && lane == owner) { mn = a1; mx = a2; }
            }
        }
        if (r < R && k && !long_walk) {
            // up to 8 node ids in ONE batch of loads (the usual short read), the rest eight at a time
            for (uint32_t i0 = b; i0 < e; i0 += 8) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = i0 + j < e ? node_id[i0 + j] : 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (i0 + j < e) { mn = min(mn, v[j]); mx = max(mx, v[j]); }
            }
        }
        if (r < R) {
            if (k) sp = find_species<SORTED>(mn, mx, s_rs, s_re, s_ridx, S);
            species_out[r] = sp;
        }
        const bool lm = sp >= 0 && m >= 3 && m <= 60, uq = sp >= 0 && m == 60;
        if (sp < 0) q = 0;
        const unsigned long long have = __ballot(sp >= 0);
        if (have == 0) continue;
        const int sp0 = __shfl(sp, __ffsll((long long)have) - 1);
        const bool uniform = __all(sp < 0 || sp == sp0);
        if (uniform) {
            unsigned long long qs = q;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) qs += __shfl_down(qs, off);
            const unsigned int c = __popcll(have), l = __popcll(__ballot(lm)), u = __popcll(__ballot(uq));
            if (lane == 0) {
                atomicAdd(&s_cnt[sp0], c); atomicAdd(&s_base[sp0], qs);
                if (l) atomicAdd(&s_cnt[S + sp0], l);
                if (u) atomicAdd(&s_cnt[2 * S + sp0], u);
            }
        } else if (sp >= 0) {
            atomicAdd(&s_cnt[sp], 1u);
            atomicAdd(&s_base[sp], (unsigned long long)q);
            if (lm) atomicAdd(&s_cnt[S + sp], 1u);
            if (uq) atomicAdd(&s_cnt[2 * S + sp], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += BIN_BLOCK) {
        unsigned int c = s_cnt[i];
        if (c) {
            atomicAdd(&counters[i], (unsigned long long)c);
            atomicAdd(&counters[S + i], s_base[i]);
            unsigned int l = s_cnt[S + i], u = s_cnt[2 * S + i];
            if (l) atomicAdd(&counters[2 * S + i], (unsigned long long)l);
            if (u) atomicAdd(&counters[3 * S + i], (unsigned long long)u);
        }
    }
}

// The binning pass of RESIDENT (locus-grouped) reads, in SLOT order: thread = slot of the grouped copy.  The walk comes from
// the grouped stream (neighbouring slots hold neighbouring walks: coalesced), read length / MAPQ from slot-order copies made
// at upload (g_qm), and the 8-byte slot record {species, node base - first id} that the coverage pass gathers is written
// COALESCED.  (Round 2 ran in read order and scattered a 16-byte record per read over the slot array: WRITE_SIZE 368 MB for
// 1e7 reads, 9x the algorithmic output; 13 ms at 1e8 reads.)  The per-read species array in file order is not written here:
// species_ensure() scatters it from the slot records when a caller asks for it (report, routing).
// Slot record (coding in common.hpp): x >= 0 species, usable by the coverage pass; x == -1 "U"; below: binned, but the row is
// dropped before get_node_abundances (drop flag).  A binned walk cannot leave its species' graph: its ids lie inside the
// species' range (rcls.rs:253-257) and db_upload refuses a range that does not span exactly the graph's nodes (optimize_otu
// derives nvert from the range, profile.rs:2938) -- which is why the coverage pass places a node with one add and no test.
template <bool SORTED, bool LDS_TAB>
__global__ void __launch_bounds__(BIN_BLOCK) bin_slots_kernel(
    uint32_t n_slots, const uint4 *__restrict__ read_rec, const uint32_t *__restrict__ node_id, const uint2 *__restrict__ g_qm,
    const uint8_t *__restrict__ g_flag /* null: no drop flags */, const uint32_t *__restrict__ rs, const uint32_t *__restrict__ re,
    const uint32_t *__restrict__ ridx, int S, uint2 *__restrict__ slot_rec, const uint32_t *__restrict__ sp_first_id /* null: db without graphs */,
    const uint32_t *__restrict__ node_base, unsigned long long *__restrict__ counters_rep,
    const uint32_t *__restrict__ node_len /* null: no walk sums */, uint32_t *__restrict__ long_sum, uint32_t *__restrict__ long_len0) {
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long *s_base = s_dyn;                                   // [S]
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_dyn + (LDS_TAB ? S : 0));    // [3S]
    uint32_t *s_rs = s_cnt + (LDS_TAB ? 3 * S : 0), *s_re = s_rs + (LDS_TAB ? S : 0), *s_ridx = s_re + (LDS_TAB ? S : 0);
    uint32_t *s_first = s_ridx + (LDS_TAB ? S : 0), *s_nb = s_first + (LDS_TAB ? S : 0);
    unsigned long long *__restrict__ counters = counters_rep + (size_t)(blockIdx.x % BIN_REPL) * 4 * S;
    if (LDS_TAB) {
        for (int i = threadIdx.x; i < S; i += BIN_BLOCK) {
            s_base[i] = 0; s_cnt[i] = 0; s_cnt[S + i] = 0; s_cnt[2 * S + i] = 0;
            s_rs[i] = rs[i]; s_re[i] = re[i]; s_ridx[i] = ridx[i];
            s_first[i] = sp_first_id ? sp_first_id[i] : 0u;
            s_nb[i] = sp_first_id ? node_base[i] : 0u;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    // (round 6: the records of the NEXT slot of this thread are requested while this one's walk is read -- the pass is a chain of two dependent levels
    // per slot, record -> walk ids, and a thread takes ~190 slots one after the other at 1e8 reads)
    const uint64_t stride = (uint64_t)gridDim.x * BIN_BLOCK;
    uint4 n_rr = make_uint4(0u, 0u, 0u, 0u);
    uint2 n_qm = make_uint2(0u, 255u);
    uint8_t n_fl = 0;
    {
        const uint64_t r0 = (uint64_t)blockIdx.x * BIN_BLOCK + threadIdx.x;
        if (r0 < n_slots) { n_rr = read_rec[r0]; n_qm = g_qm[r0]; if (g_flag) n_fl = g_flag[r0]; }
    }
    for (uint64_t base = (uint64_t)blockIdx.x * BIN_BLOCK + (threadIdx.x - lane); base < n_slots; base += stride) {
        const uint64_t r = base + lane;
        int sp = -1;
        uint32_t b = 0, e = 0, q = 0, m = 255u;
        uint8_t fl = 0;
        if (r < n_slots) {
            const uint4 rr = n_rr;
            const uint2 qm = n_qm;
            b = rr.x; e = rr.x + rr.y; q = qm.x; m = qm.y;
            fl = n_fl;
        }
        {
            const uint64_t rn = r + stride;
            if (rn < n_slots) { n_rr = read_rec[rn]; n_qm = g_qm[rn]; if (g_flag) n_fl = g_flag[rn]; }
        }
        uint32_t mn = 0xFFFFFFFFu, mx = 0;
        const uint32_t k = e - b;
        const bool long_walk = k > 64u;
        // walks of more than 64 steps (long reads): the wave's four 16-lane rows each scan one such walk at a time
        const int row = lane >> 4, rl = lane & 15;
        for (unsigned long long todo = __ballot(long_walk); todo;) {
            int src = -1;
            unsigned long long t = todo;
            for (int w_ = 0; w_ <= row && t; ++w_) { src = (w_ == row) ? __ffsll((long long)t) - 1 : -1; t &= t - 1; }
            for (int w_ = 0; w_ < 4 && todo; ++w_) todo &= todo - 1;
            const uint32_t bb = __shfl(b, src < 0 ? 0 : src), ee = __shfl(e, src < 0 ? 0 : src);
            uint32_t m1 = 0xFFFFFFFFu, m2 = 0;
            if (src >= 0)
                for (uint32_t i = bb + rl; i < ee; i += 16) { const uint32_t v = node_id[i]; m1 = min(m1, v); m2 = max(m2, v); }
            m1 = row_reduce(m1, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
            m2 = row_reduce(m2, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            // Round 6: the walk sums of the long-read coverage pass (profile.rs:857-859: `seen` of a walk's last step = the node lengths of all steps before
            // it; and the length of its first node) are taken HERE, by the row that has just streamed the walk's ids and now knows its species -- a second
            // pass over ids that are in the caches, one 4-byte gather per step -- instead of by walk_sum_kernel (2.9 of cfg5's 28 ms per step), which found
            // every step's walk again through the slot of its group, the read record and the slot record.
            if (node_len) {
                const int sp_w = src >= 0 && ee > bb ? (LDS_TAB ? find_species<SORTED>(m1, m2, s_rs, s_re, s_ridx, S) : find_species<SORTED>(m1, m2, rs, re, ridx, S)) : -1;
                uint32_t sum = 0, l0 = 0;
                if (sp_w >= 0) {
                    const uint32_t delta = (LDS_TAB ? s_nb[sp_w] : node_base[sp_w]) - (LDS_TAB ? s_first[sp_w] : sp_first_id[sp_w]);
                    for (uint32_t i = bb + rl; i + 1 < ee; i += 16) { const uint32_t ln = node_len[node_id[i] + delta]; sum += ln; if (i == bb) l0 = ln; }
                }
                sum = row_reduce(sum, [](uint32_t x, uint32_t y) { return x + y; });
                l0 = row_reduce(l0, [](uint32_t x, uint32_t y) { return x | y; });
                if (src >= 0 && rl == 0) { const uint64_t slot = base + (uint64_t)src; long_sum[slot] = sp_w >= 0 ? sum : 0u; long_len0[slot] = l0; }
            }
            for (int rr_ = 0; rr_ < 4; ++rr_) {
                const int owner = __shfl(src, rr_ * 16);
                const uint32_t a1 = __shfl(m1, rr_ * 16), a2 = __shfl(m2, rr_ * 16);
                if (owner >= 0 && lane == owner) { mn = a1; mx = a2; }
            }
        }
        if (r < n_slots && k && !long_walk) {
            for (uint32_t i0 = b; i0 < e; i0 += 8) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = i0 + j < e ? node_id[i0 + j] : 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (i0 + j < e) { mn = min(mn, v[j]); mx = max(mx, v[j]); }
            }
        }
        if (r < n_slots) {
            if (k) sp = LDS_TAB ? find_species<SORTED>(mn, mx, s_rs, s_re, s_ridx, S) : find_species<SORTED>(mn, mx, rs, re, ridx, S);
            uint2 rec = make_uint2(0xFFFFFFFFu, 0u);
            if (sp >= 0) {
                uint32_t first = 0, nb = 0;
                if (LDS_TAB) { first = s_first[sp]; nb = s_nb[sp]; }
                else if (sp_first_id) { first = sp_first_id[sp]; nb = node_base[sp]; }
                rec.x = fl ? (uint32_t)(-sp - 2) : (uint32_t)sp;
                rec.y = nb - first;
            }
            slot_rec[r] = rec;
        }
        const bool lm = sp >= 0 && m >= 3 && m <= 60, uq = sp >= 0 && m == 60;
        if (sp < 0) q = 0;
        const unsigned long long have = __ballot(sp >= 0);
        if (have == 0) continue;
        const int sp0 = __shfl(sp, __ffsll((long long)have) - 1);
        const bool uniform = __all(sp < 0 || sp == sp0);
        if (uniform) {
            unsigned long long qs = q;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) qs += __shfl_down(qs, off);
            const unsigned int c = __popcll(have), l = __popcll(__ballot(lm)), u = __popcll(__ballot(uq));
            if (lane == 0) {
                if (LDS_TAB) {
                    atomicAdd(&s_cnt[sp0], c); atomicAdd(&s_base[sp0], qs);
                    if (l) atomicAdd(&s_cnt[S + sp0], l);
                    if (u) atomicAdd(&s_cnt[2 * S + sp0], u);
                } else {
                    atomicAdd(&counters[sp0], (unsigned long long)c); atomicAdd(&counters[S + sp0], qs);
                    if (l) atomicAdd(&counters[2 * S + sp0], (unsigned long long)l);
                    if (u) atomicAdd(&counters[3 * S + sp0], (unsigned long long)u);
                }
            }
        } else if (sp >= 0) {
            if (LDS_TAB) {
                atomicAdd(&s_cnt[sp], 1u);
                atomicAdd(&s_base[sp], (unsigned long long)q);
                if (lm) atomicAdd(&s_cnt[S + sp], 1u);
                if (uq) atomicAdd(&s_cnt[2 * S + sp], 1u);
            } else {
                atomicAdd(&counters[sp], 1ull);
                atomicAdd(&counters[S + sp], (unsigned long long)q);
                if (lm) atomicAdd(&counters[2 * S + sp], 1ull);
                if (uq) atomicAdd(&counters[3 * S + sp], 1ull);
            }
        }
    }
    if (LDS_TAB) {
        __syncthreads();
        for (int i = threadIdx.x; i < S; i += BIN_BLOCK) {
            unsigned int c = s_cnt[i];
            if (c) {
                atomicAdd(&counters[i], (unsigned long long)c);
                atomicAdd(&counters[S + i], s_base[i]);
                unsigned int l = s_cnt[S + i], u = s_cnt[2 * S + i];
                if (l) atomicAdd(&counters[2 * S + i], (unsigned long long)l);
                if (u) atomicAdd(&counters[3 * S + i], (unsigned long long)u);
            }
        }
    }
}

// drop flags in slot order (one scattered byte per read, once per change of the flags -- not per binning pass)
__global__ void __launch_bounds__(256) flags_to_slots_kernel(uint64_t R, const uint8_t *__restrict__ flags, const uint32_t *__restrict__ slot_of,
                                                             uint8_t *__restrict__ g_flag) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256) {
        const uint32_t s = slot_of[r];
        if (s != 0xFFFFFFFFu) g_flag[s] = flags[r];
    }
}

// species of read r (file order) from the slot records of the grouped copy
__device__ __forceinline__ int species_of_read(uint64_t r, const uint32_t *__restrict__ slot_of, const uint2 *__restrict__ slot_rec) {
    const uint32_t s = slot_of[r];
    return s == 0xFFFFFFFFu ? -1 : slot_species((int32_t)slot_rec[s].x);
}
__global__ void __launch_bounds__(256) species_gather_kernel(uint64_t R, const uint32_t *__restrict__ slot_of, const uint2 *__restrict__ slot_rec,
                                                             int32_t *__restrict__ species_out) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256) species_out[r] = species_of_read(r, slot_of, slot_rec);
}

// sums the replicas and appends the head of (species, qlen) -- everything the host reads after binning sits in
// one contiguous block: [4*S u64 sums][BIN_PREFIX i32 species][BIN_PREFIX u32 qlen]
__global__ void __launch_bounds__(256) bin_reduce_kernel(int n, const unsigned long long *__restrict__ rep, unsigned long long *__restrict__ out,
                                                         uint32_t npre, const int32_t *__restrict__ species /* null: from the slot records */,
                                                         const uint32_t *__restrict__ qlen, const uint32_t *__restrict__ slot_of, const uint2 *__restrict__ slot_rec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        unsigned long long s = 0;
        for (int r = 0; r < BIN_REPL; ++r) s += rep[(size_t)r * n + i];
        out[i] = s;
    }
    if ((uint32_t)i < npre) {
        int32_t *pre_sp = reinterpret_cast<int32_t *>(out + n);
        uint32_t *pre_q = reinterpret_cast<uint32_t *>(pre_sp + BIN_PREFIX);
        pre_sp[i] = species ? species[i] : species_of_read(i, slot_of, slot_rec);
        pre_q[i] = qlen[i];
    }
}

size_t bin_counter_words(uint32_t S) { return (size_t)(BIN_REPL + 1) * 4 * S + BIN_PREFIX; }
size_t bin_result_words(uint32_t S) { return (size_t)4 * S + BIN_PREFIX; }

// d_counters: bin_counter_words(S) u64 = [4*S sums][prefix block][BIN_REPL replicas of 4*S]
int bin_reads_launch(Ctx *ctx, const Db *db, Reads *rd, unsigned long long *d_counters) {
    int S = (int)db->S;
    PTX_HIP(ctx, hipMemsetAsync(d_counters, 0, bin_counter_words(S) * sizeof(unsigned long long), ctx->stream));
    rd->species_valid = false;
    if (rd->R == 0) { rd->binned = true; return 0; }
    unsigned long long *d_final = d_counters;
    d_counters = d_counters + bin_result_words(S);   // replicas
    const bool lds = S <= BIN_LDS_SPECIES;
    const size_t dyn = (size_t)S * (rd->grouped ? 40 : 32);   // LDS tables per species: 8 + 12 (counters) + 12 (ranges) + 8 ({first id, node base}; slot order only)
    if (rd->grouped) {
        // resident reads: slot order (coalesced slot records); drop flags reach the slots once per change
        const uint8_t *g_flag = nullptr;
        if (rd->has_flags) {
            if (!rd->g_flags_valid) {
                PTX_HIP(ctx, rd->d_g_flag.alloc(rd->R));
                hipLaunchKernelGGL(flags_to_slots_kernel, dim3(grid_for(rd->R, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, rd->R, rd->d_flags.p,
                                   rd->d_slot_of.p, rd->d_g_flag.p);
                rd->g_flags_valid = true;
            }
            g_flag = rd->d_g_flag.p;
        }
        if (rd->n_slots) {
            KTimer t(ctx, "bin_slots_kernel");
            const int grid = grid_for(rd->n_slots, BIN_BLOCK, ctx->n_cu * 8);
            // (option walk_sum_in_bin: walk sums of long walks on the way -- a db with graphs, reads that hold walks of more than 64 steps)
            const bool sums = rd->n_long && db->d_node_len.p && db->d_sp_first_id.p && rd->d_long_sum.p && rd->d_long_len0.p && ctx->cfg.walk_sum_in_bin;
            rd->long_sums_db = sums ? db->uid : 0;
#define BIN_ARGS rd->n_slots, rd->d_g_read_rec.p, rd->d_g_node_id.p, rd->d_g_qm.p, g_flag, db->d_rng_start.p, db->d_rng_end.p, db->d_rng_idx.p, S, \
                 rd->d_g_slot_rec.p, db->d_sp_first_id.p, db->d_node_base.p, d_counters, sums ? (const uint32_t *)db->d_node_len.p : (const uint32_t *)nullptr, \
                 rd->d_long_sum.p, rd->d_long_len0.p
            if (db->ranges_sorted_disjoint) {
                if (lds) hipLaunchKernelGGL((bin_slots_kernel<true, true>), dim3(grid), dim3(BIN_BLOCK), dyn, ctx->stream, BIN_ARGS);
                else hipLaunchKernelGGL((bin_slots_kernel<true, false>), dim3(grid), dim3(BIN_BLOCK), 0, ctx->stream, BIN_ARGS);
            } else {
                if (lds) hipLaunchKernelGGL((bin_slots_kernel<false, true>), dim3(grid), dim3(BIN_BLOCK), dyn, ctx->stream, BIN_ARGS);
                else hipLaunchKernelGGL((bin_slots_kernel<false, false>), dim3(grid), dim3(BIN_BLOCK), 0, ctx->stream, BIN_ARGS);
            }
#undef BIN_ARGS
        }
    } else {
        // a slice kept as plain columns (to be routed away): read order, the species array is the product
        PTX_HIP(ctx, rd->d_species.alloc(rd->R));
        int grid = grid_for(rd->R, BIN_BLOCK, ctx->n_cu * 8);
        KTimer t(ctx, "bin_reads_kernel");
#define BIN_ARGS rd->R, rd->d_step_off.p, rd->d_node_id.p, rd->d_qlen.p, rd->d_mapq.p, db->d_rng_start.p, db->d_rng_end.p, \
                 db->d_rng_idx.p, S, rd->d_species.p, d_counters
        if (db->ranges_sorted_disjoint) {
            if (lds) hipLaunchKernelGGL((bin_reads_lds_kernel<true>), dim3(grid), dim3(BIN_BLOCK), dyn, ctx->stream, BIN_ARGS);
            else hipLaunchKernelGGL((bin_reads_kernel<true, false>), dim3(grid), dim3(BIN_BLOCK), 0, ctx->stream, BIN_ARGS);
        } else {
            if (lds) hipLaunchKernelGGL((bin_reads_lds_kernel<false>), dim3(grid), dim3(BIN_BLOCK), dyn, ctx->stream, BIN_ARGS);
            else hipLaunchKernelGGL((bin_reads_kernel<false, false>), dim3(grid), dim3(BIN_BLOCK), 0, ctx->stream, BIN_ARGS);
        }
#undef BIN_ARGS
        rd->species_valid = true;
    }
    const uint32_t npre = (uint32_t)std::min<uint64_t>(rd->R, BIN_PREFIX);
    hipLaunchKernelGGL(bin_reduce_kernel, dim3((std::max<uint32_t>(4 * S, npre) + 255) / 256), dim3(256), 0, ctx->stream, 4 * S, d_counters, d_final,
                       npre, rd->grouped ? nullptr : rd->d_species.p, rd->d_qlen.p, rd->d_slot_of.p, rd->d_g_slot_rec.p);
    PTX_HIP(ctx, hipGetLastError());
    rd->binned = true;
    return 0;
}

// the per-read species array in file order (report, routing, host-side equal-length test): resident reads keep the species
// per SLOT; this gathers them on request
int species_ensure(Ctx *ctx, Reads *rd) {
    if (rd->species_valid || rd->R == 0) return 0;
    PTX_HIP(ctx, rd->d_species.alloc(rd->R));
    hipLaunchKernelGGL(species_gather_kernel, dim3(grid_for(rd->R, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, rd->R, rd->d_slot_of.p, rd->d_g_slot_rec.p,
                       rd->d_species.p);
    PTX_HIP(ctx, hipGetLastError());
    rd->species_valid = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// a3 finishing on the device (species_profiling, profile.rs:299-349; same arithmetic as
// pantax_hip_species_profile in api_host.cpp), so that a resident step never waits for the host between
// binning and coverage.  One wave: (1) the equal-length test over the first 1000 binned rows
// (:312-319) by ballots over 64 rows at a time, (2) the MAPQ filter (:224-245) and
// predicted_coverage = base_count / avg_len (:336) per species.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) species_profile_kernel(uint64_t R, const int32_t *__restrict__ species /* null: from the slot records */,
                                                             const uint32_t *__restrict__ slot_of, const uint2 *__restrict__ slot_rec, const uint32_t *__restrict__ qlen,
                                                             uint32_t S, const unsigned long long *__restrict__ counters /*[4][S]*/,
                                                             const double *__restrict__ avg_len, int filtered, uint8_t *__restrict__ keep_out,
                                                             double *__restrict__ absolute_out) {
    const int lane = threadIdx.x;
    uint32_t seen = 0;
    long long first_len = -1;
    bool equal = true;
    // the head (2048 rows, nearly always enough to see 1000 binned reads) is loaded in one go: 32 independent loads
    // per lane instead of 32 dependent round trips; further rows, if ever needed, 64 at a time
    constexpr int PRE = 32;
    int sp_pre[PRE];
    uint32_t q_pre[PRE];
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
        const uint64_t r = (uint64_t)it * 64 + lane;
        sp_pre[it] = r < R ? (species ? species[r] : species_of_read(r, slot_of, slot_rec)) : -1;
        q_pre[it] = r < R ? qlen[r] : 0u;
    }
    auto feed = [&](int sp, uint32_t q) {
        const unsigned long long b = __ballot(sp >= 0);
        if (!b) return;
        if (seen == 0) first_len = (long long)__shfl(q, __ffsll((long long)b) - 1);
        const uint32_t rank = seen + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (__any(sp >= 0 && rank < 1000 && (long long)q != first_len)) equal = false;
        seen += (uint32_t)__popcll(b);
    };
#pragma unroll
    for (int it = 0; it < PRE; ++it) if (seen < 1000) feed(sp_pre[it], q_pre[it]);
    for (uint64_t base = (uint64_t)PRE * 64; base < R && seen < 1000; base += 64) {
        const uint64_t r = base + lane;
        int sp = -1;
        uint32_t q = 0;
        if (r < R) { sp = species ? species[r] : species_of_read(r, slot_of, slot_rec); q = qlen[r]; }
        feed(sp, q);
    }
    if (seen == 0) equal = false;
    for (uint32_t s = lane; s < S; s += 64) {
        const long long rc = (long long)counters[s], bs = (long long)counters[(size_t)S + s], lm = (long long)counters[2 * (size_t)S + s],
                        uq = (long long)counters[3 * (size_t)S + s];
        uint8_t keep = 0;
        double absolute = 0.0;
        bool ok = rc != 0;
        if (ok && filtered) ok = lm != 0 && uq > 0 && (double)lm > (double)rc / 10.0;
        if (ok) ok = avg_len[s] > 0.0;
        if (ok) {
            const long long base_count = equal ? rc * first_len : bs;
            keep = 1;
            absolute = (double)base_count / avg_len[s];
        }
        keep_out[s] = keep;
        absolute_out[s] = absolute;
    }
}

int species_profile_launch(Ctx *ctx, const Db *db, const Reads *rd, const unsigned long long *d_counters, const double *d_avg_len, int filtered,
                           uint8_t *d_keep, double *d_absolute) {
    hipLaunchKernelGGL(species_profile_kernel, dim3(1), dim3(64), 0, ctx->stream, rd->R, rd->grouped ? nullptr : rd->d_species.p, rd->d_slot_of.p,
                       rd->d_g_slot_rec.p, rd->d_qlen.p, db->S, d_counters, d_avg_len,
                       filtered, d_keep, d_absolute);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx
