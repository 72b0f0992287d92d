// stage_lad.hip -- a9/a10/a12 on device: trio statistics, candidate masks + path_cov_ratio,
// LP row grouping, and the batched exact LAD solver that replaces the Gurobi/HiGHS/CBC/GLPK
// backends (profile.rs:1297-1511, 2689-2882).
//
// The reference LP per species (gurobi_opt, profile.rs:1312-1460):
//     min (1/n) sum_{v: a_v>0} y_v,   y_v >= +-(sum_k A_vk x_k - a_v),   0 <= x_k <= 1.05*max(a)
// with A_vk = 1 iff node v lies on candidate path k (:1333-1342); the binary indicators are inert
// (SURVEY.md 8c).  That is a least-absolute-deviation fit in p <= 64 variables.  Rows with the
// same membership pattern m (a p-bit mask) only see s_m = sum_{k in m} x_k, so after ONE
// HBM-bound pass that sorts the n covered nodes by (species, mask, a) the objective is
//     f(x) = (1/n) sum_patterns F_m(s_m),  F_m(s) = sum_{rows of m} |s - a|,
// and f, its sub-gradient and exact line searches cost O(#patterns * log n) binary searches.
// The solver is an exact active-set (Bloomfield-Steiger / Barrodale-Roberts) descent on that
// representation, one workgroup per species, all species of the batch in one launch.
#include <algorithm>
#include <cstdio>
#include <vector>
#include "lad.hpp"
#include "row_sample.hpp"
#include "primitives.hpp"
#include "wave.hpp"
#include "scan_chained.hpp"

namespace ptx {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ double block_sum_f64(double v, double *red) {
    v = wave_reduce(v, [](double x, double y) { return x + y; });
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += red[w];   // fixed order: deterministic
    __syncthreads();
    return t;
}
template <int NT>
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *red) {
    v = wave_reduce(v, [](unsigned long long x, unsigned long long y) { return x + y; });
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    __syncthreads();
    return t;
}
template <int NT>
__device__ __forceinline__ double block_max_f64(double v, double *red) {
    v = wave_reduce(v, [](double x, double y) { return fmax(x, y); });
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) t = fmax(t, red[w]);
    __syncthreads();
    return t;
}

__device__ __forceinline__ uint32_t lower_bound_a(const double *__restrict__ a, uint32_t lo, uint32_t hi, double v) {
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
__device__ __forceinline__ uint32_t upper_bound_a(const double *__restrict__ a, uint32_t lo, uint32_t hi, double v) {
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (a[m] <= v) lo = m + 1; else hi = m; }
    return lo;
}
// 64-ary searches by a whole wave64 (all lanes pass the same arguments, all get the result): each
// round probes 64 evenly spaced elements with one gather and narrows the range ~65x by a ballot, so a
// search over 10^5 rows costs 3 dependent memory round trips instead of 17.
__device__ __forceinline__ uint32_t wave_bound(const double *__restrict__ a, uint32_t lo, uint32_t hi, double v, bool upper) {
    const uint32_t lane = threadIdx.x & 63;
    while (hi - lo > 64) {
        const uint32_t m = hi - lo;
        const uint32_t pj = lo + (uint32_t)(((uint64_t)(lane + 1) * m) / 65);
        const double x = a[pj];
        const bool before = upper ? (x <= v) : (x < v);
        const int c = __popcll(__ballot(before));
        const uint32_t nlo = c > 0 ? lo + (uint32_t)(((uint64_t)c * m) / 65) + 1 : lo;
        const uint32_t nhi = c < 64 ? lo + (uint32_t)(((uint64_t)(c + 1) * m) / 65) : hi;
        lo = nlo; hi = nhi;
    }
    bool before = false;
    if (lo + lane < hi) { const double x = a[lo + lane]; before = upper ? (x <= v) : (x < v); }
    return lo + (uint32_t)__popcll(__ballot(before));
}
// Two-level search: the solver keeps every (1 << shift)-th row of its species in LDS (sample j = row
// row0 + (j << shift)); the samples narrow [lo,hi) to less than one stride without touching memory, the
// remainder is one more (usually single) round in global memory.  Same result as the plain search.
struct RowIdx {
    const double *a;      // all rows (global)
    const double *idx;    // samples (LDS)
    uint32_t row0, shift, n_idx;
    // optional LDS copy of rows [c_row0, c_row0 + c_n) (the line search caches each pattern's remaining candidates)
    const double *cache;
    uint32_t c_row0, c_n;
};
__device__ __forceinline__ double row_val(const RowIdx &r, uint32_t row) {
    const uint32_t off = row - r.c_row0;
    return off < r.c_n ? r.cache[off] : r.a[row];
}
// samples only, no memory access: rmin <= (first row of [lo,hi) that is not "before" v) <= rmax
__device__ __forceinline__ void bound_idx_range(bool coop, const RowIdx &r, uint32_t lo, uint32_t hi, double v, bool upper, uint32_t &rmin,
                                                uint32_t &rmax) {
    rmin = lo; rmax = hi;
    if (hi <= lo) return;
    const uint32_t mask = (1u << r.shift) - 1u;
    const uint32_t j0 = (lo - r.row0 + mask) >> r.shift, j1 = (hi - r.row0 + mask) >> r.shift;
    if (j1 > j0) {
        const uint32_t jj = coop ? wave_bound(r.idx, j0, j1, v, upper)
                                 : (upper ? upper_bound_a(r.idx, j0, j1, v) : lower_bound_a(r.idx, j0, j1, v));
        if (jj > j0) rmin = r.row0 + ((jj - 1) << r.shift) + 1;
        if (jj < j1) rmax = r.row0 + (jj << r.shift);
    }
}
__device__ __forceinline__ uint32_t bound_idx(bool coop, const RowIdx &r, uint32_t lo, uint32_t hi, double v, bool upper) {
    if (r.c_n && lo >= r.c_row0 && hi <= r.c_row0 + r.c_n && hi >= lo) {   // entirely inside the cached window
        const uint32_t l2 = lo - r.c_row0, h2 = hi - r.c_row0;
        const uint32_t res = coop ? wave_bound(r.cache, l2, h2, v, upper)
                                  : (upper ? upper_bound_a(r.cache, l2, h2, v) : lower_bound_a(r.cache, l2, h2, v));
        return r.c_row0 + res;
    }
    if (hi - lo > 8) {
        const uint32_t mask = (1u << r.shift) - 1u;
        const uint32_t j0 = (lo - r.row0 + mask) >> r.shift, j1 = (hi - r.row0 + mask) >> r.shift;   // samples with lo <= row < hi
        if (j1 > j0) {
            const uint32_t jj = coop ? wave_bound(r.idx, j0, j1, v, upper)
                                     : (upper ? upper_bound_a(r.idx, j0, j1, v) : lower_bound_a(r.idx, j0, j1, v));
            if (jj > j0) lo = r.row0 + ((jj - 1) << r.shift) + 1;
            if (jj < j1) hi = r.row0 + (jj << r.shift);
        }
    }
    if (coop) return wave_bound(r.a, lo, hi, v, upper);
    return upper ? upper_bound_a(r.a, lo, hi, v) : lower_bound_a(r.a, lo, hi, v);
}
__device__ __forceinline__ uint32_t lb(bool coop, const RowIdx &r, uint32_t lo, uint32_t hi, double v) { return bound_idx(coop, r, lo, hi, v, false); }
__device__ __forceinline__ uint32_t ub_(bool coop, const RowIdx &r, uint32_t lo, uint32_t hi, double v) { return bound_idx(coop, r, lo, hi, v, true); }

__device__ __forceinline__ double mdot(uint64_t m, const double *x) {   // ascending-bit order
    double s = 0.0;
    while (m) { int j = __ffsll((long long)m) - 1; s += x[j]; m &= m - 1; }
    return s;
}
template <int NW>
__device__ __forceinline__ double mdotw(const uint64_t *mw, const double *x) {   // ascending-bit order over NW mask words
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { uint64_t m = mw[w]; while (m) { int j = __ffsll((long long)m) - 1; s += x[64 * w + j]; m &= m - 1; } }
    return s;
}
// s += x[base + j] over the set bits j of m in ascending order, FOUR loads in flight (the additions keep their order: same bits as one at a time)
__device__ __forceinline__ void mdot_word4(uint64_t m, const double *x, int base, double &s) {
    while (m) {
        int j[4]; bool on[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { on[r] = m != 0ull; j[r] = on[r] ? __ffsll((long long)m) - 1 : 0; m &= m - 1; }   // (0 & anything = 0: an empty m stays empty)
        double t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = x[base + j[r]];
#pragma unroll
        for (int r = 0; r < 4; ++r) if (on[r]) s += t[r];
    }
}
template <int NW>   // NW == 0: nw words, a run-time number
__device__ __forceinline__ double mdotx(const uint64_t *mw, int nw, const double *x) {
    if constexpr (NW != 0) return mdotw<NW>(mw, x);
    else {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) mdot_word4(mw[w], x, 64 * w, s);
        return s;
    }
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------------
// a9: per-hap unique-trio statistics, BY KEY (round 5).  The rows of the index are numbered in filing order -- node after node --, so
// the rows of one haplotype are scattered over its species' block; every row carries its owner (d_trio_hap).  The block of a species
// is cut into chunks of rows; ONE WAVE takes a chunk and keeps an accumulator per haplotype of the species (in LDS; in the chunk's own
// row of the partials for a species of more than HS_LDS_HAPS haplotypes): per 64 rows it walks the distinct owners among its lanes --
// neighbouring rows are windows around the same private allele, a handful of owners -- and adds each owner's lanes by a DPP reduction
// in fixed lane order.  A chunk's partials are then added in chunk order by one wave per species.  Every sum has a fixed order: same bits
// on every run (the reference's own order is that of a hash set).  Three passes like zscore_filter (profile.rs:1028-1051): (sum, count)
// of the non-zero abundances -> mean; squared deviations -> sd; (sum, count) of |z| < 3 -> the filtered mean.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t HS_CHUNK_ROWS = 1024, HS_LDS_HAPS = 1024;
constexpr int HS_SLAB = 16;   // haplotypes whose accumulators a lane keeps in registers at a time
struct HapAcc { double a; uint32_t c, n; };   // sum, count of the pass, rows seen (pass 0)

template <int PASS>
__global__ void __launch_bounds__(64) hap_rows_pass_kernel(const uint4 *__restrict__ chunks, const uint64_t *__restrict__ hap_off, const uint16_t *__restrict__ row_hap,
                                                           unsigned long long *tb /* read, and -- clean != 0, pass 0 -- zeroed behind the read */, const trio_len_t *__restrict__ tlen,
                                                           const double *__restrict__ mean0, const double *__restrict__ sd, HapAcc *__restrict__ part,
                                                           double *__restrict__ cx, uint16_t *__restrict__ chh, uint32_t *__restrict__ cn, uint32_t clean,
                                                           const uint8_t *__restrict__ active) {
    extern __shared__ HapAcc s_hap_acc[];
    __shared__ uint32_t s_qrow[128];
    __shared__ unsigned long long s_qtb[128];
    const uint4 ch = chunks[blockIdx.x];                       // {species, first row, end row, first partial}
    const uint32_t h0 = (uint32_t)hap_off[ch.x], Hs = (uint32_t)hap_off[ch.x + 1] - h0;
    const int lane = threadIdx.x;
    // a species the species level dropped: the coverage pass skipped its reads, its rows' abundances are all zero -- the partials of an empty chunk, nothing read
    if (active != nullptr && active[ch.x] == 0) {              // (chunk-uniform)
        for (uint32_t h = (uint32_t)lane; h < Hs; h += 64) part[ch.w + h] = HapAcc{0.0, 0u, 0u};
        if (PASS == 0 && lane == 0) cn[blockIdx.x] = 0u;
        return;
    }
    // Only rows with a NON-ZERO abundance count in any of the three statistics (profile.rs:1129-1133: `> 0.0`), and most rows are zero (the strains that
    // are not in the sample; a fifth of the rows at the BASELINE configurations).  Pass 0 reads the abundances of all rows (8 bytes each), QUEUES the
    // non-zero ones in LDS and handles them 64 at a time on dense lanes: length and owner are gathered, the f64 division is done, and {owner, value} go
    // to the chunk's stretch of a compacted copy -- passes 1 and 2 read that copy alone.  Before: three passes over {8, 4, 2} bytes of every row with a
    // division per row and pass, bound by VALU issue (184 wave-instructions per 64 rows: 0.8 ms a pass at 1e4 strains).
    // Up to 64 haplotypes per species (every species of the BASELINE configurations): every LANE keeps its own accumulators for a slab of
    // HS_SLAB haplotypes in registers and adds its rows to them by compare-and-select -- no cross-lane traffic and no scalar round trip inside
    // the loop over the rows; the lanes meet once per chunk and slab (DPP reductions, fixed order).  Version 1 of this kernel walked the distinct
    // owners of every 64 rows (readlane -> ballot -> DPP reduction -> owner's lane adds): ~150 cycles of scalar / vector ping-pong per owner,
    // 2.0 ms a pass at 1e4 strains whether the accumulators sat in LDS or in registers (round 4's kernel over contiguous rows: 0.85 ms for
    // all three).  A species of 17 .. 64 haplotypes reads its compacted rows once per slab.  Mean and sd of pass 0 / 1 ride in lane h and reach
    // a row's lane by one bpermute.
    const bool in_reg = Hs <= 64u;
    const bool in_lds = !in_reg && Hs <= HS_LDS_HAPS;
    HapAcc *acc = in_lds ? s_hap_acc : part + ch.w;            // (more than 64 haplotypes: LDS; beyond HS_LDS_HAPS the chunk's own, zero-filled row of partials)
    if (in_lds) for (uint32_t h = lane; h < Hs; h += 64) acc[h] = HapAcc{0.0, 0u, 0u};
    __syncthreads();
    double my_mean = 0.0, my_sd = 0.0;
    if (in_reg && PASS >= 1 && (uint32_t)lane < Hs) { my_mean = mean0[h0 + lane]; if (PASS == 2) my_sd = sd[h0 + lane]; }
    uint32_t n_c = PASS == 0 ? 0u : cn[blockIdx.x];           // non-zero rows of the chunk = entries of its compacted stretch [ch.y, ch.y + n_c)
    // what a compacted entry {h, x > 0} adds in this pass (all lanes come here: the shuffles)
    auto pass_value = [&](uint32_t h, double x, bool valid, double &val, bool &flag) {
        val = 0.0; flag = false;
        if (PASS == 0) { if (valid) { val = x; flag = true; } return; }             // :1129-1133
        double m, s_ = 0.0;
        if (in_reg) { m = __shfl(my_mean, (int)(h & 63u)); if (PASS == 2) s_ = __shfl(my_sd, (int)(h & 63u)); }
        else { m = valid ? mean0[h0 + h] : 0.0; if (PASS == 2) s_ = valid ? sd[h0 + h] : 0.0; }
        if (valid) {
            if (PASS == 1) { val = (x - m) * (x - m); flag = true; }
            else if (s_ != 0.0 && fabs((x - m) / s_) < 3.0) { val = x; flag = true; }   // :1043-1050
        }
    };
    // pass 0: the chunk's rows -> its compacted stretch, every dense batch of up to 64 entries handed to `sink` on the way
    auto compact_rows = [&](auto &&sink) {
        uint32_t qh = 0, qn = 0, nw = 0;                       // queue head, entries queued, entries written (wave-uniform)
        auto drain = [&](uint32_t nb) {
            const bool v = (uint32_t)lane < nb;
            const uint32_t row = s_qrow[(qh + (uint32_t)lane) & 127u];
            const unsigned long long t = s_qtb[(qh + (uint32_t)lane) & 127u];
            uint32_t h = 0xFFFFFFFFu;
            double x = 0.0;
            if (v) {
#if TRIO_LH_PACK
                const uint2 lh = tlen[row];
                h = lh.y;
                x = (double)(long long)t / (double)lh.x;                       // profile.rs:1013-1014
#else
                h = row_hap[row];
                x = (double)(long long)t / (double)tlen[row];                  // profile.rs:1013-1014
#endif
                cx[ch.y + nw + (uint32_t)lane] = x; chh[ch.y + nw + (uint32_t)lane] = (uint16_t)h;
            }
            sink(h, x, v);
            qh = (qh + nb) & 127u; qn -= nb; nw += nb;
        };
#ifndef HS_TB_AHEAD
#define HS_TB_AHEAD 4
#endif
        constexpr int TA = HS_TB_AHEAD;                        // stretches of 64 rows whose abundances are requested together (one at a time: a round trip per stretch)
        for (uint32_t rb = ch.y; rb < ch.z; rb += 64u * TA) {
            unsigned long long tq[TA];
#pragma unroll
            for (int q = 0; q < TA; ++q) { const uint32_t row = rb + 64u * (uint32_t)q + (uint32_t)lane; tq[q] = row < ch.z ? tb[row] : 0ull; }
#pragma unroll
            for (int q = 0; q < TA; ++q) {
                const uint32_t r0 = rb + 64u * (uint32_t)q;
                if (r0 >= ch.z) break;                             // (chunk-uniform)
                const uint32_t row = r0 + (uint32_t)lane;
                const unsigned long long t = tq[q];
                const bool nz = (long long)t > 0;
                // the resident step: pass 0 is the only reader of the coverage pass's trio_bases -- it leaves them zeroed for the next step's pass (round 6)
                if (PASS == 0 && clean && t != 0ull) tb[row] = 0ull;
                const unsigned long long bal = __ballot(nz);
                if (nz) {
                    const uint32_t idx = (qh + qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))) & 127u;
                    s_qrow[idx] = row; s_qtb[idx] = t;
                }
                qn += (uint32_t)__popcll(bal);
                if (qn >= 64u) drain(64u);
            }
        }
        if (qn) drain(qn);
        n_c = nw;
    };
#ifndef HS_NO_TRANSPOSE
    if (in_reg && Hs > (uint32_t)HS_SLAB) {                   // (up to 16 haplotypes the one slab below is faster: 1.02 against 1.46 ms at ten)
        // 17 .. 64 haplotypes (round 6): LANE h owns haplotype h.  The entries that count are handed round one by one (two readlanes for the value, one for the
        // owner: scalar broadcasts) and the owner's lane adds -- in entry order, a fixed order of additions; no slabs that read the compacted rows again, no
        // reductions at the end.  (-DHS_NO_TRANSPOSE: the slabs of 16 below, as up to 16 haplotypes.)
        double acc_t = 0.0;
        uint32_t cnt_t = 0;
        auto sink_t = [&](uint32_t h, double val, bool flag) {
            unsigned long long todo = __ballot(flag);
            while (todo) {
                const int e = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                todo &= todo - 1ull;
                const uint32_t he = (uint32_t)__builtin_amdgcn_readlane((int)h, e);
                const double ve = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(val), e), __builtin_amdgcn_readlane(__double2loint(val), e));
                if ((uint32_t)lane == he) { acc_t += ve; ++cnt_t; }
            }
        };
        if (PASS == 0) compact_rows([&](uint32_t h, double x, bool v) { sink_t(h, v ? x : 0.0, v); });
        else
            for (uint32_t r0 = 0; r0 < n_c; r0 += 64) {
                const uint32_t i = r0 + (uint32_t)lane;
                const bool v = i < n_c;
                const uint32_t h = v ? (uint32_t)chh[ch.y + i] : 0xFFFFFFFFu;
                const double x = v ? cx[ch.y + i] : 0.0;
                double val; bool flag;
                pass_value(h, x, v, val, flag);
                sink_t(h, val, flag);
            }
        if ((uint32_t)lane < Hs) part[ch.w + (uint32_t)lane] = HapAcc{acc_t, cnt_t, 0u};
        if (PASS == 0 && lane == 0) cn[blockIdx.x] = n_c;
        return;
    }
#endif
    if (in_reg) {
        for (uint32_t slab = 0; slab * HS_SLAB < Hs; ++slab) {
            double a_[HS_SLAB];
            uint32_t c_[HS_SLAB];
#pragma unroll
            for (int k = 0; k < HS_SLAB; ++k) { a_[k] = 0.0; c_[k] = 0u; }
            auto sink_reg = [&](uint32_t h, double val, bool flag) {
                const uint32_t j = h - slab * HS_SLAB;                         // (a lane without an entry: no slab holds it)
#pragma unroll
                for (int k = 0; k < HS_SLAB; ++k) {
                    if (slab * HS_SLAB + (uint32_t)k >= Hs) break;                // (chunk-uniform) haplotypes the species does not have: 2.09 -> 1.62 ms at the reference-DB shape
                    const bool m_ = j == (uint32_t)k;
                    a_[k] += m_ ? val : 0.0;
                    c_[k] += (m_ && flag) ? 1u : 0u;
                }
            };
            if (PASS == 0 && slab == 0) compact_rows([&](uint32_t h, double x, bool v) { sink_reg(h, v ? x : 0.0, v); });
            else {
                // the compacted stretch was written by this very wave, entry i by the lane that reads it back: ordering within the wave is all that is
                // needed.  (Until round 6 a __threadfence() stood here: agent scope = write-back + invalidate of the XCD's L2 on gfx950, by every chunk's
                // wave -- species of more than HS_SLAB haplotypes paid 0.84 ms for pass 0 at 125 x 50 strains where 1000 x 10 strains paid 0.53.)
                if (PASS == 0 && slab == 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                for (uint32_t r0 = 0; r0 < n_c; r0 += 64) {
                    const uint32_t i = r0 + (uint32_t)lane;
                    const bool v = i < n_c;
                    const uint32_t h = v ? (uint32_t)chh[ch.y + i] : 0xFFFFFFFFu;
                    const double x = v ? cx[ch.y + i] : 0.0;
                    double val; bool flag;
                    pass_value(h, x, v, val, flag);
                    sink_reg(h, val, flag);
                }
            }
#pragma unroll
            for (int k = 0; k < HS_SLAB; ++k) {
                if (slab * HS_SLAB + (uint32_t)k >= Hs) break;
                const double v = wave_reduce(a_[k], [](double x, double y) { return x + y; });
                const uint32_t c = wave_reduce(c_[k], [](uint32_t x, uint32_t y) { return x + y; });
                const uint32_t hh = slab * HS_SLAB + (uint32_t)k;
                if (lane == 0 && hh < Hs) part[ch.w + hh] = HapAcc{v, c, 0u};
            }
        }
        if (PASS == 0 && lane == 0) cn[blockIdx.x] = n_c;
        return;
    }
    auto sink_gen = [&](uint32_t h, double val, bool flag, bool valid) {
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const uint32_t hh = (uint32_t)__builtin_amdgcn_readlane((int)h, __builtin_ctzll(todo));
            const bool mine = valid && h == hh;
            const unsigned long long sel = __ballot(mine);
            const double v = wave_reduce(mine ? val : 0.0, [](double a2, double b2) { return a2 + b2; });
            const uint32_t c = (uint32_t)__popcll(__ballot(mine && flag));
            if (lane == 0) { HapAcc t = acc[hh]; t.a += v; t.c += c; t.n += (uint32_t)__popcll(sel); acc[hh] = t; }
            todo &= ~sel;
        }
    };
    if (PASS == 0) compact_rows([&](uint32_t h, double x, bool v) { sink_gen(h, v ? x : 0.0, v, v); });
    else
        for (uint32_t r0 = 0; r0 < n_c; r0 += 64) {
            const uint32_t i = r0 + (uint32_t)lane;
            const bool v = i < n_c;
            const uint32_t h = v ? (uint32_t)chh[ch.y + i] : 0xFFFFFFFFu;
            const double x = v ? cx[ch.y + i] : 0.0;
            double val; bool flag;
            pass_value(h, x, v, val, flag);
            sink_gen(h, val, flag, v);
        }
    __syncthreads();
    if (in_lds) for (uint32_t h = lane; h < Hs; h += 64) part[ch.w + h] = acc[h];
    if (PASS == 0 && lane == 0) cn[blockIdx.x] = n_c;
}
template <int PASS>
__global__ void __launch_bounds__(256) hap_combine_kernel(const uint32_t *__restrict__ sp_chunk_off, const uint4 *__restrict__ chunks, const uint64_t *__restrict__ hap_off,
                                                          const HapAcc *__restrict__ part, uint32_t *__restrict__ nnz, double *__restrict__ mean0, double *__restrict__ sd,
                                                          double *__restrict__ meanf) {
    // one workgroup per species; the chunks' partials of a haplotype are summed by `parts` threads (chunk c by thread c mod parts, in chunk order), the
    // parts then in part order: a fixed order of additions, whatever the launch (same bits every run)
    __shared__ double s_a[256];
    __shared__ unsigned long long s_c[256];
    const uint32_t s = blockIdx.x, c0 = sp_chunk_off[s], c1 = sp_chunk_off[s + 1];
    const uint32_t h0 = (uint32_t)hap_off[s], Hs = (uint32_t)hap_off[s + 1] - h0;
    uint32_t width = 256;                                  // threads side by side over the haplotypes: the power of two >= Hs, at most 256
    if (Hs <= 128u) { width = 8; while (width < Hs) width <<= 1; }
    const uint32_t parts = 256u / width, hl = threadIdx.x % width, pt = threadIdx.x / width;
    for (uint32_t hb = 0; hb < Hs; hb += width) {
        const uint32_t h = hb + hl;
        double a = 0.0;
        unsigned long long c = 0;
        if (h < Hs) for (uint32_t k = c0 + pt; k < c1; k += parts) { const HapAcc p = part[chunks[k].w + h]; a += p.a; c += p.c; }
        s_a[threadIdx.x] = a; s_c[threadIdx.x] = c;
        __syncthreads();
        if (pt == 0 && h < Hs) {
            for (uint32_t q = 1; q < parts; ++q) { a += s_a[q * width + hl]; c += s_c[q * width + hl]; }
            if (PASS == 0) { nnz[h0 + h] = (uint32_t)c; mean0[h0 + h] = c ? a / (double)c : 0.0; }              // profile.rs:1037
            else if (PASS == 1) { const double n = (double)nnz[h0 + h]; sd[h0 + h] = n > 0 ? sqrt(a / n) : 0.0; }   // :1038-1041
            else meanf[h0 + h] = c ? a / (double)c : 0.0;                 // sd == 0 -> empty -> 0.0 (:1043-1045, :1143-1147)
        }
        __syncthreads();
    }
}

// first build of a db (trio_index_build): the blocks of rows of the species -> chunks of rows, a row of partials per chunk
int hap_stats_layout(Ctx *ctx, Db *db, const uint64_t *sp_first_row, const uint64_t *sp_rows) {
    const uint32_t S = db->S;
    // rows per chunk (= per wave): 1024 where that gives a few thousand chunks; small dbs take shorter chunks, down to 128 rows, so that the pass has
    // waves for every CU (one species x 10 strains: 176 chunks of 1024 rows ran as 176 waves, 0.05 ms a pass)
    uint64_t total_rows = 0;
    for (uint32_t s = 0; s < S; ++s) total_rows += sp_rows[s];
    const uint64_t chunk_rows = std::min<uint64_t>(HS_CHUNK_ROWS, std::max<uint64_t>(128, ((total_rows / 4096 + 63) / 64) * 64));
    std::vector<uint4> chunks;
    std::vector<uint32_t> sp_off(S + 1, 0);
    uint64_t n_part = 0;
    uint32_t lds_haps = 1;
    bool global_rows = false;
    for (uint32_t s = 0; s < S; ++s) {
        sp_off[s] = (uint32_t)chunks.size();
        const uint64_t Hs = db->h_hap_off[s + 1] - db->h_hap_off[s];
        if (Hs > 64 && Hs <= HS_LDS_HAPS) lds_haps = std::max<uint32_t>(lds_haps, (uint32_t)Hs);
        if (Hs > HS_LDS_HAPS && sp_rows[s]) global_rows = true;
        // a chunk holds at least eight rows per haplotype of its species: the partials stay an eighth of the rows at most
        const uint64_t per = std::max<uint64_t>(chunk_rows, ((8 * Hs + 63) / 64) * 64);
        for (uint64_t r = 0; r < sp_rows[s]; r += per) {
            if (n_part + Hs >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "hap statistics: more than 2^32 chunk partials");
            chunks.push_back(make_uint4(s, (uint32_t)(sp_first_row[s] + r), (uint32_t)(sp_first_row[s] + std::min<uint64_t>(sp_rows[s], r + per)), (uint32_t)n_part));
            n_part += Hs;
        }
    }
    sp_off[S] = (uint32_t)chunks.size();
    db->n_stat_chunks = (uint32_t)chunks.size();
    db->n_stat_partials = n_part;
    db->stat_lds_haps = lds_haps;
    db->stat_global_rows = global_rows;
    if (chunks.empty()) chunks.push_back(make_uint4(0u, 0u, 0u, 0u));
    PTX_TRY(upload(ctx, db->d_stat_chunks, chunks.data(), chunks.size()));
    PTX_TRY(upload(ctx, db->d_sp_chunk_off, sp_off.data(), sp_off.size()));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging vectors go out of scope
    return 0;
}

int hap_trio_stats_launch(Ctx *ctx, const Db *db, DevBuf<uint32_t> &d_nnz, DevBuf<double> &d_mean, const uint8_t *d_active) {
    if (ctx->cfg.no_absent_skip) d_active = nullptr;
    PTX_HIP(ctx, d_nnz.alloc(db->H));
    PTX_HIP(ctx, d_mean.alloc(db->H));
    if (db->H == 0) return 0;
    Db *dbm = const_cast<Db *>(db);
    const uint32_t S = db->S, NC = db->n_stat_chunks;
    const uint64_t H = db->H;
    PTX_HIP(ctx, dbm->d_hap_part.alloc(2 * (size_t)std::max<uint64_t>(db->n_stat_partials, 1) + 2 * H));   // the chunks' partials (16 B each), then mean and sd of pass 0 / 1
    HapAcc *part = reinterpret_cast<HapAcc *>(dbm->d_hap_part.p);
    double *mean0 = dbm->d_hap_part.p + 2 * (size_t)std::max<uint64_t>(db->n_stat_partials, 1), *sd = mean0 + H;
    const size_t lds = (size_t)db->stat_lds_haps * sizeof(HapAcc);
    // the compacted copy of the non-zero rows {value, owner}, chunk by chunk in place of the chunk's rows, and its length per chunk
    PTX_HIP(ctx, dbm->d_hs_x.alloc(std::max<uint64_t>(db->U, 1))); PTX_HIP(ctx, dbm->d_hs_h.alloc(std::max<uint64_t>(db->U, 1))); PTX_HIP(ctx, dbm->d_hs_n.alloc(std::max<uint32_t>(NC, 1)));
    KTimer t(ctx, "hap_rows_pass_kernel");
#define HS_PASS(PP)                                                                                                                                            \
    if (db->stat_global_rows) PTX_TRY(zero_fill(ctx, part, (size_t)std::max<uint64_t>(db->n_stat_partials, 1) * sizeof(HapAcc)));                            \
    if (NC) hipLaunchKernelGGL(hap_rows_pass_kernel<PP>, dim3(NC), dim3(64), lds, ctx->stream, (const uint4 *)db->d_stat_chunks.p, (const uint64_t *)db->d_hap_off.p, \
                               TRIO_HAP_PTR(db), (unsigned long long *)db->d_trio_bases.p, (const trio_len_t *)db->d_trio_len.p,          \
                               (const double *)mean0, (const double *)sd, part, dbm->d_hs_x.p, dbm->d_hs_h.p, dbm->d_hs_n.p, db->cov_self_clean ? 1u : 0u, d_active);  \
    hipLaunchKernelGGL(hap_combine_kernel<PP>, dim3(S), dim3(256), 0, ctx->stream, (const uint32_t *)db->d_sp_chunk_off.p, (const uint4 *)db->d_stat_chunks.p,  \
                       (const uint64_t *)db->d_hap_off.p, (const HapAcc *)part, d_nnz.p, mean0, sd, d_mean.p);
    HS_PASS(0) HS_PASS(1) HS_PASS(2)
#undef HS_PASS
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// node abundance + per-species statistics
// ---------------------------------------------------------------------------------------------
constexpr int STAT_CHUNKS = 256;  // most workgroups per species; partials are combined in fixed order (deterministic)
// chunks per species actually used: ~2048 workgroups in all (one species: 256 chunks; a hundred species: 20)
static inline uint32_t stat_chunks(uint32_t S, uint32_t target = 2048u) { uint32_t c = target / (S ? S : 1u); return c < 1u ? 1u : (c > (uint32_t)STAT_CHUNKS ? (uint32_t)STAT_CHUNKS : c); }
struct NodePartial { double mx, zs; unsigned long long nv, zc; };

__global__ void __launch_bounds__(256) node_stats_kernel(const uint32_t *__restrict__ node_base, const uint32_t *__restrict__ node_len,
                                                         const unsigned long long *__restrict__ bases, double min_depth,
                                                         double *__restrict__ ab_out, NodePartial *__restrict__ part,
                                                         const uint32_t *__restrict__ chunk_sp, const uint32_t *__restrict__ sp_chunk_off) {
    __shared__ double red[4];
    __shared__ unsigned long long redu[4];
    // chunks by SIZE (round 6): a species takes chunks in proportion to its nodes -- with one workgroup per species (what an even split gave a db of
    // thousands of species) the 3e5-node graphs of the multi-strain species ran beside 5e3-node chunk graphs: 12.4 ms at the reference-DB shape
    const uint32_t s = chunk_sp[blockIdx.x], nch = sp_chunk_off[s + 1] - sp_chunk_off[s], ch = blockIdx.x - sp_chunk_off[s];
    const uint32_t b = node_base[s], e = node_base[s + 1];
    const uint32_t per = (e - b + nch - 1) / nch;
    uint32_t lo = b + ch * per, hi = lo + per;
    if (hi > e) hi = e;
    double mx = -INFINITY, zs = 0.0;
    unsigned long long nv = 0, zc = 0;
    for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
        double len = (double)node_len[v];                        // (the 4-byte copy of the lengths: 4V instead of 8V of offsets)
        double ab = (double)(long long)bases[v] / len;           // profile.rs:987-988
        ab_out[v] = ab;
        mx = fmax(mx, ab);
        if (ab > 0.0) ++nv;
        double o = ab > min_depth ? ab : 0.0;                    // :2941-2944
        if (o > 0.0) { zs += o; ++zc; }
    }
    mx = block_max_f64<256>(mx, red);
    zs = block_sum_f64<256>(zs, red);
    nv = block_sum_u64<256>(nv, redu);
    zc = block_sum_u64<256>(zc, redu);
    if (threadIdx.x == 0) part[blockIdx.x] = {mx, zs, nv, zc};
}
// The same pass with the covered-base count of every node folded in (node_base_cov, profile.rs:844/874, :1018-1023 -- popcount_kernel's
// work, stage_cov.hip): in the resident step nothing reads the counts between the coverage pass and this one, and the two passes share
// the node lengths.  The bit offset of a node comes from a running prefix of the lengths inside the workgroup's range (one 8-byte load
// per workgroup instead of 8V bytes of offsets); 24V + L/8 bytes instead of 32V + L/8 for the two kernels.
// CLEAN (round 6): this pass is the LAST reader of `bases`, the bit vector and the full-node flags in the resident step -- it leaves them zeroed for
// the next step's coverage pass (only what is not zero is written: the lines are in the caches, a node some step covered whole has no marked bits),
// instead of a 4-GB zero fill per step in front of it.  A word of flags / bits that a wave shares with its neighbours (the ends of its range of nodes)
// loses this wave's bits only, atomically; a word that is all its own is stored.  (No __restrict__ on the three arrays: they are read and written here.)
// LONGN (round 6): graphs of LONG nodes -- a single-genome species is a chain of 1024-bp chunks (build_eq1.rs:26-36), 32 bitmap words per node.  The
// per-lane loop over a node's interior words walks 64 different cache lines per iteration (12.4 ms at the reference-DB shape).  Instead the wave reads
// the words of its whole 64-node stretch coalesced, keeps the running count of set bits in front of every word in LDS (a DPP prefix sum per 64 words),
// and a node's covered bases are the difference of that prefix at its two ends -- two LDS reads per node, whatever its length.
constexpr uint32_t NCS_PWORDS = 2304;   // words of one stretch the prefix holds (64 nodes x 1152 bases); a longer stretch takes the per-lane loop
extern __shared__ __attribute__((aligned(16))) uint32_t s_ncs_prefix[];
template <bool CLEAN, bool LONGN = false>
__global__ void __launch_bounds__(256) node_cov_stats_kernel(const uint32_t *__restrict__ node_base, const uint32_t *__restrict__ node_len,
                                                             unsigned long long *bases, const uint64_t *__restrict__ bit_off,
                                                             uint32_t *full, uint32_t *bitmap, double min_depth,
                                                             uint32_t *__restrict__ cov_out, double *__restrict__ ab_out, NodePartial *__restrict__ part,
                                                             const uint32_t *__restrict__ chunk_sp, const uint32_t *__restrict__ sp_chunk_off,
                                                             const uint8_t *__restrict__ active) {
    __shared__ double red[4];
    __shared__ unsigned long long redu[4];
    const uint32_t s = chunk_sp[blockIdx.x], nch = sp_chunk_off[s + 1] - sp_chunk_off[s], ch = blockIdx.x - sp_chunk_off[s];   // (chunks by size: node_stats_kernel)
    const uint32_t b = node_base[s], e = node_base[s + 1];
    const uint32_t per = (e - b + nch - 1) / nch;
    uint32_t lo = b + ch * per, hi = lo + per;
    if (hi > e) hi = e;
    // A species the species level dropped (round 6): the coverage pass skipped its reads (the same flags), so its part of the arena is all zero and what this
    // pass would compute from it is known -- zeros, written without reading anything.  The work follows the species that are PRESENT in the sample, not the
    // size of the resident DB (the reference-DB shape, four fifths of the single-genome species absent: this pass 4.97 -> 3.92 ms, the step 25.4 -> 23.2).
    if (active != nullptr && active[s] == 0) {                   // (workgroup-uniform)
        for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) { cov_out[v] = 0u; ab_out[v] = 0.0; }
        if (threadIdx.x == 0) part[blockIdx.x] = {hi > lo ? 0.0 : -INFINITY, 0.0, 0ull, 0ull};
        return;
    }
    double mx = -INFINITY, zs = 0.0;
    unsigned long long nv = 0, zc = 0;
    // every WAVE walks its own quarter of the workgroup's range with its own running bit offset: no LDS, no barrier in the loop --
    // the waves of a CU hide each other's two dependent loads (lengths -> bitmap words)
#ifndef NCS_NR
#define NCS_NR 4
#endif
    constexpr int NR = NCS_NR;                                   // 64-node stretches per round: their loads are in flight together (-DNCS_NR: measurement builds)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t quarter = ((hi > lo ? hi - lo : 0u) + 3u) / 4u;
    const uint32_t wlo = min(hi, lo + wave * quarter), whi = min(hi, wlo + quarter);
    uint64_t run = wlo < whi ? bit_off[wlo] : 0ull;              // bit offset of the first node of the coming round
    // (requesting the three streams of round r + 1 at the top of round r -- what took a dependent level off the coverage kernel's chain -- LOST here:
    // 2.55 -> 3.17 ms at 1e4 strains, 16 more registers for a kernel whose rounds are already four stretches deep)
    for (uint32_t v0 = wlo; v0 < whi; v0 += 64 * NR) {
        uint32_t l[NR], fw[NR];
        unsigned long long bs[NR];
        uint64_t g0[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint32_t v = v0 + (uint32_t)r * 64u + lane;
            const bool in = v < whi;
            l[r] = in ? node_len[v] : 0u;
            bs[r] = in ? bases[v] : 0ull;
            fw[r] = in ? full[v >> 5] : 0u;
        }
        const uint64_t round_b0 = run;                           // the bits of this round's nodes: [round_b0, run) once the lengths are summed
        uint64_t sb[NR + 1];                                     // first bit of every stretch (wave-uniform)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint32_t incl = wave_incl_scan_dpp(l[r]);      // (a species' bases fit 32 bits: checked at upload)
            sb[r] = run;
            g0[r] = run + incl - l[r];
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        sb[NR] = run;
        uint32_t bw0[NR], bw1[NR];                               // first and last bitmap word of every node: independent loads, issued together
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool any = l[r] != 0u;                         // (l = 0 outside the range)
            bw0[r] = any ? bitmap[g0[r] >> 5] : 0u;
            bw1[r] = any ? bitmap[(g0[r] + l[r] - 1) >> 5] : 0u;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint32_t v = v0 + (uint32_t)r * 64u + lane;
            bool coop = false;                                   // (wave-uniform) this stretch's counts come from the prefix in LDS
            uint64_t wa = 0;
            if constexpr (LONGN) {
                const uint64_t b0 = sb[r], b1 = sb[r + 1];
                const bool has_long = __builtin_amdgcn_ballot_w64(l[r] > 64u) != 0ull;
                wa = (b0 >> 5) & ~3ull;                                  // (from a 16-byte boundary: four words per lane and load; the tail read beyond the stretch is inside the arena)
                const uint64_t nw = b1 > b0 ? ((b1 - 1) >> 5) - wa + 1 : 0;
                coop = has_long && nw <= (uint64_t)NCS_PWORDS;
                if (coop) {
                    uint32_t *pw = s_ncs_prefix + wave * NCS_PWORDS;
                    uint32_t carry = 0;
                    for (uint32_t k = 0; k < (uint32_t)nw; k += 256) {
                        const uint32_t i = k + 4u * lane;
                        const uint4 x = i < (uint32_t)nw ? *reinterpret_cast<const uint4 *>(bitmap + wa + i) : make_uint4(0u, 0u, 0u, 0u);
                        const uint32_t p0 = (uint32_t)__popc(x.x), p1 = p0 + (uint32_t)__popc(x.y), p2 = p1 + (uint32_t)__popc(x.z), p3 = p2 + (uint32_t)__popc(x.w);
                        const uint32_t incl = wave_incl_scan_dpp(p3);
                        const uint32_t base = carry + incl - p3;              // set bits in front of this lane's four words
                        if (i < (uint32_t)nw) *reinterpret_cast<uint4 *>(pw + i) = make_uint4(base, base + p0, base + p1, base + p2);
                        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                }
            }
            if (v >= whi) continue;
            uint32_t c = 0;
            if (LONGN && coop) {
                if (l[r]) {
                    const uint64_t g1 = g0[r] + l[r], w0 = g0[r] >> 5, w1 = (g1 - 1) >> 5;
                    const uint32_t below0 = (1u << (g0[r] & 31)) - 1u, m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
                    const uint32_t *pw = s_ncs_prefix + wave * NCS_PWORDS;
                    c = (pw[w1 - wa] + (uint32_t)__popc(bw1[r] & m1)) - (pw[w0 - wa] + (uint32_t)__popc(bw0[r] & below0));
                }
            } else
            if (l[r]) {
                const uint64_t g1 = g0[r] + l[r], w0 = g0[r] >> 5, w1 = (g1 - 1) >> 5;
                const uint32_t m0 = 0xFFFFFFFFu << (g0[r] & 31), m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
                c = w0 == w1 ? __popc(bw0[r] & m0 & m1) : __popc(bw0[r] & m0) + __popc(bw1[r] & m1);
                for (uint64_t w = w0 + 1; w < w1; ++w) {                           // nodes of more than 33 bases
                    const uint32_t x = bitmap[w];
                    c += __popc(x);
                    if (CLEAN && x) bitmap[w] = 0u;                                // (a word inside one node is that node's alone)
                }
                if constexpr (CLEAN) {
                    // A word is zeroed by the node that holds its LAST bit, with a plain store of what that lane has loaded anyway -- when all of the word's
                    // bits belong to THIS round of this wave [round_b0, run): every other node that touches the word has then been read, in this very round.
                    // The (at most two) words that reach over the round's ends lose this round's bits atomically, below.
                    const bool in0 = (w0 << 5) >= round_b0 && (w0 << 5) + 32 <= run, in1 = (w1 << 5) >= round_b0 && (w1 << 5) + 32 <= run;
                    if (bw0[r] && in0 && (w0 << 5) + 32 <= g1) bitmap[w0] = 0u;
                    if (w1 != w0 && bw1[r] && in1 && (g1 & 31) == 0) bitmap[w1] = 0u;
                }
            }
            if ((fw[r] >> (v & 31u)) & 1u) c = l[r];             // a step covered the whole node: a flag instead of marked bits
            cov_out[v] = c;
            const double len = (double)l[r];
            const double ab = (double)(long long)bs[r] / len;    // profile.rs:987-988
            ab_out[v] = ab;
            mx = fmax(mx, ab);
            if (ab > 0.0) ++nv;
            const double o = ab > min_depth ? ab : 0.0;          // :2941-2944
            if (o > 0.0) { zs += o; ++zc; }
        }
        if constexpr (CLEAN) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const uint32_t v = v0 + (uint32_t)r * 64u + lane;
                const bool in = v < whi;
                if (in && bs[r] != 0ull) bases[v] = 0ull;
                // the flags of this stretch's nodes, word by word: the first lane of every word's run of lanes clears the run's bits
                if (in && (lane == 0u || (v & 31u) == 0u)) {
                    const uint32_t n = min(min(32u - (v & 31u), 64u - lane), whi - v);
                    const uint32_t m = (n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u)) << (v & 31u);
                    if (fw[r] & m) { if (m == 0xFFFFFFFFu) full[v >> 5] = 0u; else atomicAnd(&full[v >> 5], ~m); }
                }
            }
            if (run > round_b0 && lane < 2u) {                   // the words over the round's two ends: this round's bits of them, atomically (lane 0: the first, lane 1: the last)
                const uint64_t ws = round_b0 >> 5, we = (run - 1) >> 5;
                const uint32_t ms = 0xFFFFFFFFu << (round_b0 & 31), me = 0xFFFFFFFFu >> (31 - (uint32_t)((run - 1) & 31));
                const bool s_part = (round_b0 & 31) != 0, e_part = (run & 31) != 0;
                if (lane == 0u && (s_part || (ws == we && e_part))) atomicAnd(&bitmap[ws], ~(ws == we ? ms & me : ms));
                if (lane == 1u && e_part && we != ws) atomicAnd(&bitmap[we], ~me);
            }
        }
    }
    __syncthreads();
    mx = block_max_f64<256>(mx, red);
    zs = block_sum_f64<256>(zs, red);
    nv = block_sum_u64<256>(nv, redu);
    zc = block_sum_u64<256>(zc, redu);
    if (threadIdx.x == 0) part[blockIdx.x] = {mx, zs, nv, zc};
}
// one wave per species: lane l combines chunks l, l+64, ... in order, then a fixed-shape wave reduction
__global__ void __launch_bounds__(64) node_stats_final_kernel(uint32_t S, const NodePartial *__restrict__ part, double *__restrict__ amax_out,
                                                              uint32_t *__restrict__ nvalid_out, double *__restrict__ nzsum_out,
                                                              uint32_t *__restrict__ nzcnt_out, const uint32_t *__restrict__ sp_chunk_off) {
    const uint32_t s = blockIdx.x, c0 = sp_chunk_off[s], nch = sp_chunk_off[s + 1] - c0;
    double mx = -INFINITY, zs = 0.0; unsigned long long nv = 0, zc = 0;
    for (uint32_t c = threadIdx.x; c < nch; c += 64) { NodePartial p = part[(size_t)c0 + c]; mx = fmax(mx, p.mx); zs += p.zs; nv += p.nv; zc += p.zc; }
    mx = wave_reduce(mx, [](double x, double y) { return fmax(x, y); });
    zs = wave_reduce(zs, [](double x, double y) { return x + y; });
    nv = wave_reduce(nv, [](unsigned long long x, unsigned long long y) { return x + y; });
    zc = wave_reduce(zc, [](unsigned long long x, unsigned long long y) { return x + y; });
    if (threadIdx.x == 0) { amax_out[s] = mx; nvalid_out[s] = (uint32_t)nv; nzsum_out[s] = zs; nzcnt_out[s] = (uint32_t)zc; }
}

int node_stats_launch(Ctx *ctx, const Db *db, LadBatch *lb, int64_t min_depth, const uint8_t *d_active) {
    if (ctx->cfg.no_absent_skip) d_active = nullptr;     // (tests compare, measurements)
    uint32_t S = db->S;
    lb->S = S;
    PTX_HIP(ctx, lb->d_ab.alloc(db->V));
    PTX_HIP(ctx, lb->d_amax.alloc(S)); PTX_HIP(ctx, lb->d_nvalid.alloc(S));
    PTX_HIP(ctx, lb->d_nzsum.alloc(S)); PTX_HIP(ctx, lb->d_nzcnt.alloc(S));
    PTX_HIP(ctx, lb->d_partial.alloc((size_t)S * STAT_CHUNKS * 4));
    const bool with_cov = db->cov_count_pending;                 // the resident step left the covered-base counts to this pass
    // the chunk table of this db (made once per variant): ~8192 workgroups in all for the fused kernel (it holds fewer workgroups per CU: shorter ones, so
    // that the last round is short), ~2048 for the plain one; every species at least one chunk and at most STAT_CHUNKS, in proportion to its nodes
    Db *dbm = const_cast<Db *>(db);
    Db::NodeChunks &nc = dbm->node_chunks[with_cov ? 1 : 0];
    if (nc.n == 0 && S) {
        const double target = std::max(1.0, (double)db->V / (with_cov ? 8192.0 : 2048.0));
        std::vector<uint32_t> off(S + 1, 0), sp;
        for (uint32_t s2 = 0; s2 < S; ++s2) {
            const double vs = (double)(db->h_node_off[s2 + 1] - db->h_node_off[s2]);
            const uint32_t k = (uint32_t)std::min<double>((double)STAT_CHUNKS, std::max(1.0, std::floor(vs / target + 0.5)));
            off[s2 + 1] = off[s2] + k;
            sp.insert(sp.end(), k, s2);
        }
        PTX_TRY(upload(ctx, nc.d_sp_off, off.data(), off.size()));
        PTX_TRY(upload(ctx, nc.d_chunk_sp, sp.data(), sp.size()));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (once per db: the staging vectors go out of scope)
        nc.n = off[S];
    }
    KTimer t(ctx, with_cov ? "node_cov_stats_kernel" : "node_stats_kernel");
    if (with_cov) {
        if (db->cov_self_clean)
        hipLaunchKernelGGL(node_cov_stats_kernel<true>, dim3(nc.n), dim3(256), 0, ctx->stream, db->d_node_base.p, db->d_node_len.p, db->d_bases.p, db->d_bit_off.p,
                           db->d_full.p, db->d_bitmap.p, (double)min_depth, db->d_cov.p, lb->d_ab.p, (NodePartial *)lb->d_partial.p, (const uint32_t *)nc.d_chunk_sp.p, (const uint32_t *)nc.d_sp_off.p, d_active);
        else if (db->V && db->L / db->V >= (uint64_t)ctx->cfg.ncs_prefix_min && !ctx->cfg.ncs_no_prefix)   // long nodes on average (chunk graphs of single-genome species among them): counts from a per-stretch prefix in LDS
        hipLaunchKernelGGL((node_cov_stats_kernel<false, true>), dim3(nc.n), dim3(256), (size_t)4 * NCS_PWORDS * sizeof(uint32_t), ctx->stream, db->d_node_base.p, db->d_node_len.p, db->d_bases.p, db->d_bit_off.p,
                           db->d_full.p, db->d_bitmap.p, (double)min_depth, db->d_cov.p, lb->d_ab.p, (NodePartial *)lb->d_partial.p, (const uint32_t *)nc.d_chunk_sp.p, (const uint32_t *)nc.d_sp_off.p, d_active);
        else
        hipLaunchKernelGGL(node_cov_stats_kernel<false>, dim3(nc.n), dim3(256), 0, ctx->stream, db->d_node_base.p, db->d_node_len.p, db->d_bases.p, db->d_bit_off.p,
                           db->d_full.p, db->d_bitmap.p, (double)min_depth, db->d_cov.p, lb->d_ab.p, (NodePartial *)lb->d_partial.p, (const uint32_t *)nc.d_chunk_sp.p, (const uint32_t *)nc.d_sp_off.p, d_active);
        dbm->cov_count_pending = false;
    } else
    hipLaunchKernelGGL(node_stats_kernel, dim3(nc.n), dim3(256), 0, ctx->stream, db->d_node_base.p, db->d_node_len.p, db->d_bases.p,
                       (double)min_depth, lb->d_ab.p, (NodePartial *)lb->d_partial.p, (const uint32_t *)nc.d_chunk_sp.p, (const uint32_t *)nc.d_sp_off.p);
    hipLaunchKernelGGL(node_stats_final_kernel, dim3(S), dim3(64), 0, ctx->stream, S, (const NodePartial *)lb->d_partial.p,
                       lb->d_amax.p, lb->d_nvalid.p, lb->d_nzsum.p, lb->d_nzcnt.p, (const uint32_t *)nc.d_sp_off.p);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// a11: row sub-sampling (sample_sorted, profile.rs:1287-1295 and its call sites :1394-1400 / :2738-2752).
// Only species with more valid rows than `sample_nodes` are touched, and only those cost a host round trip (their
// row count n decides the chosen ranks).  The chosen set is a bitmap over the RANKS of the valid rows in node
// order (row_sample.cpp); one chained scan ranks the valid nodes and clears the abundance of the unchosen ones in
// the LP's copy, so row_emit_kernel and objective_kernel see exactly the sampled rows.  max a (the x bound),
// path_cov_ratio and the single-path statistics were taken before and are not sampled (profile.rs:2700-2729).
// ---------------------------------------------------------------------------------------------
struct SampleLoad {
    const double *ab;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return ab[i] > 0.0 ? 1u : 0u; }
};
struct SampleStore {
    double *ab;
    const uint32_t *bits;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t rank, uint32_t valid) const {
        if (valid && !((bits[rank >> 5] >> (rank & 31)) & 1u)) ab[i] = 0.0;
    }
};

int row_sample_apply(Ctx *ctx, const Db *db, LadBatch *lb, int64_t sample_nodes) {
    const uint32_t S = db->S;
    bool possible = false;
    for (uint32_t s = 0; s < S && !possible; ++s) possible = (int64_t)(db->h_node_off[s + 1] - db->h_node_off[s]) > sample_nodes;
    if (sample_nodes <= 0 || !possible) return 0;   // no species can have more valid rows than the limit
    std::vector<uint32_t> nvalid(S);
    PTX_TRY(download(ctx, nvalid.data(), lb->d_nvalid.p, S));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> bits;
    DevBuf<uint32_t> d_bits;
    for (uint32_t s = 0; s < S; ++s) {
        if ((int64_t)nvalid[s] <= sample_nodes) continue;
        sample_ranks(nvalid[s], (uint64_t)sample_nodes, 42, bits);
        PTX_TRY(upload(ctx, d_bits, bits.data(), bits.size()));
        double *ab = lb->d_ab.p + db->h_node_off[s];
        PTX_TRY(exclusive_scan_fn(ctx, SampleLoad{ab}, SampleStore{ab, d_bits.p}, db->h_node_off[s + 1] - db->h_node_off[s], nullptr, "row_sample_kernel"));
        nvalid[s] = (uint32_t)sample_nodes;                // n of the objective's 1/n (profile.rs:2755)
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // d_bits is reused by the next species
    }
    PTX_HIP(ctx, hipMemcpyAsync(lb->d_nvalid.p, nvalid.data(), S * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// a10: membership masks (the 0/1 coefficient matrix, one u64 row per node) and path_cov_ratio
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t find_hap_l(const uint64_t *__restrict__ path_off, uint32_t H, uint64_t q) {
    uint32_t lo = 0, hi = H;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (path_off[mid] <= q) lo = mid + 1; else hi = mid; }
    return lo - 1;
}

__global__ void __launch_bounds__(256) mask_kernel(const uint2 *__restrict__ tiles, const uint64_t *__restrict__ path_off,
                                                   const uint32_t *__restrict__ path_nodes, const uint32_t *__restrict__ hap_species,
                                                   const uint32_t *__restrict__ node_base, const int32_t *__restrict__ hap_bit,
                                                   unsigned long long *__restrict__ mask, const int32_t *__restrict__ sp_p,
                                                   const uint32_t *__restrict__ wide_off /* null: no species can be wide */,
                                                   const uint32_t *__restrict__ wide_nw, unsigned long long *__restrict__ maskw,
                                                   const uint64_t *__restrict__ by_node_hap_off /* non-null: species of <= 64 haplotypes were done by mask_nodes_kernel */) {
    const uint2 tile = tiles[blockIdx.x];   // {hap, chunk}: see stage_trio.hip
    if (tile.x == 0xFFFFFFFFu) return;      // filler tile
    const uint32_t h = tile.x;
    const int bit = hap_bit[h];
    if (bit < 0) return;
    const uint32_t sp = hap_species[h];
    const uint32_t nb = node_base[sp];
    const uint64_t q0 = path_off[h] + (uint64_t)tile.y * PATH_TILE, qend = path_off[h + 1];
    if (wide_off && sp_p[sp] > LAD_MAXP) {   // wide species: wide_nw[sp] (LAD_WIDE_NW or more) words per node in the side array
        const unsigned long long m = 1ull << (bit & 63);
        const size_t nw = wide_nw[sp];
        unsigned long long *base = maskw + (size_t)wide_off[sp] * LAD_WIDE_NW + (bit >> 6);
        for (uint64_t q = q0 + threadIdx.x; q < q0 + PATH_TILE && q < qend; q += 256) {
            unsigned long long *w = base + (size_t)path_nodes[q] * nw;
            if ((*w & m) == 0) atomicOr(w, m);
        }
        return;
    }
    if (by_node_hap_off && by_node_hap_off[sp + 1] - by_node_hap_off[sp] <= 64ull) return;
    const unsigned long long m = 1ull << bit;
    for (uint64_t q = q0 + threadIdx.x; q < q0 + PATH_TILE && q < qend; q += 256) {
        unsigned long long *w = &mask[nb + path_nodes[q]];
        if ((*w & m) == 0) atomicOr(w, m);   // coeff_matrix[(v,pos)] = 1.0 even for repeated visits (profile.rs:1336-1340)
    }
}

// ---- the same matrix built BY NODE (round 3).  mask_kernel walks the candidates' paths and ORs a bit into the word of every
// node it meets: 2.2e9 path steps at cfg4, a read-test-atomic on a 2.5-GB array each, 9 ms of a 58-ms step at 0.8 TB/s.  Which
// haplotypes of its species visit a node depends on the database alone: node_haps_build writes that set once at upload as one
// 64-bit word per node (bit j = haplotype j of the species; a layout table like d_tiles and the node-block runs), and the step
// turns it into the candidates' word in registers -- one coalesced 8-byte load, a lookup in the species' haplotype -> column
// table per set bit, one 8-byte store, zero words included: no atomics, no zero fill, every byte touched once.
// Species of more than 64 haplotypes keep the path walk (their nodes get a zero here first).
__global__ void __launch_bounds__(256) node_haps_fill_kernel(const uint2 *__restrict__ tiles, const uint64_t *__restrict__ path_off,
                                                             const uint32_t *__restrict__ path_nodes, const uint32_t *__restrict__ hap_species,
                                                             const uint32_t *__restrict__ node_base, const uint64_t *__restrict__ hap_off,
                                                             unsigned long long *__restrict__ node_haps, uint32_t fast, const uint32_t *__restrict__ fast_slow) {
    const uint2 tile = tiles[blockIdx.x];
    if (tile.x == 0xFFFFFFFFu) return;
    const uint32_t h = tile.x, sp = hap_species[h], nb = node_base[sp];
    if (hap_off[sp + 1] - hap_off[sp] > 64ull) return;
    if (fast && !fast_slow[sp]) return;                     // a species of the visit table: node_haps_visits_kernel
    const unsigned long long m = 1ull << (h - hap_off[sp]);
    const uint64_t q0 = path_off[h] + (uint64_t)tile.y * PATH_TILE, qend = path_off[h + 1];
    for (uint64_t q = q0 + threadIdx.x; q < q0 + PATH_TILE && q < qend; q += 256) {
        unsigned long long *w = &node_haps[nb + path_nodes[q]];
        if ((*w & m) == 0) atomicOr(w, m);
    }
}
__global__ void __launch_bounds__(256) mask_nodes_kernel(uint64_t V, const uint2 *__restrict__ tile_sp, const uint32_t *__restrict__ node_base,
                                                         const uint64_t *__restrict__ hap_off, const int32_t *__restrict__ sp_p,
                                                         const int32_t *__restrict__ hap_bit, const unsigned long long *__restrict__ node_haps,
                                                         unsigned long long *__restrict__ mask, const uint32_t *__restrict__ cov,
                                                         const uint32_t *__restrict__ node_len, unsigned long long *__restrict__ ratio) {
    // one workgroup per 2048-node tile of d_emit_tile_sp (eight nodes per thread: the table below is set up once per 2048 nodes)
    // ratio != null: the path_cov_ratio sums of ratio_kernel (profile.rs:1344-1361) for every species this kernel builds the masks of,
    // taken while the mask is in a register -- ratio_kernel's 8V bytes of masks are not read a second time
    __shared__ int s_bit[64];     // haplotype -> LP column of the species the tile starts in (nearly always its only one)
    // ... and the same map BYTE-WISE: s_tab[b][x] = the columns of the haplotypes 8b .. 8b+7 whose bits are set in x.  A node's mask is the OR
    // of one entry per byte of its haplotype word (two lookups at ten haplotypes) instead of a loop over its set bits (the kernel was
    // bound by VALU issue: 205 instructions per 64 nodes, `r04_pmc_cfg4.json`); building 256 entries per used byte costs a thread one entry
    __shared__ unsigned long long s_tab[8][256];
    __shared__ unsigned long long acc[2 * LAD_MAXP];
    if (ratio && threadIdx.x < 2 * LAD_MAXP) acc[threadIdx.x] = 0;
    unsigned long long c8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t v0 = (uint64_t)blockIdx.x * 2048;
    const uint2 t = tile_sp[blockIdx.x];
    const uint32_t sp0 = t.x;
    {
        const uint64_t h0 = hap_off[sp0], nh = hap_off[sp0 + 1] - h0;
        if (threadIdx.x < 64) s_bit[threadIdx.x] = threadIdx.x < nh ? hap_bit[h0 + threadIdx.x] : -1;
    }
    const int p0 = sp_p[sp0];
    const uint64_t end0 = sp0 < t.y ? (uint64_t)node_base[sp0 + 1] : V;     // first node that is not of the tile's first species any more
    const int nbyte = (int)((hap_off[sp0 + 1] - hap_off[sp0] + 7) / 8);     // bytes of the haplotype word in use (block-uniform; > 8: a species the path walk fills)
    __syncthreads();
    if (p0 > 0 && p0 <= LAD_MAXP && nbyte <= 8) {
        for (int b = 0; b < nbyte; ++b) {
            unsigned long long e = 0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int bit = s_bit[8 * b + i]; if (((threadIdx.x >> i) & 1u) && bit >= 0) e |= 1ull << bit; }
            s_tab[b][threadIdx.x] = e;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t v = v0 + (uint64_t)r * 256 + threadIdx.x;
        if (v >= V) break;
        unsigned long long hm = node_haps[v];      // (zero for species of more than 64 haplotypes: the path walk fills those)
        const unsigned long long c = ratio ? cov[v] : 0ull, l = ratio ? node_len[v] : 0ull;
        unsigned long long m = 0ull;
        if (v < end0) {
            if (p0 > 0 && p0 <= LAD_MAXP && nbyte <= 8)
                for (int b = 0; b < nbyte; ++b) m |= s_tab[b][(hm >> (8 * b)) & 255ull];
            if (ratio && m) {                      // the first eight candidates (nearly always all) in registers, like ratio_kernel
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < p0) { const bool on = (m >> k) & 1ull; c8[k] += on ? c : 0ull; l8[k] += on ? l : 0ull; }   // (block-uniform: the columns that exist)
                unsigned long long rest = m >> 8;
                while (rest) {
                    const int k = __ffsll((long long)rest) - 1 + 8;
                    rest &= rest - 1;
                    if (c) atomicAdd(&acc[2 * k], c);
                    atomicAdd(&acc[2 * k + 1], l);
                }
            }
        } else {                                    // a species border inside the tile: the few nodes behind it look their species up
            uint32_t sp = sp0 + 1;
            while (sp < t.y && node_base[sp + 1] <= v) ++sp;
            const int p = sp_p[sp];
            if (p > 0 && p <= LAD_MAXP) {
                const int32_t *hb = hap_bit + hap_off[sp];
                while (hm) { const int j = __ffsll((long long)hm) - 1; hm &= hm - 1; const int bit = hb[j]; if (bit >= 0) m |= 1ull << bit; }
                unsigned long long rest = ratio ? m : 0ull;
                while (rest) {
                    const int k = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    if (c) atomicAdd(&ratio[2 * (hap_off[sp] + k)], c);
                    atomicAdd(&ratio[2 * (hap_off[sp] + k) + 1], l);
                }
            }
        }
        mask[v] = m;      // coeff_matrix[(v,pos)] = 1.0 for every candidate path that visits v (profile.rs:1336-1340)
    }
    if (!ratio || p0 <= 0 || p0 > LAD_MAXP) return;            // (block-uniform)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= p0) break;
        const unsigned long long cs = wave_reduce(c8[k], [](unsigned long long x, unsigned long long y) { return x + y; });
        const unsigned long long ls = wave_reduce(l8[k], [](unsigned long long x, unsigned long long y) { return x + y; });
        if ((threadIdx.x & 63) == 0) {
            if (cs) atomicAdd(&acc[2 * k], cs);
            if (ls) atomicAdd(&acc[2 * k + 1], ls);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * p0 && acc[threadIdx.x]) atomicAdd(&ratio[2 * hap_off[sp0] + threadIdx.x], acc[threadIdx.x]);
}

// Wide species: the one-word "mask" of a node becomes a 64-bit hash of its mask words (0 stays 0), so that the row grouping
// (sort by mask, runs of equal masks = patterns) works on it unchanged.  Equal hashes of different word sets are caught by
// wide_pattern_kernel / the solver (status 7), never silently merged.
constexpr int WIDE_CHUNKS = 64;
__device__ __forceinline__ unsigned long long wide_hash(const unsigned long long *w, int nw) {
    unsigned long long h = 0, any = 0;
    for (int i = 0; i < nw; ++i) { any |= w[i]; h = splitmix64(h ^ (w[i] + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1))); }
    return any ? (h ? h : 1ull) : 0ull;
}
__global__ void __launch_bounds__(256) mask_fold_kernel(const uint32_t *__restrict__ wide_list, const uint32_t *__restrict__ wide_off,
                                                        const uint32_t *__restrict__ wide_nw, const uint32_t *__restrict__ node_base, const int32_t *__restrict__ sp_p,
                                                        const unsigned long long *__restrict__ maskw, unsigned long long *__restrict__ mask) {
    const uint32_t s = wide_list[blockIdx.x / WIDE_CHUNKS], ch = blockIdx.x % WIDE_CHUNKS;
    if (sp_p[s] <= LAD_MAXP) return;
    const uint32_t b = node_base[s], n = node_base[s + 1] - b;
    const unsigned long long *mw = maskw + (size_t)wide_off[s] * LAD_WIDE_NW;
    const int nw = (int)wide_nw[s];
    for (uint32_t v = ch * 256 + threadIdx.x; v < n; v += WIDE_CHUNKS * 256) mask[b + v] = wide_hash(mw + (size_t)v * nw, nw);
}

// Wide species, after the patterns are known: every LP row's node finds its pattern (binary search of its hash among the
// species' patterns, which are sorted by it) and ORs / ANDs its mask words into the pattern's slots.  OR == AND for every
// pattern <=> all of its rows have the same words (the solver checks and reports status 7 otherwise).
__global__ void __launch_bounds__(256) wide_pattern_kernel(const uint32_t *__restrict__ wide_list, const uint32_t *__restrict__ wide_off,
                                                           const uint32_t *__restrict__ wide_nw, const uint32_t *__restrict__ node_base, const int32_t *__restrict__ sp_p,
                                                           const double *__restrict__ ab, const unsigned long long *__restrict__ mask,
                                                           const unsigned long long *__restrict__ maskw, const uint32_t *__restrict__ sp_pat_off,
                                                           const uint64_t *__restrict__ pat_mask, unsigned long long *__restrict__ pat_or,
                                                           unsigned long long *__restrict__ pat_and) {
    const uint32_t s = wide_list[blockIdx.x / WIDE_CHUNKS], ch = blockIdx.x % WIDE_CHUNKS;
    if (sp_p[s] <= LAD_MAXP) return;
    const uint32_t b = node_base[s], n = node_base[s + 1] - b;
    const uint32_t k0 = sp_pat_off[s], k1 = sp_pat_off[s + 1];
    const size_t wo = (size_t)wide_off[s] * LAD_WIDE_NW, nw = wide_nw[s];
    for (uint32_t v = ch * 256 + threadIdx.x; v < n; v += WIDE_CHUNKS * 256) {
        const unsigned long long hm = mask[b + v];
        if (!(ab[b + v] > 0.0) || hm == 0ull) continue;
        uint32_t lo = k0, hi = k1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pat_mask[mid] < hm) lo = mid + 1; else hi = mid; }
        if (lo >= k1 || pat_mask[lo] != hm) continue;   // cannot happen: every such node is a row
        for (size_t i = 0; i < nw; ++i) {
            const unsigned long long w = maskw[wo + (size_t)v * nw + i];
            atomicOr(&pat_or[wo + (size_t)(lo - k0) * nw + i], w);
            atomicAnd(&pat_and[wo + (size_t)(lo - k0) * nw + i], w);
        }
    }
}

constexpr int RATIO_CHUNKS = 128;
constexpr int ROW_ITEMS = 8;   // nodes per thread of the row compaction kernels
// path_cov_ratio sums (profile.rs:1344-1361): per candidate k, sum of covered bases and of lengths over its
// nodes.  The first 8 candidates (nearly always all of them) accumulate in registers and are combined by wave
// reductions; 64 lanes hammering 2-4 LDS addresses with 64-bit atomics serialise.
__global__ void __launch_bounds__(256) ratio_kernel(const uint32_t *__restrict__ node_base, const uint32_t *__restrict__ node_len,
                                                    const uint32_t *__restrict__ cov, const unsigned long long *__restrict__ mask,
                                                    const int32_t *__restrict__ sp_p, const uint64_t *__restrict__ hap_off,
                                                    const uint32_t *__restrict__ wide_off, const uint32_t *__restrict__ wide_nw,
                                                    const unsigned long long *__restrict__ maskw, unsigned long long *__restrict__ ratio,
                                                    int by_node_done /* the species of at most 64 haplotypes got their sums from mask_nodes_kernel */) {
    __shared__ unsigned long long acc[LAD_WIDEP * 2];
    const uint32_t s = blockIdx.x / RATIO_CHUNKS, ch = blockIdx.x % RATIO_CHUNKS;
    const int p = sp_p[s];
    if (p <= 0) return;
    if (by_node_done && hap_off[s + 1] - hap_off[s] <= 64) return;
    const uint32_t b = node_base[s], e = node_base[s + 1];
    const uint32_t per = (e - b + RATIO_CHUNKS - 1) / RATIO_CHUNKS;
    uint32_t lo = b + ch * per, hi = lo + per;
    if (hi > e) hi = e;
    if (p > LAD_MAXP) {   // wide species: the candidates through the LDS accumulators, LAD_WIDEP (four mask words) at a time
        const unsigned long long *mw = maskw + (size_t)wide_off[s] * LAD_WIDE_NW;
        const size_t nw = wide_nw[s];
        for (int kb = 0; kb < p; kb += LAD_WIDEP) {
            const int pn = p - kb < LAD_WIDEP ? p - kb : LAD_WIDEP;
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * pn; i += 256) acc[i] = 0;
            __syncthreads();
            for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
                const unsigned long long c = cov[v], l = node_len[v];
#pragma unroll
                for (int i = 0; i < LAD_WIDE_NW; ++i) {
                    unsigned long long m = mw[(size_t)(v - b) * nw + (kb >> 6) + i];
                    while (m) {
                        const int k = 64 * i + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        if (c) atomicAdd(&acc[2 * k], c);
                        atomicAdd(&acc[2 * k + 1], l);
                    }
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * pn; i += 256) if (acc[i]) atomicAdd(&ratio[2 * (hap_off[s] + kb) + i], acc[i]);
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * p; i += 256) acc[i] = 0;
    __syncthreads();
    unsigned long long c8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
        unsigned long long m = mask[v];
        if (!m) continue;
        const unsigned long long c = cov[v], l = node_len[v];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool on = (m >> k) & 1ull;
            c8[k] += on ? c : 0ull;
            l8[k] += on ? l : 0ull;
        }
        m >>= 8;
        while (m) {
            int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (c) atomicAdd(&acc[2 * (k + 8)], c);
            atomicAdd(&acc[2 * (k + 8) + 1], l);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long cs = wave_reduce(c8[k], [](unsigned long long x, unsigned long long y) { return x + y; });
        const unsigned long long ls = wave_reduce(l8[k], [](unsigned long long x, unsigned long long y) { return x + y; });
        if ((threadIdx.x & 63) == 0) {
            if (cs) atomicAdd(&acc[2 * k], cs);
            if (ls) atomicAdd(&acc[2 * k + 1], ls);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * p && acc[threadIdx.x]) atomicAdd(&ratio[2 * hap_off[s] + threadIdx.x], acc[threadIdx.x]);
}


// ---------------------------------------------------------------------------------------------
// LP rows: nodes with a_v > 0 (valid rows, profile.rs:1380-1385) and a non-empty mask; rows with an
// empty mask only add the constant a_v to the objective and are handled by objective_kernel.
// ---------------------------------------------------------------------------------------------
// One launch: every workgroup compacts its tile of nodes and claims its output range with a single atomic
// on the row counter.  Row order across workgroups is arbitrary, which is immaterial: the rows are sorted by
// their full key (species, mask, a) next, and rows with equal keys are indistinguishable.
__global__ void __launch_bounds__(256) row_emit_kernel(uint64_t V, uint32_t S, const uint32_t *__restrict__ node_base, const double *__restrict__ ab,
                                                       const unsigned long long *__restrict__ mask, uint32_t *__restrict__ n_rows,
                                                       uint64_t *__restrict__ k0, uint64_t *__restrict__ k1, uint64_t *__restrict__ k2,
                                                       int pack_shift /* >= 0: two-word rows {species << shift | mask, a} in k0, k1 */) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * ROW_ITEMS;
    double a[ROW_ITEMS];
    unsigned long long m[ROW_ITEMS];
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < ROW_ITEMS; ++i) {
        const uint64_t v = base + i;
        a[i] = 0.0; m[i] = 0;
        if (v < V) { a[i] = ab[v]; m[i] = mask[v]; }
        cnt += (a[i] > 0.0 && m[i] != 0ull) ? 1u : 0u;
    }
    // exclusive offsets inside the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { uint32_t t = s_wave[w]; if (w < wave) woff += t; tot += t; }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(n_rows, tot) : 0u;
    __syncthreads();
    uint32_t j = s_base + woff + incl - cnt;
    uint32_t sp1 = 0;              // 1 + species of the previous emitted node of this thread (its nodes are consecutive)
#pragma unroll
    for (int i = 0; i < ROW_ITEMS; ++i) {
        if (!(a[i] > 0.0 && m[i] != 0ull)) continue;
        const uint64_t v = base + i;
        uint32_t lo;
        if (sp1 == 0) {            // species of node v: last s with node_base[s] <= v
            uint32_t hi = S;
            lo = 0;
            while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (node_base[mid] <= v) lo = mid + 1; else hi = mid; }
        } else {
            lo = sp1;
            while (lo < S && node_base[lo] <= v) ++lo;   // at most a species border or two between neighbouring nodes
        }
        sp1 = lo;
        const uint64_t abits = (uint64_t)__double_as_longlong(a[i]);   // positive doubles order like their bit patterns
        if (pack_shift >= 0) {
            k0[j] = (pack_shift < 64 ? ((uint64_t)(lo - 1) << pack_shift) : 0ull) | m[i];
            k1[j] = abits;
        } else {
            k0[j] = lo - 1;
            k1[j] = m[i];
            k2[j] = abits;
        }
        ++j;
    }
}
// Patterns = runs of equal (species, mask) in the sorted rows.  One chained-scan launch: the head flag of a row is
// computed from the keys as it is loaded, and a head whose exclusive prefix is j emits pattern j on the spot.
struct PatLoad {
    const uint32_t *d_n;
    const uint64_t *k0, *k1;   // k1 == null: species and mask share k0
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
        const uint64_t n = *d_n;   // rows actually present (the scan covers the host-side bound)
        return (i < n && (i == 0 || k0[i] != k0[i - 1] || (k1 && k1[i] != k1[i - 1]))) ? 1u : 0u;
    }
};
struct PatStore {
    const uint64_t *k0, *k1;
    uint32_t k_cap;
    int pack_shift;
    uint64_t *pat_mask;
    uint32_t *pat_start, *pat_species, *overflow;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t j, uint32_t head) const {
        if (!head) return;
        if (j >= k_cap) { *overflow = 1; return; }   // more patterns than this build sizes for: reported as PANTAX_HIP_E_LIMIT
        if (pack_shift >= 0) {
            const uint64_t w = k0[i];
            pat_mask[j] = pack_shift < 64 ? (w & ((1ull << pack_shift) - 1ull)) : w;
            pat_species[j] = pack_shift < 64 ? (uint32_t)(w >> pack_shift) : 0u;
        } else {
            pat_mask[j] = k1[i];
            pat_species[j] = (uint32_t)k0[i];
        }
        pat_start[j] = (uint32_t)i;
    }
};

// species -> first pattern (patterns are sorted by species); entry S = K; also closes pat_start[K] = n_rows
__global__ void __launch_bounds__(256) sp_pat_off_kernel(uint32_t S, const uint32_t *__restrict__ d_K, uint32_t k_cap,
                                                         const uint32_t *__restrict__ pat_species, const uint32_t *__restrict__ d_n,
                                                         uint32_t *__restrict__ pat_start, uint32_t *__restrict__ sp_pat_off) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s > S) return;
    const uint32_t K = min(*d_K, k_cap), n_rows = *d_n;
    uint32_t lo = 0, hi = K;   // first pattern with species >= s
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (pat_species[m] < s) lo = m + 1; else hi = m; }
    sp_pat_off[s] = (s == S) ? K : lo;
    if (s == S) pat_start[K] = n_rows;
}

// option mask=walk: the path-walk kernel although the table exists (measurements, tests)
// ... from the VISIT TABLE where a species has one (round 5): the interior visits of a node sit in one stretch of one 64-lane group, so the word of a node
// is the OR over its stretch of (1 << owner of the visit's position) -- one wave per group, the owners from the species' walk offsets held one per lane
// (as in the filing of the index rows), one ballot per haplotype of the species, one plain 8-byte store per node; the two END positions of every walk
// are no interior visits and come in by atomics afterwards (node_haps_ends_kernel).  The pass over the walks above issued a probe + an atomic per path
// step: 21 ms at 1e4 strains, 50-62 ms per db of 2.8e9 path steps at fifty strains per species.
__global__ void __launch_bounds__(256) node_haps_visits_kernel(uint32_t NG, const uint32_t *__restrict__ vis_pos, const uint64_t *__restrict__ vis_head,
                                                               const uint32_t *__restrict__ vis_nbase, const uint32_t *__restrict__ vis_sp,
                                                               const uint64_t *__restrict__ path_off, const uint64_t *__restrict__ hap_off,
                                                               const uint32_t *__restrict__ path_nodes, const uint32_t *__restrict__ by_walk,
                                                               unsigned long long *__restrict__ node_haps) {
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= NG) return;
    if (by_walk[vis_sp[g]]) return;                        // (wave-uniform) a species of many haplotypes: the pass over its walks is cheaper
    const int lane = threadIdx.x & 63;
    const uint32_t q = vis_pos[(uint64_t)g * 64 + lane];
    const bool valid = q != 0xFFFFFFFFu;
    const uint32_t sp = vis_sp[g], nb = vis_nbase[g];
    const uint32_t h0 = (uint32_t)hap_off[sp], hs = (uint32_t)hap_off[sp + 1] - h0;
    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid), hd = vis_head[g] & vmask;
    const uint32_t woff = (uint32_t)lane < hs ? (uint32_t)path_off[h0 + (uint32_t)lane] : 0xFFFFFFFFu;     // P < 2^32 where a visit table exists
    uint32_t hl = 0;                                                                     // owner within the species: walk offsets at or below the position, minus one
    for (uint32_t j = 1; j < hs; ++j) hl += (uint32_t)__builtin_amdgcn_readlane((int)woff, (int)j) <= q ? 1u : 0u;
    const bool head = (hd >> lane) & 1ull;
    const uint32_t mid = head ? path_nodes[q] : 0u;                                      // the stretch's node (its first visit names it)
    // my stretch = lanes [lane, next head or first pad)
    const unsigned long long he = hd | (~vmask & (vmask + 1ull));
    const unsigned long long above = he & ~((2ull << lane) - 1ull);
    const int end = above ? __builtin_ctzll(above) : 64;
    const unsigned long long range = (end == 64 ? ~0ull : (1ull << end) - 1ull) & ~((1ull << lane) - 1ull);
    unsigned long long word = 0ull;
    for (uint32_t j = 0; j < hs; ++j) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(valid && hl == j);
        if (bal & range) word |= 1ull << j;
    }
    if (head) node_haps[nb + mid] = word;
}
// the first and the last position of every walk of a visit-table species (a walk of one or two positions has no interior visit at all)
__global__ void __launch_bounds__(256) node_haps_ends_kernel(uint32_t H, const uint64_t *__restrict__ path_off, const uint32_t *__restrict__ path_nodes,
                                                             const uint32_t *__restrict__ hap_species, const uint32_t *__restrict__ node_base,
                                                             const uint64_t *__restrict__ hap_off, const uint32_t *__restrict__ slow,
                                                             unsigned long long *__restrict__ node_haps) {
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const uint32_t sp = hap_species[h];
    if (slow[sp]) return;                                  // (slow = the species' words come from the pass over its walks)
    const uint64_t b = path_off[h], e = path_off[h + 1];
    if (e == b) return;
    const unsigned long long m = 1ull << (h - hap_off[sp]);
    atomicOr(&node_haps[node_base[sp] + path_nodes[b]], m);
    atomicOr(&node_haps[node_base[sp] + path_nodes[e - 1]], m);
}
bool use_node_haps(const Ctx *ctx, const Db *db) { return db->nh_built && ctx->cfg.mask != "walk"; }
// end of db upload: the node -> haplotypes words of mask_nodes_kernel (one launch over the path tiles)
int node_haps_build(Ctx *ctx, Db *db) {
    db->nh_built = false; db->nh_walk_too = false;
    const uint64_t V = db->V;
    if (!V || !db->P || !db->n_tiles) return 0;
    bool any_small = false;
    for (uint32_t s = 0; s < db->S; ++s) {
        if (db->h_hap_off[s + 1] - db->h_hap_off[s] > 64) db->nh_walk_too = true; else any_small = true;
    }
    if (!any_small) return 0;
    PTX_HIP(ctx, db->d_node_haps.alloc(V));
    PTX_TRY(zero_fill(ctx, db->d_node_haps.p, V * sizeof(uint64_t)));
    // the species of the visit table with up to NH_VISIT_HAPS haplotypes from the table, the others by the pass over their walks (the table kernel costs a
    // readlane + a ballot per haplotype of the species and group: at fifty haplotypes 78 ms per db of 2.8e9 path steps against 50 for the walks' atomics;
    // at ten: ms against tens of ms)
    constexpr uint64_t NH_VISIT_HAPS = 16;
    const bool table = db->trio_visit_ok && db->n_vgroups && db->P < 0xFFFFFFFFull;
    std::vector<uint32_t> by_walk(db->S ? db->S : 1, 1u);
    bool any_walk = false, any_visits = false;
    for (uint32_t s = 0; s < db->S; ++s) {
        const uint64_t hs = db->h_hap_off[s + 1] - db->h_hap_off[s];
        by_walk[s] = (table && !db->h_trio_slow[s] && hs <= NH_VISIT_HAPS) ? 0u : 1u;
        if (hs <= 64) { if (by_walk[s]) any_walk = true; else any_visits = true; }
    }
    DevBuf<uint32_t> d_by_walk;
    PTX_TRY(upload(ctx, d_by_walk, by_walk.data(), by_walk.size()));
    if (any_visits) {
        hipLaunchKernelGGL(node_haps_visits_kernel, dim3((db->n_vgroups + 3) / 4), dim3(256), 0, ctx->stream, db->n_vgroups, db->d_vis_pos.p, db->d_vis_head.p, db->d_vis_nbase.p,
                           db->d_vis_sp.p, db->d_path_off.p, db->d_hap_off.p, db->d_path_nodes.p, (const uint32_t *)d_by_walk.p, (unsigned long long *)db->d_node_haps.p);
        hipLaunchKernelGGL(node_haps_ends_kernel, dim3((uint32_t)((db->H + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)db->H, db->d_path_off.p, db->d_path_nodes.p,
                           db->d_hap_species.p, db->d_node_base.p, db->d_hap_off.p, (const uint32_t *)d_by_walk.p, (unsigned long long *)db->d_node_haps.p);
    }
    if (any_walk)
        hipLaunchKernelGGL(node_haps_fill_kernel, dim3((uint32_t)db->n_tiles), dim3(256), 0, ctx->stream, db->d_tiles.p, db->d_path_off.p, db->d_path_nodes.p,
                           db->d_hap_species.p, db->d_node_base.p, db->d_hap_off.p, (unsigned long long *)db->d_node_haps.p, 1u, (const uint32_t *)d_by_walk.p);
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // d_by_walk goes out of scope
    PTX_HIP(ctx, hipGetLastError());
    db->nh_built = true;
    return 0;
}

// All of it is enqueued without a host round trip: the row count n and the pattern count K stay on the
// device (lb->d_counts = {n_rows, K, overflow}); buffers are sized by their host-known bounds (n <= V,
// K <= k_cap).  cand_on_device: lb->d_hap_bit / d_p were written by first_filter_kernel; otherwise they are
// uploaded from lb->h_p / h_cand (solver seam).  pmax_bound = upper bound of candidates per species.
int lad_prepare(Ctx *ctx, const Db *db, LadBatch *lb, bool cand_on_device, int pmax_bound) {
    const uint32_t S = db->S;
    lb->rows_c0_valid = false;
    const uint64_t V = db->V, H = db->H;
    if (!cand_on_device) {
        std::vector<int32_t> hap_bit(H ? H : 1, -1);
        for (uint32_t s = 0; s < S; ++s)
            for (int k = 0; k < lb->h_p[s]; ++k) hap_bit[db->h_hap_off[s] + lb->h_cand[db->h_hap_off[s] + k]] = k;
        PTX_TRY(upload(ctx, lb->d_hap_bit, hap_bit.data(), hap_bit.size()));
        PTX_TRY(upload(ctx, lb->d_p, lb->h_p.data(), S));
    }
    PTX_HIP(ctx, lb->d_mask.alloc(V));
    PTX_HIP(ctx, lb->d_ratio.alloc((size_t)(H ? H : 1) * 2));
    if (!lb->prezeroed && !use_node_haps(ctx, db)) PTX_TRY(zero_fill(ctx, lb->d_mask.p, V * sizeof(uint64_t)));   // (mask_nodes_kernel writes every word)
    // species that can be wide (more than 64 haplotypes): side arrays laid out once per db
    // More than LAD_WIDEP haplotypes ("huge"): as many mask words as the haplotypes need, rounded up to whole groups of
    // LAD_WIDE_NW -- the reference has no cap on the LP columns (dense nvert x npaths matrix, profile.rs:1333-1342), and neither
    // has this path; what grows is the scratch (W and G: 3 x (64 nw)^2 doubles per such species) and the time of one workgroup.
    if (lb->wide_for != (const void *)db) {
        std::vector<uint32_t> off(S ? S : 1, 0xFFFFFFFFu), slot(S ? S : 1, 0xFFFFFFFFu), nwv(S ? S : 1, 0u), list;
        std::vector<uint64_t> woff, coff;
        uint64_t vw = 0, wtot = 0, ctot = 0;
        uint32_t n_huge = 0;
        for (uint32_t s = 0; s < S; ++s) {
            const uint64_t Hs = db->h_hap_off[s + 1] - db->h_hap_off[s];
            if (Hs <= (uint64_t)LAD_MAXP) continue;
            if (Hs > 30000ull) return fail(ctx, PANTAX_HIP_E_LIMIT, "lad_prepare: species %u has %llu haplotypes; the basis inverse is indexed with 32 bits (30000 columns)", s, (unsigned long long)Hs);
            const uint32_t nw = (uint32_t)((Hs + LAD_WIDEP - 1) / LAD_WIDEP) * LAD_WIDE_NW;
            off[s] = (uint32_t)vw; slot[s] = (uint32_t)list.size(); list.push_back(s); nwv[s] = nw;
            vw += (db->h_node_off[s + 1] - db->h_node_off[s]) * (nw / LAD_WIDE_NW);
            woff.push_back(wtot); coff.push_back(ctot);
            wtot += 64ull * nw * 64ull * nw;
            if (nw > (uint32_t)LAD_WIDE_NW) { ++n_huge; ctot += 64ull * nw; }
        }
        if (vw >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "lad_prepare: %llu four-word mask groups in species of more than %d haplotypes", (unsigned long long)vw, LAD_MAXP);
        lb->n_wide = (uint32_t)list.size(); lb->n_huge = n_huge; lb->Vw = vw;
        if (lb->n_wide) {
            PTX_TRY(upload(ctx, lb->d_wide_off, off.data(), S));
            PTX_TRY(upload(ctx, lb->d_wide_slot, slot.data(), S));
            PTX_TRY(upload(ctx, lb->d_wide_nw, nwv.data(), S));
            PTX_TRY(upload(ctx, lb->d_wide_list, list.data(), list.size()));
            woff.insert(woff.end(), coff.begin(), coff.end());     // [n_wide] W offsets, then [n_wide] column-state offsets
            PTX_TRY(upload(ctx, lb->d_wide_woff, woff.data(), woff.size()));
            PTX_HIP(ctx, lb->d_maskw.alloc(vw * LAD_WIDE_NW)); PTX_HIP(ctx, lb->d_pat_or.alloc(vw * LAD_WIDE_NW)); PTX_HIP(ctx, lb->d_pat_and.alloc(vw * LAD_WIDE_NW));
            // W and G are sized by ALL haplotypes of such a species (the candidate count is decided on the device, after the first filter):
            // 3 x (64 nw)^2 doubles each -- 0.6 GB at 5 000 haplotypes, 22 GB at the 30 000 limit.  A db whose scratch does not fit is refused
            // with the figure, not with a bare allocation error (INTEGRATION.md states the cost)
            if (lb->d_wide_W.alloc(wtot) != hipSuccess || lb->d_wide_G.alloc(2 * wtot) != hipSuccess) {
                (void)hipGetLastError();
                lb->d_wide_W.release(); lb->d_wide_G.release();
                return fail(ctx, PANTAX_HIP_E_LIMIT, "lad_prepare: %.1f GB of solver scratch for the %u species of more than %d haplotypes do not fit in device memory "
                            "(3 x (64 x words)^2 doubles per species)", 3.0 * (double)wtot * 8.0 / 1e9, (uint32_t)list.size(), LAD_MAXP);
            }
            if (n_huge) { PTX_HIP(ctx, lb->d_huge_f64.alloc(ctot * 8)); PTX_HIP(ctx, lb->d_huge_i32.alloc(ctot * 5)); }
        }
        lb->wide_for = (const void *)db;
    }
    const bool wide = lb->n_wide != 0;
    if (wide) {
        PTX_TRY(zero_fill(ctx, lb->d_maskw.p, lb->Vw * LAD_WIDE_NW * sizeof(uint64_t)));
        PTX_TRY(zero_fill(ctx, lb->d_pat_or.p, lb->Vw * LAD_WIDE_NW * sizeof(uint64_t)));
        PTX_HIP(ctx, hipMemsetAsync(lb->d_pat_and.p, 0xFF, lb->Vw * LAD_WIDE_NW * sizeof(uint64_t), ctx->stream));
    }
    // d_ratio and d_counts live in the step's result arena, which the caller has just zeroed
    // rows: compact -> sort by (species, mask, a)
    Db *dbm = const_cast<Db *>(db);   // staging buffers live in the db so repeated steps do not hipMalloc
    // Many species: the rows are sorted species by species straight from the node arrays (sample_sort_nodes.hip) -- no compaction pass in front, no limit
    // on a species' size below 2^26 nodes (round 3's compaction + segmented sample sort, unreachable since round 4, was deleted in round 5)
    uint64_t max_vs = 0;
    for (uint32_t s_ = 0; s_ < S; ++s_) max_vs = std::max<uint64_t>(max_vs, db->h_node_off[s_ + 1] - db->h_node_off[s_]);
    bool use_nodes = V > SS_MAX_N && max_vs <= SSN_MAX_SEG && S <= 65535;
    if (!ctx->cfg.row_sort.empty()) {   // measurements / tests: "radix" = the whole-batch sorts at any size; "nodes" = the batched sort wherever it can run
        const char *ev = ctx->cfg.row_sort.c_str();
        if (ev[0] == 'r') use_nodes = false;
        if (ev[0] == 'n') use_nodes = max_vs <= SSN_MAX_SEG && S <= 65535 && V > 0;
    }
    const bool by_node = use_node_haps(ctx, db);
    // the path_cov_ratio sums ride on the by-node mask pass (PANTAX_RATIO=kernel: ratio_kernel for every species, as in round 3)
    const bool ratio_sep = ctx->cfg.ratio_kernel;
    const bool ratio_by_node = by_node && V && !ratio_sep;
    // ... and where the rows are sorted straight from the node arrays and every species has at most 64 haplotypes, the masks are formed INSIDE the
    // sort's histogram pass (ssn_hist_kernel<true>): no mask array, no pass of its own (PANTAX_MASK_PASS=1 keeps mask_nodes_kernel; so do the
    // measurement modes that read the array afterwards)
    const bool mask_pass_env = ctx->cfg.mask_pass || ctx->cfg.objective == "nodes";
    const bool masks_in_sort = use_nodes && ratio_by_node && !db->nh_walk_too && !wide && !mask_pass_env;
    lb->masks_in_sort = masks_in_sort;
    if (!masks_in_sort) {
        KTimer t(ctx, by_node ? "mask_nodes_kernel" : "mask_kernel");   // the names rocprofv3 shows
        if (by_node && V)
            hipLaunchKernelGGL(mask_nodes_kernel, dim3((uint32_t)((V + 2047) / 2048)), dim3(256), 0, ctx->stream, V, db->d_emit_tile_sp.p, db->d_node_base.p, db->d_hap_off.p,
                               lb->d_p.p, lb->d_hap_bit.p, (const unsigned long long *)db->d_node_haps.p, (unsigned long long *)lb->d_mask.p,
                               ratio_by_node ? db->d_cov.p : (const uint32_t *)nullptr, db->d_node_len.p, ratio_by_node ? lb->d_ratio.p : (unsigned long long *)nullptr);
        if (db->n_tiles && (!by_node || db->nh_walk_too))
            hipLaunchKernelGGL(mask_kernel, dim3((uint32_t)db->n_tiles), dim3(256), 0, ctx->stream, db->d_tiles.p, db->d_path_off.p,
                               db->d_path_nodes.p, db->d_hap_species.p, db->d_node_base.p, lb->d_hap_bit.p, (unsigned long long *)lb->d_mask.p,
                               lb->d_p.p, wide ? lb->d_wide_off.p : (const uint32_t *)nullptr, lb->d_wide_nw.p, (unsigned long long *)lb->d_maskw.p,
                               by_node ? db->d_hap_off.p : (const uint64_t *)nullptr);
    }
    if (!masks_in_sort && (!ratio_by_node || db->nh_walk_too)) {   // what the mask pass did not sum: species of more than 64 haplotypes (their masks come from the path walk)
        KTimer t(ctx, "ratio_kernel");
        hipLaunchKernelGGL(ratio_kernel, dim3(S * RATIO_CHUNKS), dim3(256), 0, ctx->stream, db->d_node_base.p, db->d_node_len.p, db->d_cov.p,
                           (unsigned long long *)lb->d_mask.p, lb->d_p.p, db->d_hap_off.p, lb->d_wide_off.p, lb->d_wide_nw.p,
                           (const unsigned long long *)lb->d_maskw.p, lb->d_ratio.p, ratio_by_node ? 1 : 0);
    }
    if (wide)
        hipLaunchKernelGGL(mask_fold_kernel, dim3(lb->n_wide * WIDE_CHUNKS), dim3(256), 0, ctx->stream, lb->d_wide_list.p, lb->d_wide_off.p,
                           lb->d_wide_nw.p, db->d_node_base.p, lb->d_p.p, (const unsigned long long *)lb->d_maskw.p, (unsigned long long *)lb->d_mask.p);
    DevBuf<uint32_t> &scan_tmp = dbm->d_scan_tmp, &table = dbm->d_sort_table;
    PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(std::max<uint64_t>(V, 256ull * 2048))));
    PTX_HIP(ctx, table.alloc(sort_table_elems(V)));
    PTX_HIP(ctx, lb->d_counts.alloc(4));
    uint32_t *d_n = lb->d_counts.p, *d_K = lb->d_counts.p + 1, *d_ovf = lb->d_counts.p + 2;
    DevBuf<uint64_t> *ka = dbm->d_ka, *kb = dbm->d_kb;
    for (int w = 0; w < 3; ++w) { PTX_HIP(ctx, ka[w].alloc(V)); if (!use_nodes) PTX_HIP(ctx, kb[w].alloc(V)); }
    // above the sample-sort limit the rows go through the radix sort; species and mask then share one key word
    // whenever their bits fit (16-byte records instead of 24)
    const int sp_bits = S > 1 ? bits_for(S - 1) : 0;
    const bool use_sample = V <= SS_MAX_N;
    const int pack_shift = (!use_sample && sp_bits + pmax_bound <= 64 && !(use_nodes && pmax_bound >= 64)) ? pmax_bound : -1;
    if (use_nodes) {
        if (pack_shift >= 64) return fail(ctx, PANTAX_HIP_E_LIMIT, "lad_prepare: internal (64 candidate columns and a packed species key)");
    } else {
        KTimer t(ctx, "row_emit_kernel");   // d_n was zeroed with the step's result arena
        const uint32_t grid_rows = (uint32_t)((V + 256ull * ROW_ITEMS - 1) / (256ull * ROW_ITEMS));
        hipLaunchKernelGGL(row_emit_kernel, dim3(grid_rows ? grid_rows : 1), dim3(256), 0, ctx->stream, V, S, db->d_node_base.p, lb->d_ab.p,
                           (unsigned long long *)lb->d_mask.p, d_n, ka[0].p, ka[1].p, ka[2].p, pack_shift);
    }
    if (dbm->trio_free_pending && dbm->ev_trio_free) {   // the next step's index rebuild may start from here (api_strain.cpp)
        PTX_HIP(ctx, hipEventRecord(dbm->ev_trio_free, ctx->stream));
        dbm->trio_free_valid = true; dbm->trio_free_pending = false;
    }
    SortBufs A, B;
    A.nw = B.nw = pack_shift >= 0 ? 2 : 3;
    for (int w = 0; w < 3; ++w) { A.k[w] = ka[w].p; B.k[w] = kb[w].p; }
    bool in_b = false;
    // patterns = runs of equal (species, mask)
    const uint64_t k_cap = V;   // patterns are runs of rows and rows are nodes: never more than V, so the tables cannot overflow
    lb->k_cap = (uint32_t)k_cap;
    PTX_HIP(ctx, lb->d_pat_mask.alloc(k_cap)); PTX_HIP(ctx, lb->d_pat_start.alloc(k_cap + 1)); PTX_HIP(ctx, lb->d_pat_species.alloc(k_cap));
    PTX_HIP(ctx, lb->d_sp_pat_off.alloc(S + 1));
    if (use_nodes) {   // no compaction: the sort's passes read the node arrays and skip the nodes that are no rows; the patterns come from its splitters
        PTX_HIP(ctx, dbm->d_ss_ws.alloc(sample_sort_nodes_ws_elems(S, max_vs, V)));
        PTX_HIP(ctx, dbm->d_row16.alloc(4 * V));
        PTX_HIP(ctx, lb->d_c0.alloc(S));
        const RowPatterns pat{lb->d_pat_mask.p, lb->d_pat_start.p, lb->d_pat_species.p, lb->d_sp_pat_off.p, d_K, lb->d_c0.p};
        lb->rows_c0_valid = true;
        RowMaskSource hp;
        if (masks_in_sort) {
            uint32_t mh = 0;
            for (uint32_t s_ = 0; s_ < S; ++s_) mh = std::max<uint32_t>(mh, (uint32_t)(db->h_hap_off[s_ + 1] - db->h_hap_off[s_]));
            hp.node_haps = (const unsigned long long *)db->d_node_haps.p; hp.hap_off = db->d_hap_off.p; hp.hap_bit = lb->d_hap_bit.p; hp.sp_p = lb->d_p.p;
            hp.cov = db->d_cov.p; hp.node_len = db->d_node_len.p; hp.ratio = lb->d_ratio.p; hp.max_haps = mh;
        }
        PTX_TRY(sample_sort_nodes(ctx, lb->d_ab.p, lb->d_mask.p, db->d_node_base.p, S, max_vs, V, dbm->d_row16.p, pack_shift >= 0 ? (uint64_t *)nullptr : ka[0].p,
                                  pack_shift >= 0 ? ka[0].p : ka[1].p, pack_shift >= 0 ? ka[1].p : ka[2].p, pack_shift, dbm->d_ss_ws.p, d_n, &pat, masks_in_sort ? &hp : nullptr));
    } else if (use_sample) {   // few rows: sample sort (6 launches) instead of 10+ radix passes of 3 launches each
        PTX_HIP(ctx, dbm->d_ss_ws.alloc(sample_sort_ws_elems(V)));
        PTX_TRY(sample_sort3(ctx, A, B, V, dbm->d_ss_ws.p, d_n));
    } else {
        std::vector<SortPass> passes;
        if (pack_shift >= 0) {
            add_passes(passes, 1, 0, 63);                      // a > 0: sign bit clear
            add_passes(passes, 0, 0, pmax_bound + sp_bits);    // mask bits that can be in use, then the species
        } else {
            add_passes(passes, 2, 0, 63);
            add_passes(passes, 1, 0, pmax_bound);
            if (S > 1) add_passes(passes, 0, 0, sp_bits);
        }
        PTX_TRY(radix_sort(ctx, A, B, V, passes.data(), (int)passes.size(), table.p, scan_tmp.p, &in_b, d_n));
    }
    SortBufs Sd = in_b ? B : A;
    lb->row_a = reinterpret_cast<const double *>(Sd.k[pack_shift >= 0 ? 1 : 2]);   // sorted abundances, used in place
    if (!use_nodes) {
        const uint64_t *pk1 = pack_shift >= 0 ? (const uint64_t *)nullptr : Sd.k[1];
        PTX_TRY(exclusive_scan_fn(ctx, PatLoad{d_n, Sd.k[0], pk1},
                                  PatStore{Sd.k[0], pk1, (uint32_t)k_cap, pack_shift, lb->d_pat_mask.p, lb->d_pat_start.p, lb->d_pat_species.p, d_ovf},
                                  V, d_K, "scan_chained_kernel<Pat>"));
        hipLaunchKernelGGL(sp_pat_off_kernel, dim3((S + 1 + 255) / 256), dim3(256), 0, ctx->stream, S, d_K, (uint32_t)k_cap, lb->d_pat_species.p, d_n,
                           lb->d_pat_start.p, lb->d_sp_pat_off.p);
    }
    if (wide)
        hipLaunchKernelGGL(wide_pattern_kernel, dim3(lb->n_wide * WIDE_CHUNKS), dim3(256), 0, ctx->stream, lb->d_wide_list.p, lb->d_wide_off.p,
                           lb->d_wide_nw.p, db->d_node_base.p, lb->d_p.p, lb->d_ab.p, (const unsigned long long *)lb->d_mask.p, (const unsigned long long *)lb->d_maskw.p,
                           lb->d_sp_pat_off.p, lb->d_pat_mask.p, (unsigned long long *)lb->d_pat_or.p, (unsigned long long *)lb->d_pat_and.p);
    PTX_HIP(ctx, lb->d_pat_eps.alloc(k_cap)); PTX_HIP(ctx, lb->d_sc_s.alloc(k_cap)); PTX_HIP(ctx, lb->d_sc_rho.alloc(k_cap));
    PTX_HIP(ctx, lb->d_sc_lo.alloc(k_cap)); PTX_HIP(ctx, lb->d_sc_up.alloc(k_cap)); PTX_HIP(ctx, lb->d_ls_lo.alloc(k_cap));
    PTX_HIP(ctx, lb->d_ls_hi.alloc(k_cap)); PTX_HIP(ctx, lb->d_ls_mid.alloc(k_cap));
    if (lb->n_huge) PTX_HIP(ctx, lb->d_pat_act.alloc(k_cap));
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// a9 / a13 decisions on the device, so that the strain step never waits for the host between its stages.
// Same arithmetic as the host reporting code in api_strain.cpp (IEEE f64, no contraction-sensitive forms).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double d_round2(double x) { return round(x * 100.0) / 100.0; }   // f64::round: half away from zero

// first_filter_paths (profile.rs:1080-1227): which haplotypes become LP columns.  One thread per species.
__global__ void __launch_bounds__(64) first_filter_kernel(uint32_t S, const uint8_t *__restrict__ active, const uint64_t *__restrict__ hap_off,
                                                          const uint64_t *__restrict__ hto, const uint32_t *__restrict__ nnz,
                                                          const double *__restrict__ meanf, const uint8_t *__restrict__ all_same, double fr,
                                                          int shift, int32_t *__restrict__ hap_bit, int32_t *__restrict__ sp_p,
                                                          uint32_t *__restrict__ hap_nt, uint8_t *__restrict__ sp_trio) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    const uint64_t h0 = hap_off[s], h1 = hap_off[s + 1];
    for (uint64_t h = h0; h < h1; ++h) { hap_bit[h] = -1; hap_nt[h] = (uint32_t)(hto[h + 1] - hto[h]); }
    sp_trio[s] = hto[h1] != hto[h0];
    int p = 0;
    if (h1 > h0 && !(active && !active[s])) {
        const uint64_t Hs = h1 - h0, Us = hto[h1] - hto[h0];
        if (Hs != 1 && Us != 0) {                                          // :1098
            for (uint64_t h = h0; h < h1; ++h) {
                const uint64_t nt = hto[h + 1] - hto[h];
                if (nt == 0) continue;                                     // :1119
                const double frac = (double)nnz[h] / (double)nt;           // :1135
                const double fm = meanf[h];
                if (shift) {                                               // :1140-1165
                    double sh;
                    if (fm >= 1.0) { sh = fr + (0.8 - fr) * fm / 100.0; if (sh > 0.8) sh = 0.8; } else sh = fr * fm;
                    if (frac < sh) continue;
                } else if (frac < fr) continue;                            // :1168
                hap_bit[h] = p++;
            }
        } else if (Hs == 1 || all_same[s]) { hap_bit[h0] = 0; p = 1; }     // :1191-1205, :1211-1224
        else { for (uint64_t h = h0; h < h1; ++h) hap_bit[h] = p++; }      // :1208 (any number of columns: see lad_prepare)
    }
    sp_p[s] = p;
}

// second_filter_paths (profile.rs:1229-1285): which columns are pinned to zero in the second solve
struct SecondFilterArgs {
    const uint64_t *hap_off;
    const uint32_t *hap_nt;      // [H] unique-trio rows per haplotype, [S] any in the species: the first filter's copies
    const uint8_t *sp_trio;
    const int32_t *hap_bit, *sp_p;
    const uint32_t *nnz;
    const double *meanf;
    const unsigned long long *ratio;
    const double *x1;
    const int32_t *status1;
    double fc, sr;
    uint8_t *fixed2, *need2;
};
// second_filter_paths decisions of one species (profile.rs:1234-1268): which LP columns are pinned to zero in
// the second solve, and whether there is a second solve at all
__device__ __forceinline__ void second_filter_species(const SecondFilterArgs &F, uint32_t s) {
    const uint64_t h0 = F.hap_off[s], h1 = F.hap_off[s + 1];
    uint8_t need = 0;
    for (uint64_t h = h0; h < h1; ++h) F.fixed2[h] = 0;   // column k of the species lives at h0 + k
    if (F.sp_p[s] > 0 && F.status1[s] == 0 && (h1 - h0) != 1 && F.sp_trio[s]) {
        for (uint64_t h = h0; h < h1; ++h) {
            const int k = F.hap_bit[h];
            if (k < 0) continue;
            const double fm = F.meanf[h];
            bool keep = false;
            if (fm != 0.0) {                                               // :1238
                const double sol = F.x1[h0 + k];
                const double f = d_round2(fabs(sol - fm) / (sol + fm));
                if (f > F.fc) {
                    if (f <= 0.6) {
                        const double frac_r = d_round2((double)F.nnz[h] / (double)F.hap_nt[h]);
                        const float cov = (float)F.ratio[(h0 + k) * 2], len = (float)F.ratio[(h0 + k) * 2 + 1];
                        const double sc = frac_r * (double)(cov / len);
                        if (!(sc < F.sr || sol == 0.0)) keep = true;       // rescue
                    }
                } else if (sol != 0.0) keep = true;
            }
            if (!keep) { F.fixed2[h0 + k] = 1; need = 1; }
        }
    }
    F.need2[s] = need;
}
int first_filter_launch(Ctx *ctx, const Db *db, LadBatch *lb, const uint8_t *d_active, const FilterCfg &fc) {
    const uint32_t S = db->S;
    PTX_HIP(ctx, lb->d_hap_bit.alloc(db->H)); PTX_HIP(ctx, lb->d_p.alloc(S));
    PTX_HIP(ctx, lb->d_hap_nt.alloc(db->H ? db->H : 1)); PTX_HIP(ctx, lb->d_sp_trio.alloc(S));
    hipLaunchKernelGGL(first_filter_kernel, dim3((S + 63) / 64), dim3(64), 0, ctx->stream, S, d_active, db->d_hap_off.p, db->d_hap_trio_off.p,
                       db->d_hap_nnz.p, db->d_hap_mean.p, db->d_all_same.p, fc.fr, fc.shift, lb->d_hap_bit.p, lb->d_p.p, lb->d_hap_nt.p, lb->d_sp_trio.p);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}
static SecondFilterArgs second_filter_args(const Db *db, LadBatch *lb, const FilterCfg &fc, const double *d_x1, const int32_t *d_status1,
                                           uint8_t *d_fixed2, uint8_t *d_need2) {
    SecondFilterArgs F;
    F.hap_off = db->d_hap_off.p; F.hap_nt = lb->d_hap_nt.p; F.sp_trio = lb->d_sp_trio.p; F.hap_bit = lb->d_hap_bit.p; F.sp_p = lb->d_p.p; F.nnz = db->d_hap_nnz.p;
    F.meanf = db->d_hap_mean.p; F.ratio = lb->d_ratio.p; F.x1 = d_x1; F.status1 = d_status1; F.fc = fc.fc; F.sr = fc.sr;
    F.fixed2 = d_fixed2; F.need2 = d_need2;
    return F;
}
// ---------------------------------------------------------------------------------------------
// a12: the batched exact LAD solver
// ---------------------------------------------------------------------------------------------
constexpr int LAD_BLOCK = 256;
constexpr int LAD_KWIDE = 16;    // patterns up to which a wave searches cooperatively / the line search runs wide rounds
constexpr int LAD_KLDS = 256;    // patterns whose solver state fits the LDS arrays (72 B each); more go through global scratch
enum { C_LB = 0, C_UB = 1, C_PAT = 2, C_FIXED = 3 };

// -DLAD_PROFILE (tools/lad_phase_probe.sh builds such a library beside the product): thread 0 of every solver workgroup adds
// the 100-MHz wall clock spent in each phase of a pivot to prof[species * 16 + phase] and counts the visits in [+ 8 + phase].
#ifdef LAD_PROFILE
#define LAD_TICK(ph) do { if (tid == 0 && A.prof) { const unsigned long long now_ = wall_clock64(); A.prof[(size_t)s * 16 + (ph)] += now_ - t_prev_; \
                                                     A.prof[(size_t)s * 16 + 8 + (ph)] += 1; t_prev_ = now_; } } while (0)
#else
#define LAD_TICK(ph) do { } while (0)
#endif
struct LadArgs {
    unsigned long long *prof;   // LAD_PROFILE builds only (null otherwise)
    const double *row_a;
    const uint64_t *pat_mask;
    const uint32_t *pat_start;
    const uint32_t *sp_pat_off;
    double *pat_eps, *sc_s, *sc_rho;
    uint32_t *sc_lo, *sc_up, *ls_lo, *ls_hi, *ls_mid;
    const int32_t *sp_p;
    const uint8_t *need;    // [S] or null: solve only species with need[s] != 0
    const uint8_t *fixed;   // [H] at col_off[s] + k, or null: variables pinned to zero
    const double *amax;
    double *x_out;          // [H] at col_off[s] + k
    int32_t *status, *iters;
    const uint64_t *col_off;   // [S+1] first column slot of every species (= its first haplotype)
    // wide species (more than LAD_MAXP columns): the mask words of pattern k of species s are
    // pat_or[(wide_off[s] + k - sp_pat_off[s]) * NW ..]; W / G live in global scratch, slot wide_slot[s]
    const uint32_t *wide_list, *wide_off, *wide_slot;
    const uint64_t *pat_or, *pat_and;
    double *wide_W, *wide_G;
    // wide_off counts groups of LAD_WIDE_NW words; a species of more than LAD_WIDEP haplotypes ("huge") has wide_nw[s] > LAD_WIDE_NW
    // words per node and per pattern.  W of wide slot i starts at wide_woff[i] (G at twice that), rows of 64 * wide_nw doubles;
    // the column state of a huge species (x, c, d, ub, ... of lad_solve_body) at wide_coff[i] columns into huge_f64 (x 8) / huge_i32 (x 5);
    // pat_act[k] = the basis slot that holds pattern k, or -1 (huge species only)
    const uint32_t *wide_nw;
    const uint64_t *wide_woff, *wide_coff;
    double *huge_f64;
    int *huge_i32, *pat_act;
};

template <int PS>
struct LadShared {
    double x[PS], c[PS], lam[PS], d[PS], ub[PS], fac[PS], score[PS], deriv[PS];
    long long g[PS];
    int act_type[PS], act_jk[PS], dir[PS];
    uint32_t act_i0[PS], act_i1[PS];
    double red[LAD_BLOCK / 64];
    double red_t[LAD_BLOCK / 64];
    double xs[2][LAD_BLOCK / 64][4];   // double-buffered exchange slots of the line search: one barrier per exchange
    int red_k[LAD_BLOCK / 64];
    // control words written by one thread, read by all after a barrier
    int best, bdir, done, bj, btype, status, ent_type, piv;
    uint32_t ent_k, ent_i0, ent_i1;
    double bderiv, tmax, S_lo;
};

// number of breakpoints of pattern k crossed when moving t along the search direction
__device__ __forceinline__ uint32_t crossed(bool COOP, const RowIdx &a, double rho, double s0, double eps, uint32_t st, uint32_t en,
                                            uint32_t lo, uint32_t up, double t, uint32_t c_lo, uint32_t c_hi) {
    double sv = s0 - eps + t * rho;
    if (rho > 0) return ub_(COOP, a, up + c_lo, up + c_hi, sv) - up;           // rows a_i <= sv among [up,en)
    return lo - lb(COOP, a, lo - c_hi, lo - c_lo, sv);                         // rows a_i >= sv among [st,lo)
}

// the same count from the LDS samples alone: cmin <= crossed(...) <= cmax
__device__ __forceinline__ void crossed_range(bool COOP, const RowIdx &a, double rho, double s0, double eps, uint32_t lo, uint32_t up, double t,
                                              uint32_t c_lo, uint32_t c_hi, uint32_t &cmin, uint32_t &cmax) {
    double sv = s0 - eps + t * rho;
    uint32_t rmin, rmax;
    if (rho > 0) { bound_idx_range(COOP, a, up + c_lo, up + c_hi, sv, true, rmin, rmax); cmin = rmin - up; cmax = rmax - up; }
    else { bound_idx_range(COOP, a, lo - c_hi, lo - c_lo, sv, false, rmin, rmax); cmin = lo - rmax; cmax = lo - rmin; }
}

// COOP (per species, block-uniform) = few patterns: every wave owns whole patterns (its 64 lanes search
// cooperatively, lane 0 is the "leader" that accumulates); otherwise one thread per pattern with scalar searches.
#define LAD_BUILD_CACHE()                                                                                                   \
    {                                                                                                                      \
        if (tid == 0) {                                                                                                    \
            uint32_t off = 0;                                                                                              \
            for (uint32_t kk = 0; kk < k1 - k0; ++kk) {                                                                    \
                const double rho = L_rho[kk];                                                                              \
                const uint32_t cl = L_lslo[kk], ch = L_lshi[kk];                                                           \
                const uint32_t n = (rho != 0.0 && ch > cl) ? ch - cl : 0u;                                                 \
                L_coff[kk] = off; L_cn[kk] = n;                                                                            \
                L_crow0[kk] = rho > 0 ? L_up[kk] + cl : L_lo[kk] - ch;                                                     \
                off += n;                                                                                                  \
            }                                                                                                              \
            L_coff[k1 - k0] = off;                                                                                         \
        }                                                                                                                  \
        __syncthreads();                                                                                                   \
        const uint32_t total = L_coff[k1 - k0];                                                                            \
        for (uint32_t e = tid; e < total; e += LAD_BLOCK) {                                                                \
            uint32_t kk = 0, kh = k1 - k0;   /* last pattern whose first cached row is <= e (empty patterns repeat offsets) */  \
            while (kh - kk > 1) { const uint32_t km_ = (kk + kh) >> 1; if (L_coff[km_] <= e) kk = km_; else kh = km_; }         \
            L_cache[e] = ra.a[L_crow0[kk] + (e - L_coff[kk])];                                                             \
        }                                                                                                                  \
        __syncthreads();                                                                                                   \
        cached = true;                                                                                                     \
    }
#define PAT_LOOP(k) for (uint32_t k = k0 + (COOP ? (uint32_t)(tid >> 6) : (uint32_t)tid); k < k1; k += (COOP ? LAD_BLOCK / 64 : LAD_BLOCK))
// LDS of one solver workgroup
template <int PS, int NW>
struct LadLds {
    static constexpr uint32_t IDX_N = PS <= 16 ? 4096 : 2048;     // samples of the row index (the 64-column instance spends its LDS on W and G)
    // cached candidate rows of a line search.  The 64-column instance has no LDS to spare, but its elimination scratch G is idle
    // between two basis updates: the cache of a line search lives there (8192 rows instead of 512: with a hundred patterns the
    // sample-only rounds leave a few thousand candidates, and the exact rounds then run in LDS instead of in memory)
    static constexpr bool CACHE_IN_G = NW == 1 && PS > 16;
    static constexpr uint32_t CACHE_N = PS <= 16 ? 2048 : CACHE_IN_G ? PS * 2 * PS : 512;
    LadShared<PS> sh;
    double W[NW == 1 ? PS * PS : 1];          // wide species: W and G in global scratch (LadArgs::wide_W / wide_G)
    double G[NW == 1 ? PS * 2 * PS : 1];
    // per-pattern solver state (species with at most LAD_KLDS patterns, the usual case; else the global scratch arrays)
    double L_s[LAD_KLDS], L_rho[LAD_KLDS], L_eps[LAD_KLDS];
    uint64_t L_mask[LAD_KLDS];
    uint32_t L_lo[LAD_KLDS], L_up[LAD_KLDS], L_lslo[LAD_KLDS], L_lshi[LAD_KLDS], L_lsmid[LAD_KLDS], L_start[LAD_KLDS + 1];
    double L_idx[IDX_N];      // top level of every row search: every (1 << shift)-th row of the species' sorted rows
    double L_cache[CACHE_IN_G ? 1 : CACHE_N];
    uint32_t L_lsmid2[LAD_KLDS], L_coff[LAD_KLDS + 1], L_cn[LAD_KLDS], L_crow0[LAD_KLDS];
    // wide rounds of the line search (species with at most LAD_KWIDE patterns): crossed-count bounds of every
    // pattern at 64 pivots, and the per-wave slope contributions at those pivots
    uint32_t W_cmin[LAD_KWIDE][64], W_cmax[LAD_KWIDE][64];
    double W_acc[LAD_BLOCK / 64][2][64];
};

// USEL is a compile-time constant so that every access to the pattern state is a plain LDS (or plain global)
// instruction; a run-time choice between the two would turn them all into flat accesses.
template <int PS, bool USEL, int NW>
__device__ __forceinline__ void lad_solve_body(const LadArgs &A, LadLds<PS, NW> &m, const int s, const int p, const uint32_t k0, const uint32_t k1) {
    static_assert(NW == 1 || !USEL, "wide species keep their pattern state in global scratch");
    constexpr uint32_t IDX_N = LadLds<PS, NW>::IDX_N, CACHE_N = LadLds<PS, NW>::CACHE_N;
    // NW == 0 ("huge", more than LAD_WIDEP haplotypes): the number of mask words and with it the row length of W / G are
    // run-time values, and everything that is sized by the columns lives in global scratch -- any number of columns
    constexpr bool HUGE = NW == 0;
    const int nw = HUGE ? (int)A.wide_nw[s] : NW;
    const int ps = HUGE ? 64 * nw : PS;               // row length of W (G: 2 * ps)
    LadShared<PS> &sh = m.sh;
    double *W, *G;
    const uint64_t *patw = nullptr;   // wide: mask words of pattern k at patw + (k - k0) * nw
    double *q_x, *q_c, *q_d, *q_ub, *q_fac, *q_score, *q_deriv;
    long long *q_g;
    int *q_type, *q_jk, *q_dir;
    uint32_t *q_i0, *q_i1;
    if constexpr (NW == 1) { W = m.W; G = m.G; }
    else {
        const uint64_t wo = A.wide_woff[A.wide_slot[s]];
        W = A.wide_W + wo; G = A.wide_G + 2 * wo;
        patw = A.pat_or + (size_t)A.wide_off[s] * LAD_WIDE_NW;
    }
    if constexpr (HUGE) {
        double *f = A.huge_f64 + (size_t)A.wide_coff[A.wide_slot[s]] * 8;
        int *i32 = A.huge_i32 + (size_t)A.wide_coff[A.wide_slot[s]] * 5;
        q_x = f; q_c = f + ps; q_d = f + 2 * ps; q_ub = f + 3 * ps; q_fac = f + 4 * ps; q_score = f + 5 * ps; q_deriv = f + 6 * ps;
        q_g = reinterpret_cast<long long *>(f + 7 * (size_t)ps);
        q_type = i32; q_jk = i32 + ps; q_dir = i32 + 2 * ps;
        q_i0 = reinterpret_cast<uint32_t *>(i32 + 3 * (size_t)ps); q_i1 = reinterpret_cast<uint32_t *>(i32 + 4 * (size_t)ps);
    } else {
        q_x = sh.x; q_c = sh.c; q_d = sh.d; q_ub = sh.ub; q_fac = sh.fac; q_score = sh.score; q_deriv = sh.deriv; q_g = sh.g;
        q_type = sh.act_type; q_jk = sh.act_jk; q_dir = sh.dir; q_i0 = sh.act_i0; q_i1 = sh.act_i1;
    }
    // which slot of the basis holds pattern k (-1: none).  Up to LAD_WIDEP columns the slots are scanned; beyond, a map is kept.
    auto slot_of_pattern = [&](uint32_t k) -> int {
        if constexpr (HUGE) return A.pat_act[k];
        else { int ai = -1; for (int i = 0; i < p; ++i) if (q_type[i] == C_PAT && (uint32_t)q_jk[i] == k) ai = i; return ai; }
    };
    const uint64_t c0 = A.col_off[s];
    double *L_s = m.L_s, *L_rho = m.L_rho, *L_eps = m.L_eps, *L_idx = m.L_idx, *L_cache;
    if constexpr (LadLds<PS, NW>::CACHE_IN_G) L_cache = m.G; else L_cache = m.L_cache;
    uint64_t *L_mask = m.L_mask;
    uint32_t *L_lo = m.L_lo, *L_up = m.L_up, *L_lslo = m.L_lslo, *L_lshi = m.L_lshi, *L_lsmid = m.L_lsmid, *L_start = m.L_start,
             *L_lsmid2 = m.L_lsmid2, *L_coff = m.L_coff, *L_cn = m.L_cn, *L_crow0 = m.L_crow0;
    uint32_t (*W_cmin)[64] = m.W_cmin, (*W_cmax)[64] = m.W_cmax;
    double (*W_acc)[2][64] = m.W_acc;
    const int tid = threadIdx.x;
    const bool COOP = (k1 - k0) <= (uint32_t)LAD_KWIDE;
    const bool leader = COOP ? ((tid & 63) == 0) : true;
    constexpr bool useL = USEL;
    const uint32_t kofs = useL ? k0 : 0u;     // LDS arrays are indexed from the species' first pattern
    double *P_sc_s = useL ? L_s : A.sc_s, *P_sc_rho = useL ? L_rho : A.sc_rho, *P_pat_eps = useL ? L_eps : A.pat_eps;
    uint32_t *P_sc_lo = useL ? L_lo : A.sc_lo, *P_sc_up = useL ? L_up : A.sc_up, *P_ls_lo = useL ? L_lslo : A.ls_lo,
             *P_ls_hi = useL ? L_lshi : A.ls_hi, *P_ls_mid = useL ? L_lsmid : A.ls_mid;
    const uint64_t *P_pat_mask = useL ? L_mask : A.pat_mask;
    const uint32_t *P_pat_start = useL ? L_start : A.pat_start;
    if (useL) {
        for (uint32_t kk = k0 + tid; kk < k1; kk += LAD_BLOCK) L_mask[kk - k0] = A.pat_mask[kk];
        for (uint32_t kk = k0 + tid; kk <= k1; kk += LAD_BLOCK) L_start[kk - k0] = A.pat_start[kk];
    }
    RowIdx ra;
    ra.a = A.row_a; ra.idx = L_idx; ra.cache = L_cache; ra.c_row0 = 0; ra.c_n = 0;
    ra.row0 = A.pat_start[k0];
    {
        const uint32_t nrow = A.pat_start[k1] - ra.row0;
        uint32_t sh_ = 0;
        while (nrow && ((nrow - 1) >> sh_) + 1 > IDX_N) ++sh_;
        ra.shift = sh_;
        ra.n_idx = nrow ? ((nrow - 1) >> sh_) + 1 : 0;
        for (uint32_t j = tid; j < ra.n_idx; j += LAD_BLOCK) L_idx[j] = A.row_a[ra.row0 + (j << sh_)];
    }
    __syncthreads();
    const double tol = 1e-7;
    const double amax = A.amax[s];
    const double delta = 1e-10 * (amax > 1.0 ? amax : 1.0);
    for (uint32_t kk = k0 + tid; kk < k1; kk += LAD_BLOCK)
        P_pat_eps[kk - kofs] = delta * (0.25 + 0.5 * (double)(splitmix64(P_pat_mask[kk - kofs]) >> 11) * (1.0 / 9007199254740992.0));
    for (int j = tid; j < p; j += LAD_BLOCK) {
        // box: 0 <= x <= 1.05 * max(a) (profile.rs:1327); pinned to 0 in the second solve (:1484-1488)
        double u = (A.fixed && A.fixed[c0 + j]) ? 0.0 : 1.05 * A.amax[s];
        q_ub[j] = u;
        q_type[j] = u > 0.0 ? C_LB : C_FIXED;
        q_jk[j] = j;
        q_i0[j] = q_i1[j] = 0;
    }
    if constexpr (HUGE) for (uint32_t kk = k0 + tid; kk < k1; kk += LAD_BLOCK) A.pat_act[kk] = -1;
    for (int64_t i = tid; i < (int64_t)p * p; i += LAD_BLOCK) W[(i / p) * ps + (i % p)] = (i / p == i % p) ? 1.0 : 0.0;
    if (tid == 0) { sh.done = 0; sh.status = 0; }
    __syncthreads();
    if constexpr (NW != 1) {
        // rows of one pattern must agree in every mask word (they were grouped by a hash of the words)
        bool bad = false;
        for (uint32_t kk = tid; kk < (k1 - k0) * (uint32_t)nw; kk += LAD_BLOCK) bad |= patw[kk] != A.pat_and[(size_t)A.wide_off[s] * LAD_WIDE_NW + kk];
        if (bad) { sh.status = 7; sh.done = 1; }
        __syncthreads();
    }
    const int max_it = 200 * p + 2000;
    int it = 0;
    const bool skip = NW != 1 && sh.done;   // (block-uniform: written before the barrier above)
#ifdef LAD_PROFILE
    unsigned long long t_prev_ = wall_clock64();
#endif
    for (; it < max_it && !skip; ++it) {
        // ---- vertex of the perturbed problem: x = W c
        for (int j = tid; j < p; j += LAD_BLOCK) {
            int ty = q_type[j];
            q_c[j] = ty == C_UB ? q_ub[q_jk[j]] : ty == C_PAT ? ra.a[q_i0[j]] + P_pat_eps[q_jk[j] - kofs] : 0.0;
        }
        __syncthreads();
        if constexpr (NW == 1) {
            if (tid < p) {
                double v = 0.0;
                for (int i = 0; i < p; ++i) v += W[tid * ps + i] * q_c[i];
                q_x[tid] = v;
                q_g[tid] = 0;
            }
        } else {
            // W lives in global memory: a wave per row, lanes along the row (coalesced), instead of a thread per row -- four rows of a wave in flight
            // (round 6: their loads overlap; every row's sum keeps its order of additions: same bits)
            constexpr int XR = 4;
            for (int j0 = (tid >> 6) * XR; j0 < p; j0 += (LAD_BLOCK / 64) * XR) {
                double v[XR];
#pragma unroll
                for (int r = 0; r < XR; ++r) v[r] = 0.0;
                for (int i = tid & 63; i < p; i += 64) {
                    const double ci = q_c[i];
#pragma unroll
                    for (int r = 0; r < XR; ++r) if (j0 + r < p) v[r] += W[(j0 + r) * ps + i] * ci;
                }
#pragma unroll
                for (int r = 0; r < XR; ++r) {
                    const double vr = wave_reduce(v[r], [](double x_, double y_) { return x_ + y_; });
                    if ((tid & 63) == 0 && j0 + r < p) { q_x[j0 + r] = vr; q_g[j0 + r] = 0; }
                }
            }
        }
        __syncthreads();
        // ---- pattern pass: position of every pattern, integer sub-gradient g = sum sigma_k m_k
        PAT_LOOP(k) {
            uint64_t mk = P_pat_mask[k - kofs];
            const uint64_t *mw = NW != 1 ? patw + (size_t)(k - k0) * nw : nullptr;
            uint32_t st = P_pat_start[k - kofs], en = P_pat_start[(k + 1) - kofs];
            const int ai = slot_of_pattern(k);
            uint32_t lo, up; double sk;
            if (ai >= 0) { lo = q_i0[ai]; up = q_i1[ai]; sk = ra.a[lo] + P_pat_eps[k - kofs]; }
            else {
                if constexpr (NW == 1) sk = mdot(mk, q_x); else sk = mdotx<NW>(mw, nw, q_x);
                double sv = sk - P_pat_eps[k - kofs];
                lo = lb(COOP, ra, st, en, sv);
                up = ub_(COOP, ra, lo, en, sv);
            }
            P_sc_s[k - kofs] = sk; P_sc_lo[k - kofs] = lo; P_sc_up[k - kofs] = up;
            long long sigma = (long long)(lo - st) - (long long)(en - up);
            if (sigma && leader) {
                if constexpr (NW == 1) {
                    uint64_t bits = mk;
                    while (bits) { int j = __ffsll((long long)bits) - 1; bits &= bits - 1; atomicAdd((unsigned long long *)&q_g[j], (unsigned long long)sigma); }
                } else {
                    for (int w = 0; w < nw; ++w) {
                        uint64_t bits = mw[w];
                        while (bits) { int j = 64 * w + __ffsll((long long)bits) - 1; bits &= bits - 1; atomicAdd((unsigned long long *)&q_g[j], (unsigned long long)sigma); }
                    }
                }
            }
        }
        __syncthreads();
        LAD_TICK(0);
        // ---- multipliers lam_i = -g . W[:,i]; steepest-edge choice of the constraint to relax
        for (int i = tid; i < p; i += LAD_BLOCK) {
            double sdot = 0.0, nrm = 0.0;
            if constexpr (NW == 1) {
                for (int j = 0; j < p; ++j) { double w = W[j * ps + i]; sdot += (double)q_g[j] * w; nrm += w * w; }
            } else {   // W in global memory: eight rows' loads in flight, the sums in the same order (same bits)
                int j = 0;
                for (; j + 8 <= p; j += 8) {
                    double w8[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) w8[r] = W[(j + r) * ps + i];
#pragma unroll
                    for (int r = 0; r < 8; ++r) { sdot += (double)q_g[j + r] * w8[r]; nrm += w8[r] * w8[r]; }
                }
                for (; j < p; ++j) { double w = W[j * ps + i]; sdot += (double)q_g[j] * w; nrm += w * w; }
            }
            double lam = -sdot; nrm = sqrt(nrm);
            double deriv = 0.0; int dir = 0; int ty = q_type[i];
            if (ty == C_PAT) {
                double w = (double)(q_i1[i] - q_i0[i]);
                if (lam > w + tol) { dir = +1; deriv = w - lam; } else if (lam < -w - tol) { dir = -1; deriv = w + lam; }
            } else if (ty == C_LB) { if (lam > tol) { dir = +1; deriv = -lam; } }
            else if (ty == C_UB) { if (lam < -tol) { dir = -1; deriv = lam; } }
            q_dir[i] = dir; q_deriv[i] = deriv; q_score[i] = dir ? deriv / nrm : 0.0;
        }
        __syncthreads();
        if (tid == 0) {
            int best = -1; double bs = -tol;
            for (int i = 0; i < p; ++i) if (q_dir[i] && q_score[i] < bs) { bs = q_score[i]; best = i; }
            sh.best = best;
            if (best < 0) sh.done = 1; else { sh.bdir = q_dir[best]; sh.bderiv = q_deriv[best]; }
        }
        __syncthreads();
        if (sh.done) break;
        const int best = sh.best; const double bdir = (double)sh.bdir;
        for (int j = tid; j < p; j += LAD_BLOCK) q_d[j] = bdir * W[j * ps + best];
        __syncthreads();
        if (tid == 0) {   // ratio test against the box
            double tmax = INFINITY; int bj = -1, bt = C_LB;
            for (int j = 0; j < p; ++j) {
                if (q_ub[j] <= 0.0) continue;
                double dj = q_d[j];
                if (dj < -1e-12) { double t = q_x[j] / (-dj); if (t < 0) t = 0; if (t < tmax) { tmax = t; bj = j; bt = C_LB; } }
                else if (dj > 1e-12) { double t = (q_ub[j] - q_x[j]) / dj; if (t < 0) t = 0; if (t < tmax) { tmax = t; bj = j; bt = C_UB; } }
            }
            sh.tmax = tmax; sh.bj = bj; sh.btype = bt;
        }
        LAD_TICK(1);
        // ---- line search set-up: rate rho_k of every pattern along d
        double part = 0.0;
        PAT_LOOP(k) {
            uint64_t mk = P_pat_mask[k - kofs];
            const int ai = slot_of_pattern(k);
            double rho;
            if (ai >= 0) rho = (ai == best) ? bdir : 0.0;   // other tight patterns stay tight: n_i . d = 0
            else { if constexpr (NW == 1) rho = mdot(mk, q_d); else rho = mdotx<NW>(patw + (size_t)(k - k0) * nw, nw, q_d); if (fabs(rho) < 1e-12) rho = 0.0; if (leader) part += fabs(rho) * (double)(P_sc_up[k - kofs] - P_sc_lo[k - kofs]); }
            P_sc_rho[k - kofs] = rho;
            P_ls_lo[k - kofs] = 0;
            P_ls_hi[k - kofs] = rho > 0 ? P_pat_start[(k + 1) - kofs] - P_sc_up[k - kofs] : rho < 0 ? P_sc_lo[k - kofs] - P_pat_start[k - kofs] : 0;
        }
        double S0 = sh.bderiv + block_sum_f64<LAD_BLOCK>(part, sh.red);   // slope just after t = 0
        if (tid == 0) { sh.ent_type = -1; sh.S_lo = S0; }
        __syncthreads();
        const double tmax = sh.tmax;
        LAD_TICK(2);
        if (S0 >= -tol) {
            // degenerate: an unsplit tie group blocks the move at t = 0 -> it enters (step length 0)
            double tb = INFINITY; int kb = 0x7fffffff;
            PAT_LOOP(k)
                if (P_sc_rho[k - kofs] != 0.0 && P_sc_up[k - kofs] > P_sc_lo[k - kofs]) { const int ai = slot_of_pattern(k); if (ai < 0 && (int)k < kb) { kb = (int)k; tb = 0.0; } }
            kb = wave_reduce(kb, [](int x, int y) { return x < y ? x : y; });
            if ((tid & 63) == 0) sh.red_k[tid >> 6] = kb;
            __syncthreads();
            if (tid == 0) {
                int kk = sh.red_k[0];
                for (int w = 1; w < LAD_BLOCK / 64; ++w) if (sh.red_k[w] < kk) kk = sh.red_k[w];
                if (kk != 0x7fffffff) { sh.ent_type = C_PAT; sh.ent_k = (uint32_t)kk; sh.ent_i0 = P_sc_lo[kk - kofs]; sh.ent_i1 = P_sc_up[kk - kofs]; }
                else { sh.status = 4; sh.done = 1; }
            }
            (void)tb;
            __syncthreads();
        } else {
            // ---- bracket: find t_hi with slope(t_hi) >= -tol (or the box bound enters)
            double t_hi = isfinite(tmax) ? tmax : (amax > 1.0 ? amax : 1.0);
            double S_hi = 0.0;
            bool bound_enters = false;
            for (int grow = 0; grow < 200; ++grow) {
                double acc = 0.0;
                PAT_LOOP(k) {
                    double rho = P_sc_rho[k - kofs];
                    if (rho == 0.0) continue;
                    uint32_t st = P_pat_start[k - kofs], en = P_pat_start[(k + 1) - kofs];
                    uint32_t cmax = rho > 0 ? en - P_sc_up[k - kofs] : P_sc_lo[k - kofs] - st;
                    uint32_t c = crossed(COOP, ra, rho, P_sc_s[k - kofs], P_pat_eps[k - kofs], st, en, P_sc_lo[k - kofs], P_sc_up[k - kofs], t_hi, 0, cmax);
                    P_ls_hi[k - kofs] = c;
                    if (leader) acc += fabs(rho) * 2.0 * (double)c;
                }
                S_hi = S0 + block_sum_f64<LAD_BLOCK>(acc, sh.red);
                if (S_hi >= -tol) break;
                if (isfinite(tmax)) { bound_enters = true; break; }
                t_hi *= 4.0;
                if (grow == 199) bound_enters = true;   // cannot happen: slope(inf) >= 0 for a LAD objective
            }
            if (bound_enters) {
                if (tid == 0) {
                    if (sh.bj >= 0) { sh.ent_type = sh.btype; sh.ent_k = (uint32_t)sh.bj; }
                    else { sh.status = 2; sh.done = 1; }
                }
                __syncthreads();
            } else {
                LAD_TICK(3);
                // ---- narrow the bracket.  Rounds alternate between two pivots so that the search is
                // both scale-free and robust to many patterns: (even) the median remaining breakpoint of
                // the pattern that still holds the most weighted candidates, (odd) the midpoint in t.
                // ls_lo/ls_hi bracket the crossed-count of every pattern; slope(t_lo) < 0 <= slope(t_hi).
                double t_lo = 0.0, S_lo = S0;
                unsigned long long prev_cand = ~0ull; int stall = 0;
                // Rounds first run on the LDS samples alone (bounds on every crossed-count, no memory access) for
                // as long as the sign of the slope at the pivot is certain; then the remaining candidate rows of
                // every pattern are copied to LDS once and the exact rounds finish there.
                bool approx = useL, cached = false, wide_done = false;   // wide_done: the wide rounds left only bracket-end tie groups
                int xp = 0;   // exchange buffer in use
#define RK(k) RowIdx rk = ra; if (cached) { rk.cache = L_cache + L_coff[(k) - k0]; rk.c_row0 = L_crow0[(k) - k0]; rk.c_n = L_cn[(k) - k0]; }
                if (COOP && useL) {
                    // ---- wide rounds: the slope is bounded (samples) or evaluated (cached rows) at 64 pivots at once.
                    // Pivots are evenly spaced candidates of the pattern that holds the most weighted candidates,
                    // lane order = increasing t; every lane searches each pattern of its wave for its own pivot; the
                    // bracket moves to the last pivot that is certainly before the minimiser and the first that is
                    // certainly at or past it.  ~64x fewer candidates per round instead of 2x.
                    const int lane = tid & 63, wave = tid >> 6;
                    bool exact = false;
                    unsigned long long prev_w = ~0ull;
                    for (int wr = 0; wr < 64; ++wr) {
                        double cs = 0.0, wb = 0.0;
                        int kbest = 0x7fffffff;
                        PAT_LOOP(k) {
                            const double rho = P_sc_rho[k - kofs];
                            const uint32_t cl = P_ls_lo[k - kofs], ch = P_ls_hi[k - kofs];
                            if (rho == 0.0 || ch <= cl) continue;
                            cs += (double)(ch - cl);
                            const double w = fabs(rho) * (double)(ch - cl);
                            if (w > wb || (w == wb && (int)k < kbest)) { wb = w; kbest = (int)k; }
                        }
                        if (lane == 0) { sh.xs[xp][wave][0] = cs; sh.xs[xp][wave][1] = wb; sh.xs[xp][wave][2] = (double)kbest; }
                        __syncthreads();
                        cs = 0.0; wb = 0.0; kbest = 0x7fffffff;
#pragma unroll
                        for (int w = 0; w < LAD_BLOCK / 64; ++w) {
                            cs += sh.xs[xp][w][0];
                            const double w2 = sh.xs[xp][w][1]; const int k2 = (int)sh.xs[xp][w][2];
                            if (w2 > wb || (w2 == wb && w2 > 0.0 && k2 < kbest)) { wb = w2; kbest = k2; }
                        }
                        xp ^= 1;
                        const unsigned long long cand = (unsigned long long)cs;
                        if (cand <= 8 || kbest == 0x7fffffff) { wide_done = true; break; }
                        // exact rounds that cannot move the bracket any more: what is left sits on its ends (tie groups)
                        if (cand == prev_w) { if (exact) { wide_done = true; break; } exact = true; }
                        prev_w = cand;
                        if (exact && !cached) { if (cand > CACHE_N) break; LAD_BUILD_CACHE() }
                        // pivots of the heaviest pattern
                        const uint32_t kq = (uint32_t)kbest - kofs;
                        const double rho_b = P_sc_rho[kq], s_b = P_sc_s[kq], eps_b = P_pat_eps[kq];
                        const uint32_t rl = rho_b > 0 ? P_sc_up[kq] + P_ls_lo[kq] : P_sc_lo[kq] - P_ls_hi[kq];
                        const uint32_t rh = rho_b > 0 ? P_sc_up[kq] + P_ls_hi[kq] : P_sc_lo[kq] - P_ls_lo[kq];
                        uint32_t pa, pn;   // pivot positions [pa, pa + pn): sample indices or rows
                        if (!exact) {
                            const uint32_t msk = (1u << ra.shift) - 1u;
                            pa = (rl - ra.row0 + msk) >> ra.shift;
                            const uint32_t pb = (rh - ra.row0 + msk) >> ra.shift;
                            pn = pb > pa ? pb - pa : 0u;
                            if (pn < 2) { exact = true; prev_w = ~0ull; continue; }   // too few samples left among the candidates
                        } else { pa = rl; pn = rh - rl; }
                        const uint32_t pm = pn < 64u ? pn : 64u;
                        double t_piv = 0.0;
                        bool valid = (uint32_t)lane < pm;
                        if (valid) {
                            const uint32_t q = (uint32_t)(((uint64_t)(2 * lane + 1) * pn) / (2ull * pm));
                            const uint32_t pos = rho_b > 0 ? pa + q : pa + pn - 1 - q;
                            double av;
                            if (!exact) av = ra.idx[pos];
                            else { RK(kbest); av = row_val(rk, pos); }
                            t_piv = (av + eps_b - s_b) / rho_b;
                            if (t_piv < 0) t_piv = 0;
                            valid = t_piv > t_lo && t_piv < t_hi;
                        }
                        double a0 = 0.0, a1 = 0.0;
                        PAT_LOOP(k) {
                            const double rho = P_sc_rho[k - kofs];
                            if (rho == 0.0) continue;
                            uint32_t cmin = 0, cmax = 0;
                            if (valid) {
                                if (!exact) crossed_range(false, ra, rho, P_sc_s[k - kofs], P_pat_eps[k - kofs], P_sc_lo[k - kofs], P_sc_up[k - kofs], t_piv,
                                                          P_ls_lo[k - kofs], P_ls_hi[k - kofs], cmin, cmax);
                                else {
                                    RK(k);
                                    cmin = cmax = crossed(false, rk, rho, P_sc_s[k - kofs], P_pat_eps[k - kofs], P_pat_start[k - kofs], P_pat_start[(k + 1) - kofs],
                                                          P_sc_lo[k - kofs], P_sc_up[k - kofs], t_piv, P_ls_lo[k - kofs], P_ls_hi[k - kofs]);
                                }
                            }
                            W_cmin[k - k0][lane] = cmin; W_cmax[k - k0][lane] = cmax;
                            a0 += fabs(rho) * 2.0 * (double)cmin; a1 += fabs(rho) * 2.0 * (double)cmax;
                        }
                        W_acc[wave][0][lane] = a0; W_acc[wave][1][lane] = a1;
                        __syncthreads();
                        double S_min = S0, S_max = S0;
#pragma unroll
                        for (int w = 0; w < LAD_BLOCK / 64; ++w) { S_min += W_acc[w][0][lane]; S_max += W_acc[w][1][lane]; }
                        const unsigned long long mlo = __ballot(valid && S_max < -tol), mhi = __ballot(valid && S_min >= -tol);
                        const int jl = mlo ? 63 - __clzll((long long)mlo) : -1;
                        const int jh = mhi ? __ffsll((long long)mhi) - 1 : -1;
                        if ((jl < 0 && jh < 0) || (jl >= 0 && jh >= 0 && jl >= jh)) { if (exact) { wide_done = true; break; } exact = true; prev_w = ~0ull; continue; }
                        if (jl >= 0) {
                            t_lo = __shfl(t_piv, jl); S_lo = __shfl(S_max, jl);
                            PAT_LOOP(k) if (P_sc_rho[k - kofs] != 0.0) P_ls_lo[k - kofs] = W_cmin[k - k0][jl];
                        }
                        if (jh >= 0) {
                            t_hi = __shfl(t_piv, jh); S_hi = __shfl(S_min, jh);
                            PAT_LOOP(k) if (P_sc_rho[k - kofs] != 0.0) P_ls_hi[k - kofs] = W_cmax[k - k0][jh];
                        }
                    }
                    if (exact) approx = false;
                    __syncthreads();   // bracket state written by the owning waves is read by everyone below
                }
                LAD_TICK(4);
                // After finished wide rounds the walk starts at once (few tie groups are left); should it not close within
                // a few dozen groups, the binary rounds narrow further and the walk runs again without a cap.
                for (int attempt = 0; attempt < 2; ++attempt) {
                const bool quick = attempt == 0 && wide_done;
                for (int bi = 0; bi < 400 && !quick; ++bi) {
                    double wbest = 0.0, tprop = 0.0; unsigned long long cand = 0;
                    PAT_LOOP(k) {
                        double rho = P_sc_rho[k - kofs];
                        uint32_t cl = P_ls_lo[k - kofs], ch = P_ls_hi[k - kofs];
                        if (rho == 0.0 || ch <= cl) continue;
                        if (leader) cand += ch - cl;
                        double w = fabs(rho) * (double)(ch - cl);
                        if (w > wbest) {
                            uint32_t mth = cl + (ch - cl) / 2;   // mth breakpoint ahead (0-based) of this pattern
                            uint32_t r = rho > 0 ? P_sc_up[k - kofs] + mth : P_sc_lo[k - kofs] - 1 - mth;
                            double av;
                            bool have = false;
                            if (approx) {   // a sampled row among the candidates serves as well as the exact median
                                const uint32_t rl = rho > 0 ? P_sc_up[k - kofs] + cl : P_sc_lo[k - kofs] - ch;
                                const uint32_t rh = rho > 0 ? P_sc_up[k - kofs] + ch : P_sc_lo[k - kofs] - cl;
                                uint32_t j = (r - ra.row0) >> ra.shift;
                                uint32_t rs = ra.row0 + (j << ra.shift);
                                if (rs < rl) { ++j; rs += 1u << ra.shift; }
                                if (rs < rh && j < ra.n_idx) { av = ra.idx[j]; have = true; }
                            }
                            if (!have) { RK(k); av = row_val(rk, r); }
                            double t = (av + P_pat_eps[k - kofs] - P_sc_s[k - kofs]) / rho;
                            wbest = w; tprop = t < 0 ? 0 : t;
                        }
                    }
                    // one exchange: candidate count (sum) and the heaviest proposal (max over the block, ties -> smaller t)
                    double bt;
                    {
                        double cs = wave_reduce((double)cand, [](double x, double y) { return x + y; });
                        wave_reduce_pair(wbest, tprop, [](double w2, double t2, double w, double t) { return w2 > w || (w2 == w && t2 < t); });
                        if ((tid & 63) == 0) { sh.xs[xp][tid >> 6][0] = cs; sh.xs[xp][tid >> 6][1] = wbest; sh.xs[xp][tid >> 6][2] = tprop; }
                        __syncthreads();
                        cs = 0.0;
                        double bw = sh.xs[xp][0][1];
                        bt = sh.xs[xp][0][2];
#pragma unroll
                        for (int w = 0; w < LAD_BLOCK / 64; ++w) {
                            cs += sh.xs[xp][w][0];
                            const double w2 = sh.xs[xp][w][1], t2 = sh.xs[xp][w][2];
                            if (w > 0 && (w2 > bw || (w2 == bw && t2 < bt))) { bw = w2; bt = t2; }
                        }
                        xp ^= 1;
                        cand = (unsigned long long)cs;
                    }
                    if (cand <= 8) break;
                    // three exact rounds (one of each pivot kind) without shrinking: only tie groups remain -> walk them
                    const bool shrunk = cand != prev_cand;
                    prev_cand = cand;
                    if (approx) { if (!shrunk) approx = false; }
                    else {
                        stall = shrunk ? 0 : stall + 1;
                        if (stall >= 3) break;
                    }
                    if (!approx && !cached && useL && cand <= CACHE_N) LAD_BUILD_CACHE()
                    double t_mid;
                    const int mode = bi % 3;   // 0: heaviest pattern's median breakpoint, 1: secant on the slope, 2: midpoint
                    if (mode == 0) { t_mid = bt; if (!(t_mid >= t_lo && t_mid <= t_hi)) t_mid = 0.5 * (t_lo + t_hi); }
                    else {
                        const double den = S_hi - S_lo;
                        t_mid = (mode == 1 && den > 0.0) ? t_lo + (t_hi - t_lo) * (-S_lo / den) : 0.5 * (t_lo + t_hi);
                        if (!(t_mid > t_lo && t_mid < t_hi)) t_mid = 0.5 * (t_lo + t_hi);
                    }
                    if (approx) {
                        double acc0 = 0.0, acc1 = 0.0;
                        PAT_LOOP(k) {
                            double rho = P_sc_rho[k - kofs];
                            if (rho == 0.0) continue;
                            uint32_t cmin, cmax;
                            crossed_range(COOP, ra, rho, P_sc_s[k - kofs], P_pat_eps[k - kofs], P_sc_lo[k - kofs], P_sc_up[k - kofs], t_mid,
                                          P_ls_lo[k - kofs], P_ls_hi[k - kofs], cmin, cmax);
                            P_ls_mid[k - kofs] = cmin; L_lsmid2[k - k0] = cmax;
                            if (leader) { acc0 += fabs(rho) * 2.0 * (double)cmin; acc1 += fabs(rho) * 2.0 * (double)cmax; }
                        }
                        acc0 = wave_reduce(acc0, [](double x, double y) { return x + y; });
                        acc1 = wave_reduce(acc1, [](double x, double y) { return x + y; });
                        if ((tid & 63) == 0) { sh.xs[xp][tid >> 6][0] = acc0; sh.xs[xp][tid >> 6][1] = acc1; }
                        __syncthreads();
                        double S_min = S0, S_max = S0;
#pragma unroll
                        for (int w = 0; w < LAD_BLOCK / 64; ++w) { S_min += sh.xs[xp][w][0]; S_max += sh.xs[xp][w][1]; }
                        xp ^= 1;
                        if (S_max < -tol) {          // certainly still descending at t_mid
                            PAT_LOOP(k) if (P_sc_rho[k - kofs] != 0.0) P_ls_lo[k - kofs] = P_ls_mid[k - kofs];
                            t_lo = t_mid; S_lo = S_max;
                            continue;
                        }
                        if (S_min >= -tol) {         // certainly past the minimiser
                            PAT_LOOP(k) if (P_sc_rho[k - kofs] != 0.0) P_ls_hi[k - kofs] = L_lsmid2[k - k0];
                            t_hi = t_mid; S_hi = S_min;
                            continue;
                        }
                        approx = false;              // undecided at sample resolution: exact from here on (same pivot)
                    }
                    double acc = 0.0;
                    PAT_LOOP(k) {
                        double rho = P_sc_rho[k - kofs];
                        if (rho == 0.0) continue;
                        RK(k);
                        uint32_t c = crossed(COOP, rk, rho, P_sc_s[k - kofs], P_pat_eps[k - kofs], P_pat_start[k - kofs], P_pat_start[(k + 1) - kofs], P_sc_lo[k - kofs], P_sc_up[k - kofs],
                                             t_mid, P_ls_lo[k - kofs], P_ls_hi[k - kofs]);
                        P_ls_mid[k - kofs] = c;
                        if (leader) acc += fabs(rho) * 2.0 * (double)c;
                    }
                    acc = wave_reduce(acc, [](double x, double y) { return x + y; });
                    if ((tid & 63) == 0) sh.xs[xp][tid >> 6][0] = acc;
                    __syncthreads();
                    double S_mid = S0;
#pragma unroll
                    for (int w = 0; w < LAD_BLOCK / 64; ++w) S_mid += sh.xs[xp][w][0];
                    xp ^= 1;
                    bool go_hi = S_mid >= -tol;
                    PAT_LOOP(k) {
                        if (P_sc_rho[k - kofs] == 0.0) continue;
                        if (go_hi) P_ls_hi[k - kofs] = P_ls_mid[k - kofs]; else P_ls_lo[k - kofs] = P_ls_mid[k - kofs];
                    }
                    if (go_hi) { t_hi = t_mid; S_hi = S_mid; } else { t_lo = t_mid; S_lo = S_mid; }
                }
                (void)S_hi; (void)S_lo;
                LAD_TICK(5);
                // slope of the crossed set the walk starts from (sample-only rounds leave lower bounds in ls_lo, so
                // it is recomputed from the counts; with exact counts it equals the slope at t_lo)
                {
                    double acc = 0.0;
                    PAT_LOOP(k) { double rho = P_sc_rho[k - kofs]; if (rho != 0.0 && leader) acc += fabs(rho) * 2.0 * (double)P_ls_lo[k - kofs]; }
                    const double S_start = S0 + block_sum_f64<LAD_BLOCK>(acc, sh.red);
                    if (tid == 0) sh.S_lo = S_start;
                }
                __syncthreads();
                // ---- walk the few remaining breakpoint groups in order of t.  ls_lo may split a tie group (sample
                // bounds are not group aligned): a step crosses the rest of the group its first uncrossed row is in.
                for (int step = 0, cap = quick ? 48 : 4096; step < cap; ++step) {
                    double tb = INFINITY; int kb = 0x7fffffff;
                    PAT_LOOP(k) {
                        double rho = P_sc_rho[k - kofs];
                        if (rho == 0.0 || P_ls_hi[k - kofs] <= P_ls_lo[k - kofs]) continue;
                        uint32_t r = rho > 0 ? P_sc_up[k - kofs] + P_ls_lo[k - kofs] : P_sc_lo[k - kofs] - 1 - P_ls_lo[k - kofs];
                        RK(k);
                        double t = (row_val(rk, r) + P_pat_eps[k - kofs] - P_sc_s[k - kofs]) / rho;
                        if (t < 0) t = 0;
                        if (t < tb || (t == tb && (int)k < kb)) { tb = t; kb = (int)k; }
                    }
                    wave_reduce_pair(tb, kb, [](double t2, int k2, double t, int k) { return t2 < t || (t2 == t && k2 < k); });
                    if ((tid & 63) == 0) { sh.red_t[tid >> 6] = tb; sh.red_k[tid >> 6] = kb; }
                    __syncthreads();
                    if (tid < 64) {   // wave 0: its lanes search together, lane 0 writes
                        const bool l0 = tid == 0;
                        double tt = sh.red_t[0]; int kk = sh.red_k[0];
                        for (int w = 1; w < LAD_BLOCK / 64; ++w) if (sh.red_t[w] < tt || (sh.red_t[w] == tt && sh.red_k[w] < kk)) { tt = sh.red_t[w]; kk = sh.red_k[w]; }
                        if (kk == 0x7fffffff) {
                            // bracket exhausted without crossing (rounding): fall back to the box bound or fail
                            if (l0) { if (sh.bj >= 0 && isfinite(sh.tmax)) { sh.ent_type = sh.btype; sh.ent_k = (uint32_t)sh.bj; } else { sh.status = 5; sh.done = 1; } }
                        } else {
                            double rho = P_sc_rho[kk - kofs];
                            uint32_t st = P_pat_start[kk - kofs], en = P_pat_start[(kk + 1) - kofs];
                            uint32_t r = rho > 0 ? P_sc_up[kk - kofs] + P_ls_lo[kk - kofs] : P_sc_lo[kk - kofs] - 1 - P_ls_lo[kk - kofs];
                            RK(kk);
                            double av = row_val(rk, r);
                            // extent [g0,g1) of the tie group of r: inside the cached rows first; a group that reaches the
                            // end of the cached rows is settled by the neighbouring row, else by the full searches
                            uint32_t g0 = r, g1 = r + 1;
                            bool open0 = true, open1 = true;
                            if (r - rk.c_row0 < rk.c_n) {
                                const uint32_t o = r - rk.c_row0;
                                g0 = rk.c_row0 + wave_bound(rk.cache, 0, o, av, false);
                                g1 = rk.c_row0 + wave_bound(rk.cache, o + 1, rk.c_n, av, true);
                                open0 = g0 == rk.c_row0 && g0 > st;
                                open1 = g1 == rk.c_row0 + rk.c_n && g1 < en;
                            }
                            if (g0 <= st) open0 = false;
                            if (g1 >= en) open1 = false;
                            const double below = open0 ? ra.a[g0 - 1] : 0.0, above = open1 ? ra.a[g1] : 0.0;
                            if (open0 && below != av) open0 = false;
                            if (open1 && above != av) open1 = false;
                            if (open0) g0 = lb(true, ra, st, g0, av);
                            if (open1) g1 = ub_(true, ra, g1, en, av);
                            uint32_t gs = rho > 0 ? g1 - r : r - g0 + 1;      // rows of the group not crossed yet
                            double Sn = sh.S_lo + 2.0 * fabs(rho) * (double)gs;
                            if (l0) {
                                P_ls_lo[kk - kofs] += gs;
                                sh.S_lo = Sn;
                                if (Sn >= -tol) { sh.ent_type = C_PAT; sh.ent_k = (uint32_t)kk; sh.ent_i0 = g0; sh.ent_i1 = g1; }
                            }
                        }
                    }
                    __syncthreads();
                    if (sh.ent_type >= 0 || sh.done) break;
                }
                if (sh.ent_type >= 0 || sh.done || !quick) break;
                }   // attempt
#undef RK
            }
        }
        LAD_TICK(6);
        if (sh.done) break;
        if (sh.ent_type < 0) { if (tid == 0) { sh.status = 6; sh.done = 1; } __syncthreads(); break; }
        // ---- pivot: constraint `best` leaves, the entering one takes its slot; W = N^-1 by Gauss-Jordan
        if (tid == 0) {
            if constexpr (HUGE) {
                if (q_type[best] == C_PAT) A.pat_act[q_jk[best]] = -1;
                if (sh.ent_type == C_PAT) A.pat_act[sh.ent_k] = best;
            }
            q_type[best] = sh.ent_type;
            q_jk[best] = (int)sh.ent_k;
            q_i0[best] = sh.ent_i0; q_i1[best] = sh.ent_i1;
        }
        __syncthreads();
        // One row of N changed: W = N^-1 follows by a rank-one (Sherman-Morrison) update, O(p^2) instead of the O(p^3)
        // elimination with its 4 barriers per column -- 110 of 200 us per pivot at p = 36.  With y = n_new^T W and
        // z = y - e_best:  W' = W - (W e_best) z^T / y[best].  The inverse is rebuilt from scratch every 16th pivot and
        // whenever y[best] is small, so rounding cannot accumulate; up to 8 columns the elimination is cheap and stays.
        // W in global memory (more than 64 columns): the O(p^3) rebuild by one workgroup costs 40 ms at 256 columns, 5 s at 1100 -- a
        // hundred pivots' worth.  There the inverse is VERIFIED every 64th pivot instead (one probe vector: |N (W v) - v|, O(p^2) like
        // a pivot) and rebuilt only if the probe fails, every 1024th pivot, or -- as everywhere -- when the update's divisor is small.
        const int rebuild_mask = NW == 1 ? 15 : 1023;
        bool refactor = p <= 8 || (it & rebuild_mask) == rebuild_mask;
        if (!refactor) {
            for (int c_ = tid; c_ < p; c_ += LAD_BLOCK) {
                double y;
                if (q_type[best] == C_PAT) {
                    y = 0.0;
                    if constexpr (NW == 1) {
                        uint64_t bits = P_pat_mask[q_jk[best] - kofs];
                        while (bits) { const int j = __ffsll((long long)bits) - 1; bits &= bits - 1; y += W[j * ps + c_]; }
                    } else {
                        const uint64_t *mb = patw + (size_t)((uint32_t)q_jk[best] - k0) * nw;
                        for (int w = 0; w < nw; ++w) {     // four rows' loads in flight, added in bit order (same bits as one at a time)
                            uint64_t bits = mb[w];
                            while (bits) {
                                int jj[4]; bool on[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) { on[r] = bits != 0ull; jj[r] = on[r] ? 64 * w + __ffsll((long long)bits) - 1 : 0; bits &= bits - 1; }
                                double t4[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) t4[r] = W[jj[r] * ps + c_];
#pragma unroll
                                for (int r = 0; r < 4; ++r) if (on[r]) y += t4[r];
                            }
                        }
                    }
                } else y = W[q_jk[best] * ps + c_];
                q_fac[c_] = y;
                q_score[c_] = W[c_ * ps + best];
            }
            __syncthreads();
            const double alpha = q_fac[best];
            if (fabs(alpha) < 1e-7) refactor = true;   // (block-uniform: read from LDS after the barrier)
            else {
                const double ainv = 1.0 / alpha;
                if constexpr (NW == 1) {
                    for (int i = tid; i < p * p; i += LAD_BLOCK) {
                        const int j = i / p, cc = i % p;
                        W[j * ps + cc] -= q_score[j] * (q_fac[cc] - (cc == best ? 1.0 : 0.0)) * ainv;
                    }
                } else {   // W in global memory: a wave per row, lanes along it -- coalesced, no divisions; every element's arithmetic is the line above
                    for (int j = tid >> 6; j < p; j += LAD_BLOCK / 64) {
                        const double sj = q_score[j];
#pragma unroll 4
                        for (int cc = tid & 63; cc < p; cc += 64) W[j * ps + cc] -= sj * (q_fac[cc] - (cc == best ? 1.0 : 0.0)) * ainv;
                    }
                }
                __syncthreads();
            }
        }
        if constexpr (NW != 1) {
            if (!refactor && (it & 63) == 63) {   // (block-uniform)
                auto probe_v = [](int i) { return 1.0 + 0.25 * (double)(i % 7); };
                for (int j = tid >> 6; j < p; j += LAD_BLOCK / 64) {          // u = W v: a wave per row
                    double a_ = 0.0;
                    for (int i = tid & 63; i < p; i += 64) a_ += W[j * ps + i] * probe_v(i);
                    a_ = wave_reduce(a_, [](double x_, double y_) { return x_ + y_; });
                    if ((tid & 63) == 0) q_fac[j] = a_;
                }
                __syncthreads();
                double worst = 0.0;
                for (int i = tid; i < p; i += LAD_BLOCK) {                    // row i of N: a pattern's membership or a unit vector
                    const double r_ = q_type[i] == C_PAT ? mdotx<NW>(patw + (size_t)((uint32_t)q_jk[i] - k0) * nw, nw, q_fac) : q_fac[q_jk[i]];
                    worst = fmax(worst, fabs(r_ - probe_v(i)));
                }
                worst = block_max_f64<LAD_BLOCK>(worst, sh.red);
                if (!(worst <= 1e-10)) refactor = true;
            }
        }
        if (refactor) {
        for (int i = tid; i < p * 2 * p; i += LAD_BLOCK) {
            int r = i / (2 * p), cc = i % (2 * p);
            double v;
            if (cc >= p) v = (cc - p == r) ? 1.0 : 0.0;
            else if (q_type[r] == C_PAT) {
                if constexpr (NW == 1) v = (P_pat_mask[q_jk[r] - kofs] >> cc) & 1ull ? 1.0 : 0.0;
                else v = (patw[(size_t)((uint32_t)q_jk[r] - k0) * nw + (cc >> 6)] >> (cc & 63)) & 1ull ? 1.0 : 0.0;
            }
            else v = (q_jk[r] == cc) ? 1.0 : 0.0;
            G[r * 2 * ps + cc] = v;
        }
        __syncthreads();
        static_assert(HUGE || PS <= LAD_BLOCK, "one matrix row per thread in the pivot search");
        for (int col = 0; col < p; ++col) {
            // partial pivoting, all rows at once: thread r loads G[r][col] (its elimination factor as well), rows >= col compete
            // for the largest magnitude (ties: the smallest row, as a serial scan would choose); three barriers per column
            double cand_v = 0.0;
            int cand_r = 0x7fffffff;
            for (int r = tid; r < p; r += LAD_BLOCK) {   // (one trip up to LAD_WIDEP columns)
                const double v = G[r * 2 * ps + col];
                q_fac[r] = v;
                if (r >= col && (cand_r == 0x7fffffff || fabs(v) > fabs(cand_v))) { cand_v = v; cand_r = r; }   // ascending r: ties keep the smaller row
            }
            wave_reduce_pair(cand_v, cand_r, [](double v2, int r2, double v1, int r1) { return r2 != 0x7fffffff && (r1 == 0x7fffffff || fabs(v2) > fabs(v1) || (fabs(v2) == fabs(v1) && r2 < r1)); });
            if ((tid & 63) == 0) { sh.red_t[tid >> 6] = cand_v; sh.red_k[tid >> 6] = cand_r; }
            __syncthreads();
            double pv = sh.red_t[0]; int piv = sh.red_k[0];
#pragma unroll
            for (int w = 1; w < LAD_BLOCK / 64; ++w) {
                const double v2 = sh.red_t[w]; const int r2 = sh.red_k[w];
                if (r2 != 0x7fffffff && (piv == 0x7fffffff || fabs(v2) > fabs(pv) || (fabs(v2) == fabs(pv) && r2 < piv))) { pv = v2; piv = r2; }
            }
            if (!(fabs(pv) >= 1e-12)) { if (tid == 0) { sh.status = 3; sh.done = 1; } __syncthreads(); break; }   // (block-uniform)
            const double dinv = 1.0 / pv;
            // row `piv` scaled becomes row `col`; the old row `col` moves to `piv`
            for (int c2 = tid; c2 < 2 * p; c2 += LAD_BLOCK) {
                const double a_ = G[col * 2 * ps + c2], b_ = G[piv * 2 * ps + c2];
                G[col * 2 * ps + c2] = b_ * dinv;
                if (piv != col) G[piv * 2 * ps + c2] = a_;
            }
            const double f_colrow = q_fac[col];   // the factor of the row that now sits at `piv`
            __syncthreads();
            for (int i = tid; i < p * 2 * p; i += LAD_BLOCK) {
                const int r = i / (2 * p), cc = i % (2 * p);
                if (r != col) G[r * 2 * ps + cc] -= (r == piv ? f_colrow : q_fac[r]) * G[col * 2 * ps + cc];
            }
            __syncthreads();
        }
        if (sh.done) break;
        for (int i = tid; i < p * p; i += LAD_BLOCK) W[(i / p) * ps + (i % p)] = G[(i / p) * 2 * ps + p + (i % p)];
        __syncthreads();
        }   // refactor
        LAD_TICK(7);
    }
    // ---- final vertex with the UNPERTURBED right-hand sides, clipped to the box
    __syncthreads();
    for (int j = tid; j < p; j += LAD_BLOCK) {
        int ty = q_type[j];
        q_c[j] = ty == C_UB ? q_ub[q_jk[j]] : ty == C_PAT ? ra.a[q_i0[j]] : 0.0;
    }
    __syncthreads();
    for (int j = tid; j < p; j += LAD_BLOCK) {
        double v = 0.0;
        for (int i = 0; i < p; ++i) v += W[j * ps + i] * q_c[i];
        if (v < 0.0) v = 0.0;
        if (v > q_ub[j]) v = q_ub[j];
        A.x_out[c0 + j] = v;
    }
    if (tid == 0) {
        A.status[s] = (it >= max_it) ? 1 : sh.status;
        A.iters[s] = it;
    }
}

// NW == 1: one workgroup per species of the db, species with more than LAD_MAXP columns are left to the wide launches;
// NW == LAD_WIDE_NW / 0: one workgroup per entry of wide_list; the instance takes the species with more than LAD_MAXP columns
// whose mask has LAD_WIDE_NW words / more words ("huge": more than LAD_WIDEP haplotypes).
template <int NW>
__device__ __forceinline__ bool lad_instance_takes(const LadArgs &A, int s, int p) {
    if (NW == 1) return p <= LAD_MAXP;
    if (p <= LAD_MAXP) return false;
    return (NW == 0) == (A.wide_nw[s] > (uint32_t)LAD_WIDE_NW);
}
template <int PS, int NW>
__global__ void __launch_bounds__(LAD_BLOCK) lad_solve_kernel(LadArgs A) {
    __shared__ LadLds<PS, NW> m;
    const int s = NW == 1 ? (int)blockIdx.x : (int)A.wide_list[blockIdx.x];
    const int p = A.sp_p[s];
    if (!lad_instance_takes<NW>(A, s, p)) return;
    if (A.need && !A.need[s]) return;
    if (p <= 0) { if (threadIdx.x == 0) { A.status[s] = 0; A.iters[s] = 0; } return; }
    const uint32_t k0 = A.sp_pat_off[s], k1 = A.sp_pat_off[s + 1];
    if constexpr (NW == 1) {
        if (k1 - k0 <= (uint32_t)LAD_KLDS) lad_solve_body<PS, true, 1>(A, m, s, p, k0, k1);
        else lad_solve_body<PS, false, 1>(A, m, s, p, k0, k1);
    } else lad_solve_body<PS, false, NW>(A, m, s, p, k0, k1);
}

// Both LP solves of the strain step in ONE launch: solve, take the second-filter decision of this species
// (one thread), and solve again with the dropped columns pinned to zero -- only where a column was dropped;
// elsewhere LP2 == LP1 (m.reset() + no new constraint, profile.rs:1482-1490).
template <int PS, int NW>
__global__ void __launch_bounds__(LAD_BLOCK) lad_pair_kernel(LadArgs A1, LadArgs A2, SecondFilterArgs F) {
    __shared__ LadLds<PS, NW> m;
    const int s = NW == 1 ? (int)blockIdx.x : (int)A1.wide_list[blockIdx.x];
    const int p = A1.sp_p[s];
    if (!lad_instance_takes<NW>(A1, s, p)) return;
    const uint32_t k0 = A1.sp_pat_off[s], k1 = A1.sp_pat_off[s + 1];
    const bool lds_state = NW == 1 && k1 - k0 <= (uint32_t)LAD_KLDS;
    if (p > 0) {
        if constexpr (NW == 1) {
            if (lds_state) lad_solve_body<PS, true, 1>(A1, m, s, p, k0, k1);
            else lad_solve_body<PS, false, 1>(A1, m, s, p, k0, k1);
        } else lad_solve_body<PS, false, NW>(A1, m, s, p, k0, k1);
    } else if (threadIdx.x == 0) { A1.status[s] = 0; A1.iters[s] = 0; }
    __syncthreads();   // x1 / status1 of this species are visible to the workgroup
    if (threadIdx.x == 0) second_filter_species(F, (uint32_t)s);
    __syncthreads();
    if (p <= 0 || !F.need2[s]) return;
    if constexpr (NW == 1) {
        if (lds_state) lad_solve_body<PS, true, 1>(A2, m, s, p, k0, k1);
        else lad_solve_body<PS, false, 1>(A2, m, s, p, k0, k1);
    } else lad_solve_body<PS, false, NW>(A2, m, s, p, k0, k1);
}

// objective (1/n) sum_{a_v>0} |m_v . x - a_v| over the nodes of each solved species (profile.rs:1440-1450),
// for the first solution and -- where need2 says there was a second solve -- the second one, in one pass over
// the nodes.  The workgroup that finishes a species last adds the chunk partials in fixed order.
__global__ void __launch_bounds__(256) objective_kernel(const int32_t *__restrict__ sp_p, const uint8_t *__restrict__ need2,
                                                        const uint32_t *__restrict__ node_base, const double *__restrict__ ab,
                                                        const unsigned long long *__restrict__ mask, const uint64_t *__restrict__ col_off,
                                                        const uint32_t *__restrict__ wide_off, const uint32_t *__restrict__ wide_nw,
                                                        const unsigned long long *__restrict__ maskw, const double *__restrict__ x1,
                                                        const double *__restrict__ x2, double *part /*[S][STAT_CHUNKS][2]*/,
                                                        uint32_t *__restrict__ done /*[S], zero between launches*/,
                                                        const uint32_t *__restrict__ nvalid, double *__restrict__ obj1, double *__restrict__ obj2, uint32_t nch) {
    __shared__ double red[4];
    __shared__ double xs1[LAD_WIDEP], xs2[LAD_WIDEP];
    __shared__ int s_last;
    const int s = blockIdx.x / nch;
    const int p = sp_p[s];
    if (p <= 0) return;
    const bool two = x2 && need2 && need2[s];
    const uint32_t ch = blockIdx.x % nch;
    static_assert(LAD_WIDEP <= 256, "one column per thread");
    if ((int)threadIdx.x < p && p <= LAD_WIDEP) { xs1[threadIdx.x] = x1[col_off[s] + threadIdx.x]; xs2[threadIdx.x] = two ? x2[col_off[s] + threadIdx.x] : 0.0; }
    __syncthreads();
    const uint32_t b = node_base[s], e = node_base[s + 1];
    const uint32_t per = (e - b + nch - 1) / nch;
    uint32_t lo = b + ch * per, hi = lo + per;
    if (hi > e) hi = e;
    double acc1 = 0.0, acc2 = 0.0;
    if (p > LAD_MAXP && wide_nw[s] > (uint32_t)LAD_WIDE_NW) {   // huge species: any number of mask words, x read where the solver left it
        const unsigned long long *mw = maskw + (size_t)wide_off[s] * LAD_WIDE_NW;
        const int nw = (int)wide_nw[s];
        const double *X1 = x1 + col_off[s], *X2 = two ? x2 + col_off[s] : nullptr;
        for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
            const double a = ab[v];
            if (a > 0.0) {
                const uint64_t *mn = (const uint64_t *)(mw + (size_t)(v - b) * nw);
                acc1 += fabs(mdotx<0>(mn, nw, X1) - a);
                if (two) acc2 += fabs(mdotx<0>(mn, nw, X2) - a);
            }
        }
    } else if (p > LAD_MAXP) {   // wide species: the mask words of the node
        const unsigned long long *mw = maskw + (size_t)wide_off[s] * LAD_WIDE_NW;
        for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
            const double a = ab[v];
            if (a > 0.0) {
                const uint64_t *m4 = (const uint64_t *)(mw + (size_t)(v - b) * LAD_WIDE_NW);
                acc1 += fabs(mdotw<LAD_WIDE_NW>(m4, xs1) - a);
                if (two) acc2 += fabs(mdotw<LAD_WIDE_NW>(m4, xs2) - a);
            }
        }
    } else
    for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
        const double a = ab[v];
        if (a > 0.0) {
            const unsigned long long mk = mask[v];
            acc1 += fabs(mdot(mk, xs1) - a);
            if (two) acc2 += fabs(mdot(mk, xs2) - a);
        }
    }
    acc1 = block_sum_f64<256>(acc1, red);
    acc2 = block_sum_f64<256>(acc2, red);
    if (threadIdx.x == 0) {
        part[((size_t)s * nch + ch) * 2] = acc1;
        part[((size_t)s * nch + ch) * 2 + 1] = acc2;
        // release: the partials are visible device-wide before the count; acquire: the last arriver sees all of them
        s_last = __hip_atomic_fetch_add(&done[s], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nch - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup adds the partials: thread c takes chunk c (nch <= block size), fixed-shape block sum
    static_assert(STAT_CHUNKS <= 256, "one partial per thread");
    double t1 = 0.0, t2 = 0.0;
    if (threadIdx.x < nch) { t1 = part[((size_t)s * nch + threadIdx.x) * 2]; t2 = part[((size_t)s * nch + threadIdx.x) * 2 + 1]; }
    t1 = block_sum_f64<256>(t1, red);
    t2 = block_sum_f64<256>(t2, red);
    if (threadIdx.x != 0) return;
    obj1[s] = nvalid[s] ? t1 / (double)nvalid[s] : 0.0;
    if (two) obj2[s] = nvalid[s] ? t2 / (double)nvalid[s] : 0.0;
    done[s] = 0;
}

// The same objective from the SORTED ROWS (the many-species step, species of at most 64 columns): every row of a pattern has the pattern's
// prediction, so the pass reads 8 bytes per row -- 1.6 GB at cfg4 where the pass over the nodes reads abundance and mask of every node,
// 5.1 GB; the nodes with a > 0 and an empty mask, which are no rows, contribute the constant c0[s] that the row sort's histogram pass
// summed on its way (fixed order).  The sums run in another order than objective_kernel's: equal to the last bits of a double, not bit for bit.
__global__ void __launch_bounds__(256) pattern_pred_kernel(const int32_t *__restrict__ sp_p, const uint8_t *__restrict__ need2, const uint32_t *__restrict__ sp_pat_off,
                                                           const uint64_t *__restrict__ pat_mask, const uint64_t *__restrict__ col_off, const double *__restrict__ x1,
                                                           const double *__restrict__ x2, double *__restrict__ pred1, double *__restrict__ pred2) {
    __shared__ double xs1[LAD_MAXP], xs2[LAD_MAXP];
    const int s = blockIdx.x, p = sp_p[s];
    if (p <= 0 || p > LAD_MAXP) return;
    const bool two = x2 && need2 && need2[s];
    if ((int)threadIdx.x < p) { xs1[threadIdx.x] = x1[col_off[s] + threadIdx.x]; xs2[threadIdx.x] = two ? x2[col_off[s] + threadIdx.x] : 0.0; }
    __syncthreads();
    for (uint32_t k = sp_pat_off[s] + threadIdx.x; k < sp_pat_off[s + 1]; k += 256) {
        const uint64_t mk = pat_mask[k];
        pred1[k] = mdot(mk, xs1);
        if (two) pred2[k] = mdot(mk, xs2);
    }
}
__global__ void __launch_bounds__(256) objective_rows_kernel(const int32_t *__restrict__ sp_p, const uint8_t *__restrict__ need2, const uint32_t *__restrict__ sp_pat_off,
                                                             const uint32_t *__restrict__ pat_start, const double *__restrict__ row_a, const double *__restrict__ pred1,
                                                             const double *__restrict__ pred2, const double *__restrict__ c0, double *part /*[S][STAT_CHUNKS][2]*/,
                                                             uint32_t *__restrict__ done /*[S], zero between launches*/, const uint32_t *__restrict__ nvalid,
                                                             double *__restrict__ obj1, double *__restrict__ obj2, uint32_t nch, bool have2) {
    __shared__ double red[4];
    __shared__ int s_last;
    const int s = blockIdx.x / nch;
    const int p = sp_p[s];
    if (p <= 0) return;
    const bool two = have2 && need2 && need2[s];
    const uint32_t ch = blockIdx.x % nch;
    const uint32_t k0 = sp_pat_off[s], k1 = sp_pat_off[s + 1];
    const uint32_t r0 = pat_start[k0], r1 = pat_start[k1];              // the species' rows (pat_start[K] = all rows)
    const uint32_t per = (r1 - r0 + nch - 1) / nch;
    uint32_t lo = r0 + ch * per, hi = lo + per;
    if (lo > r1) lo = r1;
    if (hi > r1) hi = r1;
    double acc1 = 0.0, acc2 = 0.0;
    constexpr uint32_t KL = 256;                                          // patterns whose starts and predictions ride in LDS (a species has a handful)
    __shared__ uint32_t s_ps[KL + 1];
    __shared__ double s_p1[KL], s_p2[KL];
    const uint32_t K = k1 - k0;
    if (K <= KL) {                                                        // (block-uniform)
        for (uint32_t q = threadIdx.x; q <= K; q += 256) s_ps[q] = pat_start[k0 + q];
        for (uint32_t q = threadIdx.x; q < K; q += 256) { s_p1[q] = pred1[k0 + q]; s_p2[q] = two ? pred2[k0 + q] : 0.0; }
        __syncthreads();
        uint32_t a = 0;                                                   // last pattern that starts at or before row i: searched for the thread's first row,
        {                                                                 // walked on from there (its rows ascend: round 6 -- a search per row was eight dependent LDS reads)
            const uint32_t i0 = lo + threadIdx.x;
            uint32_t b = K;
            while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (s_ps[m] <= i0) a = m; else b = m; }
        }
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
            while (a + 1 < K && s_ps[a + 1] <= i) ++a;
            const double av = row_a[i];
            acc1 += fabs(s_p1[a] - av);
            if (two) acc2 += fabs(s_p2[a] - av);
        }
    } else
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        uint32_t a = k0, b = k1;
        while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (pat_start[m] <= i) a = m; else b = m; }
        const double av = row_a[i];
        acc1 += fabs(pred1[a] - av);
        if (two) acc2 += fabs(pred2[a] - av);
    }
    acc1 = block_sum_f64<256>(acc1, red);
    acc2 = block_sum_f64<256>(acc2, red);
    if (threadIdx.x == 0) {
        part[((size_t)s * nch + ch) * 2] = acc1;
        part[((size_t)s * nch + ch) * 2 + 1] = acc2;
        s_last = __hip_atomic_fetch_add(&done[s], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nch - 1;
    }
    __syncthreads();
    if (!s_last) return;
    double t1 = 0.0, t2 = 0.0;
    if (threadIdx.x < nch) { t1 = part[((size_t)s * nch + threadIdx.x) * 2]; t2 = part[((size_t)s * nch + threadIdx.x) * 2 + 1]; }
    t1 = block_sum_f64<256>(t1, red);
    t2 = block_sum_f64<256>(t2, red);
    if (threadIdx.x != 0) return;
    obj1[s] = nvalid[s] ? (t1 + c0[s]) / (double)nvalid[s] : 0.0;
    if (two) obj2[s] = nvalid[s] ? (t2 + c0[s]) / (double)nvalid[s] : 0.0;
    done[s] = 0;
}

static int objective_launch(Ctx *ctx, const Db *db, LadBatch *lb, const uint8_t *d_need2, const double *d_x1, const double *d_x2, double *d_obj1,
                            double *d_obj2) {
    const uint32_t S = db->S;
    const bool by_nodes = ctx->cfg.objective == "nodes";   // measurements / tests: the pass over the nodes
    if (lb->rows_c0_valid && lb->n_wide == 0 && (!by_nodes || lb->masks_in_sort)) {
        KTimer t(ctx, "objective_rows_kernel");
        PTX_HIP(ctx, lb->d_partial.alloc((size_t)S * STAT_CHUNKS * 4));
        if (lb->d_obj_done.n < S) {
            PTX_HIP(ctx, lb->d_obj_done.alloc(S));
            PTX_HIP(ctx, hipMemsetAsync(lb->d_obj_done.p, 0, lb->d_obj_done.bytes(), ctx->stream));   // the kernel leaves it zero
        }
        const uint32_t nch = stat_chunks(S);
        // (the solver's per-pattern scratch is free again: the predictions of both solutions go there)
        hipLaunchKernelGGL(pattern_pred_kernel, dim3(S), dim3(256), 0, ctx->stream, lb->d_p.p, d_need2, lb->d_sp_pat_off.p, lb->d_pat_mask.p, db->d_hap_off.p, d_x1, d_x2,
                           lb->d_sc_s.p, lb->d_sc_rho.p);
        hipLaunchKernelGGL(objective_rows_kernel, dim3(S * nch), dim3(256), 0, ctx->stream, lb->d_p.p, d_need2, lb->d_sp_pat_off.p, lb->d_pat_start.p, lb->row_a,
                           (const double *)lb->d_sc_s.p, (const double *)lb->d_sc_rho.p, (const double *)lb->d_c0.p, lb->d_partial.p, lb->d_obj_done.p, lb->d_nvalid.p,
                           d_obj1, d_obj2, nch, d_x2 != nullptr);
        PTX_HIP(ctx, hipGetLastError());
        return 0;
    }
    KTimer t(ctx, "objective_kernel");
    PTX_HIP(ctx, lb->d_partial.alloc((size_t)S * STAT_CHUNKS * 4));
    if (lb->d_obj_done.n < S) {
        PTX_HIP(ctx, lb->d_obj_done.alloc(S));
        PTX_HIP(ctx, hipMemsetAsync(lb->d_obj_done.p, 0, lb->d_obj_done.bytes(), ctx->stream));   // the kernel leaves it zero
    }
    const uint32_t nch = stat_chunks(S);
    hipLaunchKernelGGL(objective_kernel, dim3(S * nch), dim3(256), 0, ctx->stream, lb->d_p.p, d_need2, db->d_node_base.p, lb->d_ab.p,
                       (unsigned long long *)lb->d_mask.p, db->d_hap_off.p, lb->d_wide_off.p, lb->d_wide_nw.p, (const unsigned long long *)lb->d_maskw.p, d_x1, d_x2, lb->d_partial.p, lb->d_obj_done.p, lb->d_nvalid.p, d_obj1, d_obj2, nch);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

static LadArgs lad_args(const Db *db, LadBatch *lb, const uint8_t *d_need, const uint8_t *d_fixed, double *d_x, int32_t *d_status, int32_t *d_iters) {
    LadArgs A;
    A.prof = nullptr;
    A.row_a = lb->row_a; A.pat_mask = lb->d_pat_mask.p; A.pat_start = lb->d_pat_start.p; A.sp_pat_off = lb->d_sp_pat_off.p;
    A.pat_eps = lb->d_pat_eps.p; A.sc_s = lb->d_sc_s.p; A.sc_rho = lb->d_sc_rho.p;
    A.sc_lo = lb->d_sc_lo.p; A.sc_up = lb->d_sc_up.p; A.ls_lo = lb->d_ls_lo.p; A.ls_hi = lb->d_ls_hi.p; A.ls_mid = lb->d_ls_mid.p;
    A.sp_p = lb->d_p.p; A.need = d_need; A.fixed = d_fixed; A.amax = lb->d_amax.p; A.x_out = d_x; A.status = d_status; A.iters = d_iters;
    A.col_off = db->d_hap_off.p;
    A.wide_list = lb->d_wide_list.p; A.wide_off = lb->d_wide_off.p; A.wide_slot = lb->d_wide_slot.p;
    A.pat_or = lb->d_pat_or.p; A.pat_and = lb->d_pat_and.p; A.wide_W = lb->d_wide_W.p; A.wide_G = lb->d_wide_G.p;
    A.wide_nw = lb->d_wide_nw.p; A.wide_woff = lb->d_wide_woff.p; A.wide_coff = lb->d_wide_woff.p + lb->n_wide;
    A.huge_f64 = lb->d_huge_f64.p; A.huge_i32 = lb->d_huge_i32.p; A.pat_act = lb->d_pat_act.p;
    return A;
}

#ifdef LAD_PROFILE
// phase table of the solver workgroups (profiling builds): zeroed before the launch, printed after it
static const char *const LAD_PHASE_NAMES[8] = {"vertex+pattern pass", "multipliers+choice", "line-search setup", "bracket", "wide rounds",
                                               "binary rounds", "walk", "basis update"};
static int lad_prof_begin(Ctx *ctx, uint32_t S, DevBuf<unsigned long long> &buf, LadArgs &A) {
    PTX_HIP(ctx, buf.alloc((size_t)S * 16));
    PTX_HIP(ctx, hipMemsetAsync(buf.p, 0, buf.bytes(), ctx->stream));
    A.prof = buf.p;
    return 0;
}
static int lad_prof_end(Ctx *ctx, uint32_t S, DevBuf<unsigned long long> &buf, const int32_t *d_iters) {
    std::vector<unsigned long long> h((size_t)S * 16);
    std::vector<int32_t> it(S);
    PTX_TRY(download(ctx, h.data(), buf.p, h.size()));
    PTX_TRY(download(ctx, it.data(), d_iters, S));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t s = 0; s < S && s < 4; ++s) {
        unsigned long long tot = 0;
        for (int ph = 0; ph < 8; ++ph) tot += h[(size_t)s * 16 + ph];
        if (!tot) continue;
        std::fprintf(stderr, "[lad profile] species %u: %d pivots, %.1f us in the pivot loop (%.1f us per pivot)\n", s, it[s], tot * 0.01, it[s] ? tot * 0.01 / it[s] : 0.0);
        for (int ph = 0; ph < 8; ++ph)
            std::fprintf(stderr, "    %-22s %9.1f us  %5.1f %%  (%llu visits, %.2f us each)\n", LAD_PHASE_NAMES[ph], h[(size_t)s * 16 + ph] * 0.01,
                         100.0 * h[(size_t)s * 16 + ph] / tot, h[(size_t)s * 16 + 8 + ph], h[(size_t)s * 16 + 8 + ph] ? h[(size_t)s * 16 + ph] * 0.01 / h[(size_t)s * 16 + 8 + ph] : 0.0);
    }
    return 0;
}
#endif

// the strain step's two solves + second filter + both objectives: two launches
int lad_pair_launch(Ctx *ctx, const Db *db, LadBatch *lb, int pmax_bound, const FilterCfg &fc) {
    const uint32_t S = db->S;
    LadArgs A1 = lad_args(db, lb, nullptr, nullptr, lb->d_x.p, lb->d_status.p, lb->d_iters.p);
    LadArgs A2 = lad_args(db, lb, nullptr, lb->d_fixed2.p, lb->d_x2.p, lb->d_status2.p, lb->d_iters2.p);
    SecondFilterArgs F = second_filter_args(db, lb, fc, lb->d_x.p, lb->d_status.p, lb->d_fixed2.p, lb->d_need2.p);
#ifdef LAD_PROFILE
    DevBuf<unsigned long long> prof;
    PTX_TRY(lad_prof_begin(ctx, S, prof, A1));   // the first solve only
#endif
    {
        KTimer t(ctx, "lad_pair_kernel");
        if (pmax_bound <= 16) hipLaunchKernelGGL((lad_pair_kernel<16, 1>), dim3(S), dim3(LAD_BLOCK), 0, ctx->stream, A1, A2, F);
        else hipLaunchKernelGGL((lad_pair_kernel<LAD_MAXP, 1>), dim3(S), dim3(LAD_BLOCK), 0, ctx->stream, A1, A2, F);
        if (lb->n_wide > lb->n_huge) hipLaunchKernelGGL((lad_pair_kernel<LAD_WIDEP, LAD_WIDE_NW>), dim3(lb->n_wide), dim3(LAD_BLOCK), 0, ctx->stream, A1, A2, F);
        if (lb->n_huge) hipLaunchKernelGGL((lad_pair_kernel<1, 0>), dim3(lb->n_wide), dim3(LAD_BLOCK), 0, ctx->stream, A1, A2, F);
    }
    PTX_HIP(ctx, hipGetLastError());
#ifdef LAD_PROFILE
    PTX_TRY(lad_prof_end(ctx, S, prof, lb->d_iters.p));
#endif
    return objective_launch(ctx, db, lb, lb->d_need2.p, lb->d_x.p, lb->d_x2.p, lb->d_obj.p, lb->d_obj2.p);
}

int lad_solve_launch(Ctx *ctx, const Db *db, LadBatch *lb, int pmax_bound, const uint8_t *d_need, const uint8_t *d_fixed, double *d_x,
                     double *d_obj, int32_t *d_status, int32_t *d_iters) {
    const uint32_t S = db->S;
    LadArgs A = lad_args(db, lb, d_need, d_fixed, d_x, d_status, d_iters);
#ifdef LAD_PROFILE
    DevBuf<unsigned long long> prof;
    PTX_TRY(lad_prof_begin(ctx, S, prof, A));
#endif
    {
        KTimer t(ctx, "lad_solve_kernel");
        if (pmax_bound <= 16) hipLaunchKernelGGL((lad_solve_kernel<16, 1>), dim3(S), dim3(LAD_BLOCK), 0, ctx->stream, A);
        else hipLaunchKernelGGL((lad_solve_kernel<LAD_MAXP, 1>), dim3(S), dim3(LAD_BLOCK), 0, ctx->stream, A);
        if (lb->n_wide > lb->n_huge) hipLaunchKernelGGL((lad_solve_kernel<LAD_WIDEP, LAD_WIDE_NW>), dim3(lb->n_wide), dim3(LAD_BLOCK), 0, ctx->stream, A);
        if (lb->n_huge) hipLaunchKernelGGL((lad_solve_kernel<1, 0>), dim3(lb->n_wide), dim3(LAD_BLOCK), 0, ctx->stream, A);
    }
    PTX_HIP(ctx, hipGetLastError());
#ifdef LAD_PROFILE
    PTX_TRY(lad_prof_end(ctx, S, prof, d_iters));
#endif
    return objective_launch(ctx, db, lb, nullptr, d_x, nullptr, d_obj, nullptr);
}

}  // namespace ptx
