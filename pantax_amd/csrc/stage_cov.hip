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
// HBM-bound scatter kernel.  Algorithmic bytes per launch:
//   4T + 12R + 4R(species) + 8V(bit_off) + 8V(bases) + L/8 (bitmap) + 4V(cov) + 12*(T-2R) (trio probes)
// Layout: node arrays of all resident species are concatenated; a node's coverage bitmap starts
// at bit bit_off[v] of one global bit vector (1 bit per graph base instead of the reference's
// 1 byte, profile.rs:776-781).
#include <cstdlib>
#include "common.hpp"

namespace ptx {

constexpr int COV_BLOCK = 256;

__device__ __forceinline__ void bitmap_or_range(uint32_t *__restrict__ bm, uint64_t g0, uint64_t g1) {
    if (g1 <= g0) return;
    uint64_t w0 = g0 >> 5, w1 = (g1 - 1) >> 5;
    uint32_t m0 = 0xFFFFFFFFu << (g0 & 31);
    uint32_t m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
    if (w0 == w1) {
        uint32_t m = m0 & m1;
        if ((bm[w0] & m) != m) atomicOr(&bm[w0], m);  // a stale 0 only costs a redundant OR; bits never clear
    } else {
        if ((bm[w0] & m0) != m0) atomicOr(&bm[w0], m0);
        for (uint64_t w = w0 + 1; w < w1; ++w)
            if (bm[w] != 0xFFFFFFFFu) atomicOr(&bm[w], 0xFFFFFFFFu);
        if ((bm[w1] & m1) != m1) atomicOr(&bm[w1], m1);
    }
}

__device__ __forceinline__ int trio_find(const uint32_t *__restrict__ trio_first, const uint2 *__restrict__ trio_bc,
                                         uint32_t gnode_a, uint32_t b, uint32_t c) {
    uint32_t lo = trio_first[gnode_a], hi = trio_first[gnode_a + 1];
    for (uint32_t j = lo; j < hi; ++j) {  // rows per first node are very short (usually 0-3)
        uint2 k = trio_bc[j];
        if (k.x == b && k.y == c) return (int)j;
    }
    return -1;
}

template <bool WITH_TRIO>
__global__ void __launch_bounds__(COV_BLOCK) coverage_kernel(
    uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
    const uint32_t *__restrict__ pstart, const uint32_t *__restrict__ pend, const int32_t *__restrict__ species,
    const uint8_t *__restrict__ flags, const uint8_t *__restrict__ active, const uint32_t *__restrict__ sp_first_id,
    const uint32_t *__restrict__ node_base, const uint64_t *__restrict__ bit_off,
    unsigned long long *__restrict__ bases, uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ trio_first,
    const uint2 *__restrict__ trio_bc, const uint32_t *__restrict__ trio_row,
    unsigned long long *__restrict__ trio_bases, unsigned long long *__restrict__ n_abort, int ablate) {
    // `ablate` is 0 in production; PANTAX_HIP_COV_ABLATE (debug) switches sub-stages off for profiling
    for (uint64_t r = (uint64_t)blockIdx.x * COV_BLOCK + threadIdx.x; r < R; r += (uint64_t)gridDim.x * COV_BLOCK) {
        int sp = species[r];
        if (sp < 0) continue;                       // "U" (profile.rs:3352-3356)
        if (active && !active[sp]) continue;        // species not kept by load_species_range (:553-656)
        if (flags && flags[r]) continue;            // null field / duplicate-id drop (:380-437)
        uint32_t b = step_off[r], e = step_off[r + 1];
        uint32_t k = e - b;
        if (k == 0) continue;                       // :794-796
        uint32_t first_id = sp_first_id[sp];
        uint32_t nb = node_base[sp];
        uint32_t Vs = node_base[sp + 1] - nb;
        bool bad = false;
        for (uint32_t i = b; i < e; ++i) {
            uint32_t id = node_id[i];
            if (id < first_id || id - first_id >= Vs) { bad = true; break; }   // index panic at :849
        }
        if (bad) { atomicAdd(n_abort, 1ull); continue; }
        long long ps = pstart[r], pe = pend[r];
        long long target = pe - ps;                 // :800
        uint32_t l0 = node_id[b] - first_id;
        uint64_t bo0 = bit_off[nb + l0];
        long long len0 = (long long)(bit_off[nb + l0 + 1] - bo0);
        if (k == 1) {                               // :811
            if (target < 0) continue;               // :821-827
            if (target) atomicAdd(&bases[nb + l0], (unsigned long long)target);
            if (ps < pe && pe <= len0) bitmap_or_range(bitmap, bo0 + ps, bo0 + pe);   // :832
            continue;
        }
        if (ps > len0) { atomicAdd(n_abort, 1ull); continue; }                       // assert :854
        long long seen = 0;
        uint32_t lm2 = 0, lm1 = 0;       // local ids at i-2, i-1
        long long rl2 = 0, rl1 = 0;      // read-local aligned lengths (read_nodes_len) at i-2, i-1
        for (uint32_t i = 0; i < k; ++i) {
            uint32_t id = node_id[b + i];
            uint32_t l = id - first_id;
            uint32_t v = nb + l;
            uint64_t bo = bit_off[v];
            long long nl = (long long)(bit_off[v + 1] - bo);
            long long aln, sidx;
            if (i == 0) { aln = nl - ps; sidx = ps; }
            else if (i == k - 1) { long long t = target < seen ? seen : target; aln = t - seen; sidx = 0; }
            else { aln = nl; sidx = 0; }
            long long hi = sidx + aln;
            if (hi > nl) hi = nl;                                                     // :871
            if (!(ablate & 1)) bitmap_or_range(bitmap, bo + sidx, bo + hi);
            seen += aln;
            int jf = -1;                                                              // first occurrence? (:879)
            if (!(ablate & 8))
            for (uint32_t j = 0; j < i; ++j)
                if (node_id[b + j] == id) { jf = (int)j; break; }
            long long rl;
            if (jf < 0) {
                rl = aln;
                if (aln && !(ablate & 2)) atomicAdd(&bases[v], (unsigned long long)aln);   // :881
            } else {
                rl = (jf == 0) ? (len0 - ps) : nl;   // read_nodes_len holds the first occurrence's length
            }
            if (WITH_TRIO && i >= 2 && !(ablate & 4)) {                               // :890-907
                uint32_t a = lm2, c = l;
                if (a > c) { uint32_t t = a; a = c; c = t; }
                int j = trio_find(trio_first, trio_bc, nb + a, lm1, c);
                if (j >= 0) {
                    long long s = rl2 + rl1 + rl;
                    if (s) atomicAdd(&trio_bases[trio_row[j]], (unsigned long long)s);
                }
            }
            lm2 = lm1; lm1 = l;
            rl2 = rl1; rl1 = rl;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Thread-per-STEP form of the same histogram (default).  The per-read kernel above walks a read
// serially (a chain of ~5 dependent memory round trips per step); here every walk step is its own
// thread, so a wave holds 64 consecutive steps (of ~8 neighbouring reads), all gathers of a step
// are independent, and the only serial work left is the last step's sum over its read's interior
// node lengths.  Values of steps i-1 / i-2 come from the neighbouring lanes by wave shuffle (wave64),
// with a memory fallback for the first two lanes.  step_read[t] = read index of step t.
// A read that reaches this kernel was binned to its species, i.e. every node id lies inside the
// species' id range (rcls.rs:253-257), so the index panic of profile.rs:849 cannot occur; an
// out-of-range id (inconsistent external binning) is counted as an abort per step instead.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long rl_from_memory(uint32_t j, uint32_t b, const uint32_t *__restrict__ node_id, uint32_t first_id,
                                                    uint32_t nb, const uint32_t *__restrict__ node_len, long long len0, long long ps) {
    // read_nodes_len of position j (never the last position): length aligned at the node's FIRST occurrence
    uint32_t idj = node_id[b + j];
    int jf = -1;
    for (uint32_t q = 0; q < j; ++q) if (node_id[b + q] == idj) { jf = (int)q; break; }
    uint32_t src = jf < 0 ? j : (uint32_t)jf;
    if (src == 0) return len0 - ps;
    return (long long)node_len[nb + (idj - first_id)];
}

template <bool WITH_TRIO>
__global__ void __launch_bounds__(COV_BLOCK) coverage_step_kernel(
    uint64_t T, const uint32_t *__restrict__ step_read, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
    const uint32_t *__restrict__ pstart, const uint32_t *__restrict__ pend, const int32_t *__restrict__ species,
    const uint8_t *__restrict__ flags, const uint8_t *__restrict__ active, const uint32_t *__restrict__ sp_first_id,
    const uint32_t *__restrict__ node_base, const uint64_t *__restrict__ bit_off, const uint32_t *__restrict__ node_len,
    unsigned long long *__restrict__ bases, uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ trio_first,
    const uint2 *__restrict__ trio_bc, const uint32_t *__restrict__ trio_row,
    unsigned long long *__restrict__ trio_bases, unsigned long long *__restrict__ n_abort, int ablate) {
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * COV_BLOCK;
    for (uint64_t base = (uint64_t)blockIdx.x * COV_BLOCK + (threadIdx.x - lane); base < T; base += stride) {
        const uint64_t t = base + lane;
        bool ok = t < T;
        uint32_t r = 0, b = 0, k = 0, i = 0, id = 0, l = 0, v = 0, first_id = 0, nb = 0;
        long long ps = 0, pe = 0, nl = 0;
        uint64_t bo = 0;
        if (ok) {
            r = step_read[t];
            int sp = species[r];
            ok = sp >= 0 && !(active && !active[sp]) && !(flags && flags[r]);
            if (ok) {
                b = step_off[r]; k = step_off[r + 1] - b; i = (uint32_t)(t - b);
                id = node_id[t];
                first_id = sp_first_id[sp]; nb = node_base[sp];
                uint32_t Vs = node_base[sp + 1] - nb;
                if (id < first_id || id - first_id >= Vs) { atomicAdd(n_abort, 1ull); ok = false; }
            }
            if (ok) {
                l = id - first_id; v = nb + l;
                bo = bit_off[v];
                nl = (long long)node_len[v];
                ps = pstart[r]; pe = pend[r];
            }
        }
        // first node length: from the lane that holds step b, else from memory
        long long nl_src = __shfl(nl, (lane >= (int)i) ? lane - (int)i : lane);
        long long len0 = nl;
        if (ok && i > 0) len0 = (lane >= (int)i) ? nl_src : (long long)node_len[nb + (node_id[b] - first_id)];
        const long long target = pe - ps;                         // profile.rs:800
        long long rl = 0;
        if (ok && k == 1) {                                       // :811
            if (target >= 0) {                                    // :821-827
                if (target) atomicAdd(&bases[v], (unsigned long long)target);
                if (ps < pe && pe <= nl) bitmap_or_range(bitmap, bo + ps, bo + pe);   // :832
            }
            ok = false;
        }
        if (ok && ps > len0) {                                    // assert :854 -> whole read contributes nothing
            if (i == 0) atomicAdd(n_abort, 1ull);
            ok = false;
        }
        if (ok) {
            long long aln, sidx;
            if (i == 0) { aln = nl - ps; sidx = ps; }             // :853-856
            else if (i == k - 1) {                                // :857-859
                long long seen = len0 - ps;
                if (!(ablate & 16))
                for (uint32_t j = 1; j + 1 < k; ++j) seen += (long long)node_len[nb + (node_id[b + j] - first_id)];
                long long tt = target < seen ? seen : target;
                aln = tt - seen; sidx = 0;
            } else { aln = nl; sidx = 0; }                        // :860-862
            long long hi = sidx + aln;
            if (hi > nl) hi = nl;                                 // :871
            if (!(ablate & 1)) bitmap_or_range(bitmap, bo + sidx, bo + hi);
            int jf = -1;                                          // first occurrence of this node in the read? (:879)
            if (!(ablate & 8))
            for (uint32_t j = 0; j < i; ++j)
                if (node_id[b + j] == id) { jf = (int)j; break; }
            if (jf < 0) {
                rl = aln;
                if (aln && !(ablate & 2)) atomicAdd(&bases[v], (unsigned long long)aln);   // :881
            } else rl = (jf == 0) ? (len0 - ps) : nl;
        }
        if (WITH_TRIO) {                                          // :890-907
            // (local id, read_nodes_len) of steps t-1 and t-2 from the neighbouring lanes
            uint32_t l1 = __shfl_up(l, 1), l2 = __shfl_up(l, 2);
            long long rl1 = __shfl_up(rl, 1), rl2 = __shfl_up(rl, 2);
            if (ok && i >= 2 && !(ablate & 4)) {
                if (lane < 1) { l1 = node_id[b + i - 1] - first_id; rl1 = rl_from_memory(i - 1, b, node_id, first_id, nb, node_len, len0, ps); }
                if (lane < 2) { l2 = node_id[b + i - 2] - first_id; rl2 = rl_from_memory(i - 2, b, node_id, first_id, nb, node_len, len0, ps); }
                uint32_t a = l2, c = l;
                if (a > c) { uint32_t tmp = a; a = c; c = tmp; }
                int j = trio_find(trio_first, trio_bc, nb + a, l1, c);
                if (j >= 0) {
                    long long sum = rl2 + rl1 + rl;
                    if (sum) atomicAdd(&trio_bases[trio_row[j]], (unsigned long long)sum);
                }
            }
        }
    }
}

// step_read[t] = r for every step of read r (derived index of the packed stream; built once per upload)
__global__ void __launch_bounds__(256) step_read_kernel(uint64_t R, const uint32_t *__restrict__ step_off, uint32_t *__restrict__ step_read) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256)
        for (uint32_t q = step_off[r]; q < step_off[r + 1]; ++q) step_read[q] = (uint32_t)r;
}

int build_step_read(Ctx *ctx, Reads *rd) {
    PTX_HIP(ctx, rd->d_step_read.alloc(rd->T));
    if (rd->R == 0) return 0;
    hipLaunchKernelGGL(step_read_kernel, dim3(grid_for(rd->R, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, rd->R, rd->d_step_off.p,
                       rd->d_step_read.p);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

// node_base_cov[v] = number of covered bases (profile.rs:844/874, :1018-1023)
__global__ void __launch_bounds__(256) popcount_kernel(uint64_t V, const uint64_t *__restrict__ bit_off,
                                                       const uint32_t *__restrict__ bitmap, uint32_t *__restrict__ cov) {
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (uint64_t)gridDim.x * 256) {
        uint64_t g0 = bit_off[v], g1 = bit_off[v + 1];
        uint32_t c = 0;
        if (g1 > g0) {
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

int coverage_launch(Ctx *ctx, Db *db, Reads *rd, const uint8_t *d_active, bool with_trio, unsigned long long *d_abort) {
    uint64_t words = (db->L + 31) / 32 + 1;
    PTX_HIP(ctx, db->d_bases.alloc(db->V));
    PTX_HIP(ctx, db->d_bitmap.alloc(words));
    PTX_HIP(ctx, db->d_cov.alloc(db->V));
    PTX_HIP(ctx, hipMemsetAsync(db->d_bases.p, 0, db->V * sizeof(unsigned long long), ctx->stream));
    PTX_HIP(ctx, hipMemsetAsync(db->d_bitmap.p, 0, words * sizeof(uint32_t), ctx->stream));
    PTX_HIP(ctx, hipMemsetAsync(d_abort, 0, sizeof(unsigned long long), ctx->stream));
    if (with_trio) {
        PTX_HIP(ctx, db->d_trio_bases.alloc(db->U));
        PTX_HIP(ctx, hipMemsetAsync(db->d_trio_bases.p, 0, (db->U ? db->U : 1) * sizeof(unsigned long long), ctx->stream));
    }
    static const bool per_read = getenv("PANTAX_HIP_COV_MODE") && getenv("PANTAX_HIP_COV_MODE")[0] == 'r';   // A/B switch (debug)
    if (rd->R && !per_read) {
        static const int ablate_s = getenv("PANTAX_HIP_COV_ABLATE") ? atoi(getenv("PANTAX_HIP_COV_ABLATE")) : 0;
        int grid = grid_for(rd->T, COV_BLOCK, ctx->n_cu * 16);
        KTimer t(ctx, "coverage_step_kernel");
#define COVS_ARGS rd->T, rd->d_step_read.p, rd->d_step_off.p, rd->d_node_id.p, rd->d_pstart.p, rd->d_pend.p, rd->d_species.p,           \
                  rd->has_flags ? rd->d_flags.p : nullptr, d_active, db->d_sp_first_id.p, db->d_node_base.p, db->d_bit_off.p,        \
                  db->d_node_len.p, db->d_bases.p, db->d_bitmap.p, db->d_trio_first.p, db->d_trio_bc.p, db->d_trio_row.p, db->d_trio_bases.p, d_abort, ablate_s
        if (with_trio && db->U) hipLaunchKernelGGL((coverage_step_kernel<true>), dim3(grid), dim3(COV_BLOCK), 0, ctx->stream, COVS_ARGS);
        else hipLaunchKernelGGL((coverage_step_kernel<false>), dim3(grid), dim3(COV_BLOCK), 0, ctx->stream, COVS_ARGS);
#undef COVS_ARGS
    } else if (rd->R) {
        static const int ablate_env = getenv("PANTAX_HIP_COV_ABLATE") ? atoi(getenv("PANTAX_HIP_COV_ABLATE")) : 0;
        const int ablate = ablate_env;
        int grid = grid_for(rd->R, COV_BLOCK, ctx->n_cu * 8);
        KTimer t(ctx, "coverage_kernel");
#define COV_ARGS rd->R, rd->d_step_off.p, rd->d_node_id.p, rd->d_pstart.p, rd->d_pend.p, rd->d_species.p,                \
                 rd->has_flags ? rd->d_flags.p : nullptr, d_active, db->d_sp_first_id.p, db->d_node_base.p, db->d_bit_off.p, \
                 db->d_bases.p, db->d_bitmap.p, db->d_trio_first.p, db->d_trio_bc.p, db->d_trio_row.p, db->d_trio_bases.p, d_abort, ablate
        if (with_trio && db->U) hipLaunchKernelGGL((coverage_kernel<true>), dim3(grid), dim3(COV_BLOCK), 0, ctx->stream, COV_ARGS);
        else hipLaunchKernelGGL((coverage_kernel<false>), dim3(grid), dim3(COV_BLOCK), 0, ctx->stream, COV_ARGS);
#undef COV_ARGS
    }
    PTX_HIP(ctx, hipGetLastError());
    if (db->V) {
        KTimer t(ctx, "popcount_kernel");
        hipLaunchKernelGGL(popcount_kernel, dim3(grid_for(db->V, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, db->V,
                           db->d_bit_off.p, db->d_bitmap.p, db->d_cov.p);
    }
    PTX_HIP(ctx, hipGetLastError());
    db->cov_done = true;
    return 0;
}

}  // namespace ptx
