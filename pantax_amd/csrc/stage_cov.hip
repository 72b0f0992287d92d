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
    unsigned long long *__restrict__ trio_bases, unsigned long long *__restrict__ n_abort) {
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
            bitmap_or_range(bitmap, bo + sidx, bo + hi);
            seen += aln;
            int jf = -1;                                                              // first occurrence? (:879)
            for (uint32_t j = 0; j < i; ++j)
                if (node_id[b + j] == id) { jf = (int)j; break; }
            long long rl;
            if (jf < 0) {
                rl = aln;
                if (aln) atomicAdd(&bases[v], (unsigned long long)aln);               // :881
            } else {
                rl = (jf == 0) ? (len0 - ps) : nl;   // read_nodes_len holds the first occurrence's length
            }
            if (WITH_TRIO && i >= 2) {                                                // :890-907
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
    if (rd->R) {
        int grid = grid_for(rd->R, COV_BLOCK, ctx->n_cu * 8);
        KTimer t(ctx, "coverage_kernel");
#define COV_ARGS rd->R, rd->d_step_off.p, rd->d_node_id.p, rd->d_pstart.p, rd->d_pend.p, rd->d_species.p,                \
                 rd->has_flags ? rd->d_flags.p : nullptr, d_active, db->d_sp_first_id.p, db->d_node_base.p, db->d_bit_off.p, \
                 db->d_bases.p, db->d_bitmap.p, db->d_trio_first.p, db->d_trio_bc.p, db->d_trio_row.p, db->d_trio_bases.p, d_abort
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
