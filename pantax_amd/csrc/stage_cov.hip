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
// with a memory fallback only for the part of a read that lies in the previous wave.  The kernel
// is bound by the number of divergent (one cache line per lane) vector-memory instructions, so the
// tables it gathers from are packed into 16-byte records (one dwordx4 per lookup):
//   read_rec[r] = {first step, #steps, pstart, pend}      node_rec[v] = {bit_off (u64), len, -}
//   trio_node[v] = {first row, #rows}                      trio_ent[j] = {b, c, row, -}
// A read that reaches this kernel was binned to its species, so every node id lies inside the
// species' id range (rcls.rs:253-257) and the index panic of profile.rs:849 cannot occur; an
// out-of-range id (inconsistent external binning) is counted as an abort per step instead.
//
// Algorithmic bytes per launch (the a8 row minus its popcount pass):
//   4T + 12R + 4V(node_len) + 8V(bases) + L/8 (bitmap) + 12*(T-2R) (trio probes)
// Layout: node arrays of all resident species are concatenated; a node's coverage bitmap starts
// at bit bit_off[v] of one global bit vector (1 bit per graph base instead of the reference's
// 1 byte, profile.rs:776-781).
#include <cstdlib>
#include "common.hpp"
#include "primitives.hpp"

namespace ptx {

constexpr int COV_BLOCK = 256;

// The test-before-set must see other CUs' ORs to be worth anything: the ORs execute below the
// per-CU L1 (which is never refreshed by them), so the probe is an agent-scope load (sc1: L1
// bypass, served by L2).  A stale 0 only costs a redundant OR; bits never clear, so it is safe.
__device__ __forceinline__ uint32_t bm_peek(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bitmap_or_range(uint32_t *__restrict__ bm, uint64_t g0, uint64_t g1) {
    if (g1 <= g0) return;
    uint64_t w0 = g0 >> 5, w1 = (g1 - 1) >> 5;
    uint32_t m0 = 0xFFFFFFFFu << (g0 & 31);
    uint32_t m1 = 0xFFFFFFFFu >> (31 - (uint32_t)((g1 - 1) & 31));
    if (w0 == w1) {
        uint32_t m = m0 & m1;
        if ((bm_peek(&bm[w0]) & m) != m) atomicOr(&bm[w0], m);
    } else {
        if ((bm_peek(&bm[w0]) & m0) != m0) atomicOr(&bm[w0], m0);
        for (uint64_t w = w0 + 1; w < w1; ++w)
            if (bm_peek(&bm[w]) != 0xFFFFFFFFu) atomicOr(&bm[w], 0xFFFFFFFFu);
        if ((bm_peek(&bm[w1]) & m1) != m1) atomicOr(&bm[w1], m1);
    }
}

__device__ __forceinline__ int trio_find(const uint2 *__restrict__ trio_node, const uint4 *__restrict__ trio_ent, uint32_t gnode_a,
                                         uint32_t b, uint32_t c) {
    uint2 nd = trio_node[gnode_a];            // {first row, #rows}: usually 0-3 rows
    for (uint32_t j = 0; j < nd.y; ++j) {
        uint4 e = trio_ent[nd.x + j];
        if (e.x == b && e.y == c) return (int)e.z;
    }
    return -1;
}

// read_nodes_len of position j (never the last position) recomputed from memory: the length aligned
// at the node's FIRST occurrence in the read (profile.rs:879-882)
__device__ __forceinline__ long long rl_from_memory(uint32_t j, uint32_t b, const uint32_t *__restrict__ node_id, uint32_t first_id,
                                                    uint32_t nb, const uint4 *__restrict__ node_rec, long long len0, long long ps) {
    uint32_t idj = node_id[b + j];
    int jf = -1;
    for (uint32_t q = 0; q < j; ++q) if (node_id[b + q] == idj) { jf = (int)q; break; }
    uint32_t src = jf < 0 ? j : (uint32_t)jf;
    if (src == 0) return len0 - ps;
    return (long long)node_rec[nb + (idj - first_id)].z;
}

constexpr int COV_CHUNK = 1024;   // steps per workgroup
constexpr int COV_WIN = 1024;     // nodes in the LDS window
constexpr int COV_WIN_BACK = 128; // window starts this many nodes before the chunk's first start node

__device__ __forceinline__ void add_bases(unsigned long long *__restrict__ bases, uint32_t *s_win, uint32_t wlo, uint32_t v, long long aln) {
    const uint32_t off = v - wlo;   // unsigned wrap puts nodes below the window out of range too
    if (off < (uint32_t)COV_WIN && aln < (1ll << 18)) atomicAdd(&s_win[off], (uint32_t)aln);   // <= 8192 steps x 2^18 < 2^32
    else atomicAdd(&bases[v], (unsigned long long)aln);
}

// Steps arrive grouped by the locus of their read's first node (group_reads below), so a workgroup's
// chunk of COV_CHUNK consecutive steps lands in a narrow node window: `bases` is accumulated in an LDS
// window of COV_WIN nodes (32-bit LDS atomics) and flushed with one 64-bit global atomic per touched
// node -- the LDS-staged segmented reduction of the scatter.  Nodes outside the window (or oversized
// lengths) fall back to the global atomic; the result is identical either way.
template <bool WITH_TRIO>
__global__ void __launch_bounds__(COV_BLOCK) coverage_step_kernel(
    uint64_t T, const uint32_t *__restrict__ step_read, const uint4 *__restrict__ read_rec, const uint32_t *__restrict__ orig_read,
    const uint32_t *__restrict__ node_id, const int32_t *__restrict__ species, const uint8_t *__restrict__ flags,
    const uint8_t *__restrict__ active, const uint32_t *__restrict__ sp_first_id, const uint32_t *__restrict__ node_base,
    const uint4 *__restrict__ node_rec, unsigned long long *__restrict__ bases, uint32_t *__restrict__ bitmap,
    const uint2 *__restrict__ trio_node, const uint4 *__restrict__ trio_ent, unsigned long long *__restrict__ trio_bases,
    unsigned long long *__restrict__ n_abort) {
    __shared__ uint32_t s_win[COV_WIN];
    __shared__ uint32_t s_wlo;
    const int lane = threadIdx.x & 63;
    const uint64_t chunk_b = (uint64_t)blockIdx.x * COV_CHUNK;
    uint64_t chunk_e = chunk_b + COV_CHUNK;
    if (chunk_e > T) chunk_e = T;
    for (int i = threadIdx.x; i < COV_WIN; i += COV_BLOCK) s_win[i] = 0;
    // window base = global node index of the first node of the first LIVE read of the chunk
    if (threadIdx.x == 0) {
        uint32_t w = 0;
        for (uint64_t t0 = chunk_b; t0 < chunk_e;) {
            const uint32_t slot = step_read[t0];
            const uint4 rr = read_rec[slot];
            const uint32_t o = orig_read[slot];
            const int sp0 = species[o];
            if (sp0 >= 0 && !(active && !active[sp0]) && !(flags && flags[o])) {
                const uint32_t id0 = node_id[rr.x], f0 = sp_first_id[sp0];
                const uint32_t v0 = node_base[sp0] + (id0 >= f0 ? id0 - f0 : 0u);
                w = v0 > (uint32_t)COV_WIN_BACK ? v0 - COV_WIN_BACK : 0u;
                break;
            }
            t0 = (uint64_t)rr.x + rr.y;   // next read
        }
        s_wlo = w;
    }
    __syncthreads();
    const uint32_t wlo = s_wlo;
    for (uint64_t base = chunk_b + (threadIdx.x - lane); base < chunk_e; base += COV_BLOCK) {
        const uint64_t t = base + lane;
        bool ok = t < chunk_e;
        uint32_t b = 0, k = 0, i = 0, id = 0, l = 0, v = 0, first_id = 0, nb = 0;
        long long ps = 0, pe = 0, nl = 0;
        uint64_t bo = 0;
        if (ok) {
            const uint32_t slot = step_read[t];
            const uint32_t o = orig_read[slot];
            const int sp = species[o];
            ok = sp >= 0 && !(active && !active[sp]) && !(flags && flags[o]);   // "U" / unselected species / dropped rows
            if (ok) {
                const uint4 rr = read_rec[slot];
                b = rr.x; k = rr.y; ps = rr.z; pe = rr.w;
                i = (uint32_t)(t - b);
                id = node_id[t];
                first_id = sp_first_id[sp]; nb = node_base[sp];
                const uint32_t Vs = node_base[sp + 1] - nb;
                if (id < first_id || id - first_id >= Vs) { atomicAdd(n_abort, 1ull); ok = false; }
            }
            if (ok) {
                l = id - first_id; v = nb + l;
                const uint4 nr = node_rec[v];
                bo = ((uint64_t)nr.y << 32) | nr.x;
                nl = (long long)nr.z;
            }
        }
        const int dist = ok ? (int)min(i, (uint32_t)lane) : 0;   // earlier steps of my read held by lower lanes
        // first node length: from the lane that holds step b, else from memory
        const long long nl_src = __shfl(nl, lane - dist);
        long long len0 = nl;
        if (ok && i > 0) len0 = ((int)i <= lane) ? nl_src : (long long)node_rec[nb + (node_id[b] - first_id)].z;
        const long long target = pe - ps;                         // profile.rs:800
        if (ok && k == 1) {                                       // :811
            if (target >= 0) {                                    // :821-827
                if (target) add_bases(bases, s_win, wlo, v, target);
                if (ps < pe && pe <= nl) bitmap_or_range(bitmap, bo + ps, bo + pe);   // :832
            }
            ok = false;
        }
        if (ok && ps > len0) {                                    // assert :854 -> whole read contributes nothing
            if (i == 0) atomicAdd(n_abort, 1ull);
            ok = false;
        }
        // ---- `seen` before this step = sum of the aligned lengths of steps 0..i-1: segmented wave scan
        const long long contrib = ok ? (i == 0 ? nl - ps : nl) : 0;
        long long incl = contrib;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long up = __shfl_up(incl, d);
            if (dist >= d) incl += up;
        }
        // ---- first occurrence of this node in the read (:879): compare with the earlier lanes
        int dmax = 0;
        for (int d = 1; __any(dist >= d); ++d) {
            uint32_t other = __shfl_up(id, d);
            if (dist >= d && other == id) dmax = d;
        }
        long long rl = 0;
        if (ok) {
            int jf = dmax ? (int)i - dmax : -1;
            if ((int)i > lane) {                                  // the read began in the previous wave: finish from memory
                const uint32_t nprev = i - (uint32_t)lane;
                for (uint32_t j = 0; j < nprev; ++j) if (node_id[b + j] == id) { jf = (int)j; break; }
            }
            long long aln, sidx;
            if (i == 0) { aln = nl - ps; sidx = ps; }             // :853-856
            else if (i == k - 1) {                                // :857-859
                long long seen = incl - contrib;
                if ((int)i > lane) {
                    const uint32_t nprev = i - (uint32_t)lane;
                    seen += len0 - ps;
                    for (uint32_t j = 1; j < nprev; ++j) seen += (long long)node_rec[nb + (node_id[b + j] - first_id)].z;
                }
                const long long tt = target < seen ? seen : target;
                aln = tt - seen; sidx = 0;
            } else { aln = nl; sidx = 0; }                        // :860-862
            long long hi = sidx + aln;
            if (hi > nl) hi = nl;                                 // :871
            bitmap_or_range(bitmap, bo + sidx, bo + hi);
            if (jf < 0) {
                rl = aln;
                if (aln) add_bases(bases, s_win, wlo, v, aln);            // :881
            } else rl = (jf == 0) ? (len0 - ps) : nl;
        }
        if (WITH_TRIO) {                                          // :890-907
            uint32_t l1 = __shfl_up(l, 1), l2 = __shfl_up(l, 2);
            long long rl1 = __shfl_up(rl, 1), rl2 = __shfl_up(rl, 2);
            if (ok && i >= 2) {
                if (lane < 1) { l1 = node_id[b + i - 1] - first_id; rl1 = rl_from_memory(i - 1, b, node_id, first_id, nb, node_rec, len0, ps); }
                if (lane < 2) { l2 = node_id[b + i - 2] - first_id; rl2 = rl_from_memory(i - 2, b, node_id, first_id, nb, node_rec, len0, ps); }
                uint32_t a = l2, c = l;
                if (a > c) { uint32_t tmp = a; a = c; c = tmp; }
                const int row = trio_find(trio_node, trio_ent, nb + a, l1, c);
                if (row >= 0) {
                    const long long sum = rl2 + rl1 + rl;
                    if (sum) atomicAdd(&trio_bases[row], (unsigned long long)sum);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COV_WIN; i += COV_BLOCK) {
        const uint32_t c = s_win[i];
        if (c) atomicAdd(&bases[wlo + i], (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------------------------
// Resident layout of the packed reads: grouped by the locus of their first node.  Key = first node
// id >> shift (ids are globally ordered by species and position, sort_range.rs:25-33), counting sort
// (histogram -> exclusive scans over reads and steps -> scatter) into {read_rec, orig_read, node_id,
// step_read}.  Done once per upload: it depends on the reads only, not on the binning.  Slot order
// inside a bucket is arbitrary; every output of the path is an order-independent integer sum, so
// results stay bit-exact.  Reads with an empty walk own no step and are left out (profile.rs:794-796).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) group_count_kernel(uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
                                                          int shift, uint32_t *__restrict__ cnt_r, uint32_t *__restrict__ cnt_s) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256) {
        const uint32_t b = step_off[r], k = step_off[r + 1] - b;
        if (!k) continue;
        const uint32_t key = node_id[b] >> shift;
        atomicAdd(&cnt_r[key], 1u);
        atomicAdd(&cnt_s[key], k);
    }
}
__global__ void __launch_bounds__(256) group_scatter_kernel(uint64_t R, const uint32_t *__restrict__ step_off, const uint32_t *__restrict__ node_id,
                                                            const uint32_t *__restrict__ pstart, const uint32_t *__restrict__ pend, int shift,
                                                            const uint32_t *__restrict__ base_r, const uint32_t *__restrict__ base_s,
                                                            uint32_t *__restrict__ cur_r, uint32_t *__restrict__ cur_s,
                                                            uint4 *__restrict__ g_read_rec, uint32_t *__restrict__ g_orig,
                                                            uint32_t *__restrict__ g_node_id, uint32_t *__restrict__ g_step_read) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (uint64_t)gridDim.x * 256) {
        const uint32_t b = step_off[r], k = step_off[r + 1] - b;
        if (!k) continue;
        const uint32_t key = node_id[b] >> shift;
        const uint32_t slot = base_r[key] + atomicAdd(&cur_r[key], 1u);
        const uint32_t sb = base_s[key] + atomicAdd(&cur_s[key], k);
        g_read_rec[slot] = make_uint4(sb, k, pstart[r], pend[r]);
        g_orig[slot] = (uint32_t)r;
        for (uint32_t i = 0; i < k; ++i) { g_node_id[sb + i] = node_id[b + i]; g_step_read[sb + i] = slot; }
    }
}

int build_step_read(Ctx *ctx, Reads *rd, uint32_t max_node_id) {
    if (rd->R == 0 || rd->T == 0) return 0;
    int shift = 5;
    while (((uint64_t)max_node_id >> shift) + 1 > (1u << 20)) ++shift;
    const uint32_t NB = (uint32_t)(max_node_id >> shift) + 1;
    DevBuf<uint32_t> cnt, scan_tmp;
    PTX_HIP(ctx, cnt.alloc(4ull * NB + 8));
    uint32_t *cnt_r = cnt.p, *cnt_s = cnt_r + NB, *base_r = cnt_s + NB, *base_s = base_r + NB;
    PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(NB)));
    PTX_HIP(ctx, rd->d_g_read_rec.alloc(rd->R)); PTX_HIP(ctx, rd->d_g_orig.alloc(rd->R));
    PTX_HIP(ctx, rd->d_g_node_id.alloc(rd->T)); PTX_HIP(ctx, rd->d_g_step_read.alloc(rd->T));
    PTX_HIP(ctx, hipMemsetAsync(cnt_r, 0, 2ull * NB * sizeof(uint32_t), ctx->stream));
    int gridR = grid_for(rd->R, 256, ctx->n_cu * 8);
    hipLaunchKernelGGL(group_count_kernel, dim3(gridR), dim3(256), 0, ctx->stream, rd->R, rd->d_step_off.p, rd->d_node_id.p, shift, cnt_r, cnt_s);
    PTX_TRY(exclusive_scan_u32(ctx, cnt_r, base_r, NB, scan_tmp.p, nullptr));
    PTX_TRY(exclusive_scan_u32(ctx, cnt_s, base_s, NB, scan_tmp.p, nullptr));
    PTX_HIP(ctx, hipMemsetAsync(cnt_r, 0, 2ull * NB * sizeof(uint32_t), ctx->stream));   // reused as cursors
    hipLaunchKernelGGL(group_scatter_kernel, dim3(gridR), dim3(256), 0, ctx->stream, rd->R, rd->d_step_off.p, rd->d_node_id.p, rd->d_pstart.p,
                       rd->d_pend.p, shift, base_r, base_s, cnt_r, cnt_s, rd->d_g_read_rec.p, rd->d_g_orig.p, rd->d_g_node_id.p, rd->d_g_step_read.p);
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // cnt / scan_tmp are released on return
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
    if (rd->R && rd->T) {
        int grid = (int)((rd->T + COV_CHUNK - 1) / COV_CHUNK);
        KTimer t(ctx, "coverage_step_kernel");
#define COVS_ARGS rd->T, rd->d_g_step_read.p, rd->d_g_read_rec.p, rd->d_g_orig.p, rd->d_g_node_id.p, rd->d_species.p,                    \
                  rd->has_flags ? rd->d_flags.p : nullptr, d_active, db->d_sp_first_id.p, db->d_node_base.p, db->d_node_rec.p, db->d_bases.p, \
                  db->d_bitmap.p, db->d_trio_node.p, db->d_trio_ent.p, db->d_trio_bases.p, d_abort
        if (with_trio && db->U) hipLaunchKernelGGL((coverage_step_kernel<true>), dim3(grid), dim3(COV_BLOCK), 0, ctx->stream, COVS_ARGS);
        else hipLaunchKernelGGL((coverage_step_kernel<false>), dim3(grid), dim3(COV_BLOCK), 0, ctx->stream, COVS_ARGS);
#undef COVS_ARGS
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
