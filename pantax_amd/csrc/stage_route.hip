// stage_route.hip -- SURVEY 8e, reads over N GPUs: every rank tokenises and bins a 1/N slice of the GAF, then the packed
// records travel to the rank that OWNS their species (all-to-all(v) over xGMI, or whatever the host plugs in) and the
// owner rebuilds resident reads from what it received.  Reference analogue: group_reads_by_species profile.rs:439-463
// (one frame per species) -- here one message per (source rank, owner rank).
//
// Message of source i for owner j, all 32-bit words:
//   [n_steps of every read | pstart | pend | qlen | mapq]  (5 x n_reads, column after column)  [node ids of the walks, n_steps]
// Reads keep their order (the order of the source's slice): the pack is a stable partition, so a run over N ranks
// hands every owner exactly the reads a one-process run would have looked up for its species, in the same order.
// Dropped on the way: reads binned "U", reads of species no rank owns (not selected), reads carrying a drop flag
// (null field / duplicate id, profile.rs:380-437) -- none of them reaches get_node_abundances in the reference either.
//
// Two launches over 256-read tiles around one chained scan:
//   route_count_kernel   : per tile, reads and steps per owner                       -> table[2W][n_tiles]
//   exclusive scan of the table (row-major: the scan of row d continues row d-1, so owner blocks are contiguous)
//   route_scatter_kernel : rank of a read among the tile's reads of the same owner by wave ballots, of its steps by a
//                          masked wave scan; columns and walk written at the final place.  Walks of more than 64 steps
//                          are copied by the whole workgroup (coalesced), shorter ones by their thread.
#include <algorithm>
#include <memory>
#include "common.hpp"
#include "primitives.hpp"
#include "wave.hpp"

namespace ptx {

constexpr int ROUTE_MAXW = 64;     // owners per call (one node has 8; LDS counters are sized for this)
constexpr int ROUTE_TILE = 256;

__device__ __forceinline__ int route_dest(uint64_t r, uint64_t R, const int32_t *__restrict__ species, const int32_t *__restrict__ owner,
                                          const uint8_t *__restrict__ flags) {
    if (r >= R) return -1;
    const int sp = species[r];
    if (sp < 0 || (flags && flags[r])) return -1;
    return owner[sp];
}

__global__ void __launch_bounds__(ROUTE_TILE) route_count_kernel(uint64_t R, const int32_t *__restrict__ species, const int32_t *__restrict__ owner,
                                                                 const uint8_t *__restrict__ flags, const uint32_t *__restrict__ step_off, int W,
                                                                 uint32_t n_tiles, uint32_t *__restrict__ table) {
    __shared__ uint32_t s_cnt[2 * ROUTE_MAXW];
    if (threadIdx.x < 2 * W) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * ROUTE_TILE + threadIdx.x;
    const int d = route_dest(r, R, species, owner, flags);
    if (d >= 0) {
        atomicAdd(&s_cnt[d], 1u);
        atomicAdd(&s_cnt[W + d], step_off[r + 1] - step_off[r]);
    }
    __syncthreads();
    if (threadIdx.x < 2 * W) table[(size_t)threadIdx.x * n_tiles + blockIdx.x] = s_cnt[threadIdx.x];
}

// totals[k] = sum of row k of the table, from its exclusive scan (row k ends where row k+1 begins)
__global__ void route_totals_kernel(int rows, uint32_t n_tiles, const uint32_t *__restrict__ scanned, const uint32_t *__restrict__ grand_total,
                                    uint32_t *__restrict__ totals) {
    const int k = threadIdx.x;
    if (k < rows) {
        const uint32_t b = scanned[(size_t)k * n_tiles], e = k + 1 < rows ? scanned[(size_t)(k + 1) * n_tiles] : *grand_total;
        totals[k] = e - b;
    }
}

__global__ void __launch_bounds__(ROUTE_TILE) route_scatter_kernel(uint64_t R, const int32_t *__restrict__ species, const int32_t *__restrict__ owner,
                                                                   const uint8_t *__restrict__ flags, const uint32_t *__restrict__ step_off,
                                                                   const uint32_t *__restrict__ node_id, const uint32_t *__restrict__ pstart,
                                                                   const uint32_t *__restrict__ pend, const uint32_t *__restrict__ qlen,
                                                                   const uint8_t *__restrict__ mapq, int W, uint32_t n_tiles,
                                                                   const uint32_t *__restrict__ scanned, const uint64_t *__restrict__ blk_off /*[W] words*/,
                                                                   const uint32_t *__restrict__ blk_reads /*[W]*/, uint32_t *__restrict__ out) {
    __shared__ uint32_t s_w[ROUTE_TILE / 64][2 * ROUTE_MAXW];   // per wave: reads and steps per owner
    __shared__ uint32_t s_long[ROUTE_TILE][3];                  // walks of > 64 steps: {source begin, #steps, owner} + destination below
    __shared__ uint64_t s_long_dst[ROUTE_TILE];
    __shared__ uint32_t s_n_long;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (ROUTE_TILE / 64) * 2 * ROUTE_MAXW; i += ROUTE_TILE) (&s_w[0][0])[i] = 0;
    if (threadIdx.x == 0) s_n_long = 0;
    __syncthreads();
    const uint64_t r = (uint64_t)blockIdx.x * ROUTE_TILE + threadIdx.x;
    const int d = route_dest(r, R, species, owner, flags);
    uint32_t b = 0, n = 0;
    if (d >= 0) { b = step_off[r]; n = step_off[r + 1] - b; }
    // rank among the wave's earlier reads of the same owner, and the steps those hold
    uint32_t rank = 0, srank = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (unsigned long long todo = __ballot(d >= 0); todo;) {
        const int dd = __shfl(d, __ffsll((long long)todo) - 1);
        const unsigned long long m = __ballot(d == dd);
        const uint32_t incl = wave_incl_scan_dpp(d == dd ? n : 0u);
        if (d == dd) { rank = (uint32_t)__popcll(m & lt); srank = incl - n; }
        if (lane == 63 - __clzll((long long)m)) { s_w[wave][dd] = (uint32_t)__popcll(m); s_w[wave][W + dd] = incl; }
        todo &= ~m;
    }
    __syncthreads();
    if (d >= 0) {
        uint32_t wr = 0, ws = 0;
        for (int w = 0; w < wave; ++w) { wr += s_w[w][d]; ws += s_w[w][W + d]; }
        const uint32_t nr = blk_reads[d];
        const uint32_t pr = scanned[(size_t)d * n_tiles + blockIdx.x] - scanned[(size_t)d * n_tiles] + wr + rank;
        const uint32_t ps = scanned[(size_t)(W + d) * n_tiles + blockIdx.x] - scanned[(size_t)(W + d) * n_tiles] + ws + srank;
        uint32_t *blk = out + blk_off[d];
        blk[pr] = n;
        blk[(size_t)nr + pr] = pstart[r];
        blk[2 * (size_t)nr + pr] = pend[r];
        blk[3 * (size_t)nr + pr] = qlen[r];
        blk[4 * (size_t)nr + pr] = mapq[r];
        uint32_t *dst = blk + 5 * (size_t)nr + ps;
        if (n <= 64u) {
            for (uint32_t i = 0; i < n; ++i) dst[i] = node_id[b + i];
        } else {
            const uint32_t k = atomicAdd(&s_n_long, 1u);
            s_long[k][0] = b; s_long[k][1] = n;
            s_long_dst[k] = (uint64_t)(dst - out);
        }
    }
    __syncthreads();
    const uint32_t nl = s_n_long;
    for (uint32_t k = 0; k < nl; ++k) {
        const uint32_t sb = s_long[k][0], sn = s_long[k][1];
        uint32_t *dst = out + s_long_dst[k];
        for (uint32_t i = threadIdx.x; i < sn; i += ROUTE_TILE) dst[i] = node_id[sb + i];
    }
}

__global__ void __launch_bounds__(256) route_mapq_kernel(uint64_t n, const uint32_t *__restrict__ in, uint8_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = (uint8_t)in[i];
}
__global__ void __launch_bounds__(256) route_max_kernel(uint64_t n, const uint32_t *__restrict__ v, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) m = max(m, v[i]);
    m = wave_reduce(m, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// reads (binned against `db`) -> one message per owner.  owner_of_species[s] = rank that owns species s of `db`, or < 0.
int route_pack(Ctx *ctx, const Db *db, const Reads *rd, const int32_t *owner_of_species, int W, Route &rt) {
    if (W < 1 || W > ROUTE_MAXW) return fail(ctx, PANTAX_HIP_E_LIMIT, "route_pack: %d owners (at most %d)", W, ROUTE_MAXW);
    if (!rd->binned) return fail(ctx, PANTAX_HIP_E_STATE, "route_pack: call pantax_hip_bin_reads on these reads first");
    PTX_TRY(species_ensure(ctx, const_cast<Reads *>(rd)));   // resident (grouped) reads keep the species per slot
    for (uint32_t s = 0; s < db->S; ++s)
        if (owner_of_species[s] >= W) return fail(ctx, PANTAX_HIP_E_INVALID, "route_pack: species %u is owned by rank %d of %d", s, owner_of_species[s], W);
    rt.W = W;
    rt.n_reads.assign(W, 0); rt.n_steps.assign(W, 0); rt.word_off.assign(W + 1, 0);
    rt.h_valid = false;
    if (rd->R == 0) { PTX_HIP(ctx, rt.d_send.alloc(1)); return 0; }
    const uint32_t n_tiles = (uint32_t)((rd->R + ROUTE_TILE - 1) / ROUTE_TILE);
    const uint64_t n_tab = (uint64_t)2 * W * n_tiles;
    DevBuf<int32_t> d_owner;
    DevBuf<uint32_t> table, scanned, scan_tmp, totals, blk_reads;
    DevBuf<uint64_t> blk_off;
    PTX_TRY(upload_small(ctx, d_owner, owner_of_species, db->S));
    PTX_HIP(ctx, table.alloc(n_tab)); PTX_HIP(ctx, scanned.alloc(n_tab)); PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(n_tab)));
    PTX_HIP(ctx, totals.alloc(2 * W + 1));
    const uint8_t *flags = rd->has_flags ? rd->d_flags.p : nullptr;
    {
        KTimer t(ctx, "route_count_kernel");
        hipLaunchKernelGGL(route_count_kernel, dim3(n_tiles), dim3(ROUTE_TILE), 0, ctx->stream, rd->R, rd->d_species.p, d_owner.p, flags, rd->d_step_off.p, W,
                           n_tiles, table.p);
    }
    PTX_TRY(exclusive_scan_u32(ctx, table.p, scanned.p, n_tab, scan_tmp.p, totals.p + 2 * W));
    hipLaunchKernelGGL(route_totals_kernel, dim3(1), dim3(2 * ROUTE_MAXW), 0, ctx->stream, 2 * W, n_tiles, scanned.p, totals.p + 2 * W, totals.p);
    std::vector<uint32_t> h_tot(2 * W);
    PTX_TRY(download(ctx, h_tot.data(), totals.p, (size_t)2 * W));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> h_blk_reads(W);
    for (int d = 0; d < W; ++d) {
        rt.n_reads[d] = h_tot[d]; rt.n_steps[d] = h_tot[W + d];
        h_blk_reads[d] = h_tot[d];
        rt.word_off[d + 1] = rt.word_off[d] + 5ull * rt.n_reads[d] + rt.n_steps[d];
    }
    PTX_HIP(ctx, rt.d_send.alloc(rt.word_off[W] ? rt.word_off[W] : 1));
    PTX_TRY(upload_small(ctx, blk_off, rt.word_off.data(), (size_t)W));
    PTX_TRY(upload_small(ctx, blk_reads, h_blk_reads.data(), (size_t)W));
    {
        KTimer t(ctx, "route_scatter_kernel");
        hipLaunchKernelGGL(route_scatter_kernel, dim3(n_tiles), dim3(ROUTE_TILE), 0, ctx->stream, rd->R, rd->d_species.p, d_owner.p, flags, rd->d_step_off.p,
                           rd->d_node_id.p, rd->d_pstart.p, rd->d_pend.p, rd->d_qlen.p, rd->d_mapq.p, W, n_tiles, scanned.p, blk_off.p, blk_reads.p,
                           rt.d_send.p);
    }
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the scratch tables are released on return
    return 0;
}

// messages received from `W` sources (source-major, each in the layout above; device memory) -> resident reads
int reads_from_routed(Ctx *ctx, const uint32_t *d_recv, int W, const uint64_t *n_reads_from, const uint64_t *n_steps_from, bool group, Reads *rd) {
    uint64_t R = 0, T = 0;
    for (int k = 0; k < W; ++k) { R += n_reads_from[k]; T += n_steps_from[k]; }
    if (T >= 0xFFFFFFFFull || R >= 0xFFFFFFFFull)
        return fail(ctx, PANTAX_HIP_E_LIMIT, "reads_from_routed: %llu reads / %llu steps exceed the 32-bit offsets of one batch", (unsigned long long)R, (unsigned long long)T);
    rd->R = R; rd->T = T;
    PTX_HIP(ctx, rd->d_step_off.alloc(R + 1)); PTX_HIP(ctx, rd->d_node_id.alloc(T ? T : 1));
    PTX_HIP(ctx, rd->d_pstart.alloc(R ? R : 1)); PTX_HIP(ctx, rd->d_pend.alloc(R ? R : 1)); PTX_HIP(ctx, rd->d_qlen.alloc(R ? R : 1));
    PTX_HIP(ctx, rd->d_mapq.alloc(R ? R : 1));
    rd->has_flags = false;
    rd->binned = false;
    DevBuf<uint32_t> nst, mq, scan_tmp, d_max;
    PTX_HIP(ctx, nst.alloc(R + 1)); PTX_HIP(ctx, mq.alloc(R ? R : 1)); PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(R + 1))); PTX_HIP(ctx, d_max.alloc(1));
    auto d2d = [&](void *dst, const void *src, uint64_t words) {
        return words ? hipMemcpyAsync(dst, src, words * 4, hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess;
    };
    uint64_t r0 = 0, t0 = 0, w0 = 0;
    for (int k = 0; k < W; ++k) {
        const uint64_t nr = n_reads_from[k], nt = n_steps_from[k];
        const uint32_t *blk = d_recv + w0;
        PTX_HIP(ctx, d2d(nst.p + r0, blk, nr));
        PTX_HIP(ctx, d2d(rd->d_pstart.p + r0, blk + nr, nr));
        PTX_HIP(ctx, d2d(rd->d_pend.p + r0, blk + 2 * nr, nr));
        PTX_HIP(ctx, d2d(rd->d_qlen.p + r0, blk + 3 * nr, nr));
        PTX_HIP(ctx, d2d(mq.p + r0, blk + 4 * nr, nr));
        PTX_HIP(ctx, d2d(rd->d_node_id.p + t0, blk + 5 * nr, nt));
        r0 += nr; t0 += nt; w0 += 5 * nr + nt;
    }
    PTX_HIP(ctx, hipMemsetAsync(nst.p + R, 0, sizeof(uint32_t), ctx->stream));
    PTX_TRY(exclusive_scan_u32(ctx, nst.p, rd->d_step_off.p, R + 1, scan_tmp.p, nullptr));   // step_off[R] = all steps
    PTX_HIP(ctx, hipMemsetAsync(d_max.p, 0, sizeof(uint32_t), ctx->stream));
    if (R) hipLaunchKernelGGL(route_mapq_kernel, dim3(grid_for(R, 256, ctx->n_cu * 4)), dim3(256), 0, ctx->stream, R, mq.p, rd->d_mapq.p);
    if (T) hipLaunchKernelGGL(route_max_kernel, dim3(grid_for(T, 256, ctx->n_cu * 4)), dim3(256), 0, ctx->stream, T, rd->d_node_id.p, d_max.p);
    uint32_t h_chk[2] = {0, 0};
    PTX_TRY(download(ctx, &h_chk[0], d_max.p, 1));
    PTX_TRY(download(ctx, &h_chk[1], rd->d_step_off.p + R, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((uint64_t)h_chk[1] != T)
        return fail(ctx, PANTAX_HIP_E_INVALID, "reads_from_routed: the messages hold %u steps in their reads but %llu were announced", h_chk[1], (unsigned long long)T);
    if (group) {
        PTX_TRY(build_step_read(ctx, rd, h_chk[0]));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

}  // namespace ptx

struct pantax_hip_route : ptx::Route {};
using namespace ptx;

extern "C" {

int pantax_hip_reads_route_pack(pantax_hip_ctx *ctx, const pantax_hip_db *db, const pantax_hip_reads *reads, const int32_t *owner_of_species,
                                int world_size, pantax_hip_route **out, uint64_t *n_reads_to, uint64_t *n_steps_to) {
    if (!ctx || !db || !reads || !owner_of_species || !out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    std::unique_ptr<pantax_hip_route> rt(new pantax_hip_route());
    PTX_TRY(route_pack(ctx, db, reads, owner_of_species, world_size, *rt));
    for (int d = 0; d < world_size; ++d) {
        if (n_reads_to) n_reads_to[d] = rt->n_reads[d];
        if (n_steps_to) n_steps_to[d] = rt->n_steps[d];
    }
    *out = rt.release();
    return 0;
}

int pantax_hip_route_buffer(pantax_hip_ctx *ctx, pantax_hip_route *route, int on_device, const uint32_t **buf_out, uint64_t *word_off_out) {
    if (!ctx || !route || !buf_out) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    const uint64_t words = route->word_off[route->W];
    if (word_off_out) for (int d = 0; d <= route->W; ++d) word_off_out[d] = route->word_off[d];
    if (on_device) { *buf_out = route->d_send.p; return 0; }
    if (!route->h_valid) {
        PTX_HIP(ctx, route->h_send.reserve(words ? words * 4 : 4));
        if (words) PTX_HIP(ctx, hipMemcpyAsync(route->h_send.p, route->d_send.p, words * 4, hipMemcpyDeviceToHost, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        route->h_valid = true;
    }
    *buf_out = reinterpret_cast<const uint32_t *>(route->h_send.p);
    return 0;
}

void pantax_hip_route_free(pantax_hip_ctx *ctx, pantax_hip_route *route) {
    std::unique_lock<std::recursive_mutex> lk;
    if (ctx) lk = std::unique_lock<std::recursive_mutex>(ctx->mu);
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    delete route;
}

int pantax_hip_reads_from_routed(pantax_hip_ctx *ctx, const uint32_t *recv, int on_device, int world_size, const uint64_t *n_reads_from,
                                 const uint64_t *n_steps_from, pantax_hip_reads **out) {
    if (!ctx || !n_reads_from || !n_steps_from || !out || world_size < 1) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    uint64_t words = 0;
    for (int k = 0; k < world_size; ++k) words += 5 * n_reads_from[k] + n_steps_from[k];
    if (words && !recv) return PANTAX_HIP_E_INVALID;
    DevBuf<uint32_t> staged;
    const uint32_t *d_recv = recv;
    if (!on_device) {
        PTX_HIP(ctx, staged.alloc(words ? words : 1));
        if (words) PTX_TRY(upload_big(ctx, staged.p, recv, words * 4));
        d_recv = staged.p;
    }
    std::unique_ptr<pantax_hip_reads> rd(new pantax_hip_reads());
    PTX_TRY(reads_from_routed(ctx, d_recv, world_size, n_reads_from, n_steps_from, true, rd.get()));
    *out = rd.release();
    return 0;
}

}  // extern "C"
