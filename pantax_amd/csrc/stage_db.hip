// stage_db.hip -- device side of loading a resident DB (pipeline seam, a6; db_image.cpp): the big arrays arrive as they
// lie in the files (32-bit node lengths, walks of species-local node ids); everything derived from them is made HERE, not by host
// passes over V and P -- the per-node tables (bit offsets = global prefix of the lengths, the packed node records), the checks the
// reference makes while parsing (a node of length 0, profile.rs:494; a walk that leaves its graph would panic at :849) and the
// identical-walk test of first_filter_paths (profile.rs:1188-1190).  Nothing here is on the step's path.
#include "common.hpp"
#include "primitives.hpp"
#include "wave.hpp"

namespace ptx {

namespace {
constexpr int NT_PER = 16, NT_TILE = 256 * NT_PER;   // nodes per thread / per workgroup

// sum of the lengths of every tile of NT_TILE nodes (64-bit: a GPU holds up to 2^40 graph bases); a zero length raises the flag
__global__ void __launch_bounds__(256) node_len_tile_sum_kernel(const uint32_t *__restrict__ len, uint64_t V, unsigned long long *__restrict__ sums,
                                                                uint32_t *__restrict__ zero_flag) {
    __shared__ unsigned long long s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * NT_TILE;
    unsigned long long c = 0;
    bool zero = false;
#pragma unroll
    for (int k = 0; k < NT_PER; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < V) { const uint32_t l = len[i]; c += l; zero = zero || l == 0u; }
    }
    if (__any(zero) && (threadIdx.x & 63) == 0) *zero_flag = 1u;
    c = wave_reduce(c, [](unsigned long long x, unsigned long long y) { return x + y; });
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// exclusive prefix of the tile sums by ONE workgroup (V / 4096 entries: 8e4 at 3e8 nodes); entry n_tiles = the total
__global__ void __launch_bounds__(1024) u64_scan_kernel(unsigned long long *__restrict__ sums, uint32_t n_tiles) {
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_tiles ? sums[i] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned long long woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const unsigned long long t = s_wave[w]; if (w < wave) woff += t; tot += t; }
        const unsigned long long carry = s_carry;
        if (i < n_tiles) sums[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[n_tiles] = s_carry;
}
// bit offsets + node records of a tile (element i of a tile = stretch k, thread t: coalesced loads and stores)
__global__ void __launch_bounds__(256) node_tables_kernel(const uint32_t *__restrict__ len, uint64_t V, const unsigned long long *__restrict__ sums,
                                                          uint64_t *__restrict__ bit_off, uint4 *__restrict__ node_rec) {
    __shared__ unsigned long long s_w[2][4];
    const uint64_t base = (uint64_t)blockIdx.x * NT_TILE;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long run = sums[blockIdx.x];
#pragma unroll 4
    for (int k = 0; k < NT_PER; ++k) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        const uint32_t l = i < V ? len[i] : 0u;
        unsigned long long incl = l;                          // 64-bit throughout: a node may be up to 2^32 - 1 bases long (load time, not the step)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += t; }
        if (lane == 63) s_w[k & 1][wave] = incl;
        __syncthreads();
        unsigned long long woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned long long t = s_w[k & 1][w]; woff += w < (int)wave ? t : 0ull; tot += t; }
        if (i < V) {
            const unsigned long long bo = run + woff + (incl - l);
            bit_off[i] = bo;
            node_rec[i] = nr_make(bo, l);
        }
        run += tot;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) bit_off[V] = sums[gridDim.x];
}

// one workgroup per haplotype: every node of the walk must lie inside its species' graph (profile.rs:849 would panic)
__global__ void __launch_bounds__(256) walk_check_kernel(const uint64_t *__restrict__ path_off, const uint32_t *__restrict__ path_nodes,
                                                         const uint32_t *__restrict__ hap_species, const uint32_t *__restrict__ node_base,
                                                         uint32_t *__restrict__ bad /* [0] = 1 + first offending hap */) {
    const uint32_t h = blockIdx.x;
    const uint32_t s = hap_species[h];
    const uint32_t nv = node_base[s + 1] - node_base[s];
    uint32_t mx = 0;
    for (uint64_t q = path_off[h] + threadIdx.x; q < path_off[h + 1]; q += 256) mx = max(mx, path_nodes[q]);
    mx = wave_reduce(mx, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    if ((threadIdx.x & 63) == 0 && path_off[h + 1] > path_off[h] && mx >= nv) atomicMin(bad, h + 1);
}

// all_same[s] (preset to 1 for species of two or more haplotypes): cleared by any haplotype whose walk differs from the first
// haplotype's (profile.rs:1188-1190: `paths_vec.all(|x| x == first_path)`).  One workgroup per haplotype; it stops at the first
// stretch that differs, so a database of distinct strains costs one stretch per haplotype.
__global__ void __launch_bounds__(256) walks_same_kernel(const uint64_t *__restrict__ path_off, const uint32_t *__restrict__ path_nodes,
                                                         const uint32_t *__restrict__ hap_species, const uint64_t *__restrict__ hap_off,
                                                         uint8_t *__restrict__ all_same) {
    const uint32_t h = blockIdx.x, s = hap_species[h];
    const uint64_t h0 = hap_off[s];
    if (h == h0) return;
    const uint64_t a0 = path_off[h0], a1 = path_off[h0 + 1], b0 = path_off[h], b1 = path_off[h + 1];
    if (a1 - a0 != b1 - b0) { if (threadIdx.x == 0) all_same[s] = 0; return; }
    for (uint64_t q = 0; q < a1 - a0; q += 256) {
        const uint64_t i = q + threadIdx.x;
        const bool diff = i < a1 - a0 && path_nodes[a0 + i] != path_nodes[b0 + i];
        if (__syncthreads_or(diff)) { if (threadIdx.x == 0) all_same[s] = 0; return; }
    }
}
}  // namespace

int node_tables_launch(Ctx *ctx, Db *db, uint32_t *d_flags) {
    const uint64_t V = db->V;
    const uint32_t n_tiles = (uint32_t)((V + NT_TILE - 1) / NT_TILE);
    DevBuf<unsigned long long> sums;
    PTX_HIP(ctx, sums.alloc((size_t)n_tiles + 1));
    if (n_tiles) {
        hipLaunchKernelGGL(node_len_tile_sum_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, (const uint32_t *)db->d_node_len.p, V, sums.p, d_flags);
        hipLaunchKernelGGL(u64_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums.p, n_tiles);
        hipLaunchKernelGGL(node_tables_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, (const uint32_t *)db->d_node_len.p, V, (const unsigned long long *)sums.p,
                           db->d_bit_off.p, db->d_node_rec.p);
    } else PTX_HIP(ctx, hipMemsetAsync(db->d_bit_off.p, 0, sizeof(uint64_t), ctx->stream));
    if (db->H) {
        hipLaunchKernelGGL(walk_check_kernel, dim3((uint32_t)db->H), dim3(256), 0, ctx->stream, db->d_path_off.p, db->d_path_nodes.p,
                           db->d_hap_species.p, db->d_node_base.p, d_flags + 1);
        hipLaunchKernelGGL(walks_same_kernel, dim3((uint32_t)db->H), dim3(256), 0, ctx->stream, db->d_path_off.p, db->d_path_nodes.p,
                           db->d_hap_species.p, db->d_hap_off.p, db->d_all_same.p);
    }
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `sums` goes out of scope
    return 0;
}

}  // namespace ptx
