// stage_db.hip -- device side of loading a resident DB (pipeline seam, a6; db_image.cpp): the big arrays arrive as they
// lie in the files (32-bit node lengths, walks of species-local node ids); everything derived from them is made HERE, not by host
// passes over V and P -- the per-node tables (bit offsets = global prefix of the lengths, the packed node records), the checks the
// reference makes while parsing (a node of length 0, profile.rs:494; a walk that leaves its graph would panic at :849) and the
// identical-walk test of first_filter_paths (profile.rs:1188-1190).  Nothing here is on the step's path.
#include <memory>
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

// ---- image format 4 (round 6): packed walks / 16-bit node lengths -> the arrays the kernels read ----
// One WAVE per block of PK_BLOCK positions: its species by a binary search over the species' first blocks (wave-uniform loads), then four rounds
// of 64 zigzag deltas -> a wave prefix sum on top of the block's first id and the rounds before.  A block whose width is not 1, 2 or 4 (a damaged
// file) is written as 0xFFFFFFFF: the walk check that follows every upload reports it.
__global__ void __launch_bounds__(256) walks_unpack_kernel(const UnpackSpecies *__restrict__ table, uint32_t n_species, uint32_t n_blocks, const uint32_t *__restrict__ first,
                                                           const uint32_t *__restrict__ off, const uint8_t *__restrict__ payload, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint32_t gb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (gb >= n_blocks) return;
    uint32_t lo = 0, hi = n_species - 1;                               // the last species whose first block is <= gb
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (table[mid].blk_base <= gb) lo = mid; else hi = mid - 1; }
    const UnpackSpecies sp = table[lo];
    const uint32_t b = gb - sp.blk_base;
    const uint32_t o0 = off[sp.off_base + b], o1 = off[sp.off_base + b + 1], w = o1 - o0;
    const uint64_t pos0 = (uint64_t)b * PK_BLOCK;
    const uint32_t n_in = (uint32_t)min((uint64_t)PK_BLOCK, (uint64_t)sp.n_steps - pos0);
    const uint8_t *src = payload + ((uint64_t)sp.payload_base + o0) * PK_UNIT;
    uint32_t *dst = out + (uint64_t)sp.out_base + pos0;
    const bool good = w == 1u || w == 2u || w == 4u;
    uint32_t carry = first[gb];
#pragma unroll
    for (uint32_t r = 0; r < PK_BLOCK / 64; ++r) {
        const uint32_t i = r * 64u + (uint32_t)lane;
        uint32_t zz = 0;
        if (good && i < n_in) zz = w == 1u ? (uint32_t)src[i] : w == 2u ? (uint32_t)reinterpret_cast<const uint16_t *>(src)[i] : reinterpret_cast<const uint32_t *>(src)[i];
        const uint32_t d = (zz >> 1) ^ (0u - (zz & 1u));               // zigzag -> two's complement
        const uint32_t incl = wave_incl_scan_dpp(d);
        if (i < n_in) dst[i] = good ? carry + incl : 0xFFFFFFFFu;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}
// 16-bit node lengths -> 32-bit: a wave takes 1024 consecutive source entries (the species' stretches lie back to back, each padded to 4 bytes)
__global__ void __launch_bounds__(256) lens_widen_kernel(const WidenSpecies *__restrict__ table, uint32_t n_species, uint64_t n_total, const uint16_t *__restrict__ src,
                                                         uint32_t *__restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const uint64_t e0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024;
    if (e0 >= n_total) return;
    uint32_t lo = 0, hi = n_species - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (table[mid].src_base <= e0 + (uint64_t)lane) lo = mid; else hi = mid - 1; }
    uint32_t s = lo;
    for (uint32_t r = 0; r < 16; ++r) {
        const uint64_t i = e0 + (uint64_t)r * 64 + (uint64_t)lane;
        if (i >= n_total) break;
        while (s + 1 < n_species && table[s + 1].src_base <= i) ++s;
        const uint64_t k = i - table[s].src_base;
        if (k < table[s].n) dst[table[s].dst_base + k] = (uint32_t)src[i];     // (the pad entry behind an odd stretch belongs to nobody)
    }
}
}  // namespace

int walks_unpack_launch(Ctx *ctx, const UnpackSpecies *d_table, uint32_t n_species, uint32_t n_blocks, const uint32_t *d_first, const uint32_t *d_off,
                        const uint8_t *d_payload, uint32_t *d_path_nodes, hipStream_t stream) {
    if (!n_blocks || !n_species) return 0;
    std::unique_ptr<KTimer> t(stream ? nullptr : new KTimer(ctx, "walks_unpack_kernel"));   // (the launch timers belong to ctx->stream and its thread)
    hipLaunchKernelGGL(walks_unpack_kernel, dim3((n_blocks + 3) / 4), dim3(256), 0, stream ? stream : ctx->stream, d_table, n_species, n_blocks, d_first, d_off, d_payload, d_path_nodes);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}
int lens_widen_launch(Ctx *ctx, const WidenSpecies *d_table, uint32_t n_species, uint64_t n_total, const uint16_t *d_len16, uint32_t *d_node_len, hipStream_t stream) {
    if (!n_total || !n_species) return 0;
    std::unique_ptr<KTimer> t(stream ? nullptr : new KTimer(ctx, "lens_widen_kernel"));
    hipLaunchKernelGGL(lens_widen_kernel, dim3((uint32_t)((n_total + 4095) / 4096)), dim3(256), 0, stream ? stream : ctx->stream, d_table, n_species, n_total, d_len16, d_node_len);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

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
