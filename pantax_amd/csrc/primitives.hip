// primitives.hip -- exclusive scan + stable LSD radix sort (see primitives.hpp).
// Both are HBM-streaming: scan moves 2 reads + 1 write of the input; each sort pass moves
// (8*nw + 4) bytes per record in the histogram sweep and twice that in the scatter sweep.
#include <algorithm>
#include "primitives.hpp"
#include "wave.hpp"
#include "scan_chained.hpp"

namespace ptx {

size_t scan_tmp_elems(uint64_t n) { return 16; }   // the scan workspace lives in the context (ctx->d_scan_ws); callers' scratch is unused

template <class T>
struct ScanLoadArray { const T *in; __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return (uint32_t)in[i]; } };
struct ScanStoreArray { uint32_t *out; __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t) const { out[i] = excl; } };
template <class T>
static int exclusive_scan_impl(Ctx *ctx, const T *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp, uint32_t *d_total) {
    (void)d_tmp;
    return exclusive_scan_fn(ctx, ScanLoadArray<T>{d_in}, ScanStoreArray{d_out}, n, d_total, "scan_chained_kernel");
}
int exclusive_scan_u32(Ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp, uint32_t *d_total) {
    return exclusive_scan_impl<uint32_t>(ctx, d_in, d_out, n, d_tmp, d_total);
}
int exclusive_scan_u8(Ctx *ctx, const uint8_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp, uint32_t *d_total) {
    return exclusive_scan_impl<uint8_t>(ctx, d_in, d_out, n, d_tmp, d_total);
}

// ---------------------------------------------------------------------------------------------
// zero fill of large device ranges: 16-byte stores from every CU.  The runtime's fill kernel behind hipMemsetAsync
// reached about 1 TB/s on the arenas of a 100-species step (0.4 GB in 0.40 ms); a step zeroes 1-2 GB.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4 *__restrict__ p, uint64_t n16, uint32_t word) {
    const uint4 z = make_uint4(word, word, word, word);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) { p[i] = z; p[i + stride] = z; p[i + 2 * stride] = z; p[i + 3 * stride] = z; }
    for (; i < n16; i += stride) p[i] = z;
}
// every byte of [ptr, ptr + bytes) = `byte` (0x00 / 0xFF fills of the upload-time tables: 9 GB of visit-table pads at 1e4 strains)
int byte_fill(Ctx *ctx, void *ptr, int byte, size_t bytes) {
    if (bytes == 0) return 0;
    uint8_t *p = static_cast<uint8_t *>(ptr);
    if (bytes < (1u << 20)) { PTX_HIP(ctx, hipMemsetAsync(p, byte, bytes, ctx->stream)); return 0; }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
    if (head) PTX_HIP(ctx, hipMemsetAsync(p, byte, head, ctx->stream));
    const uint64_t n16 = (bytes - head) / 16;
    const size_t tail = bytes - head - n16 * 16;
    const uint32_t word = 0x01010101u * (uint32_t)(byte & 0xFF);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid_for(n16 / 4 + 1, 256, ctx->n_cu * 16)), dim3(256), 0, ctx->stream, reinterpret_cast<uint4 *>(p + head), n16, word);
    if (tail) PTX_HIP(ctx, hipMemsetAsync(p + head + n16 * 16, byte, tail, ctx->stream));
    return 0;
}
int zero_fill(Ctx *ctx, void *ptr, size_t bytes) { return byte_fill(ctx, ptr, 0, bytes); }

// ---------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------
constexpr int SORT_BLOCK = 256;
constexpr int SORT_ROUNDS = 4;                                  // 64-record rounds per wave per tile
constexpr int SORT_TILE = SORT_BLOCK * SORT_ROUNDS;             // 1024 records
constexpr int SORT_SLOTS = (SORT_BLOCK / 64) * SORT_ROUNDS;     // (wave, round) slots per tile
constexpr int SORT_MAX_BLOCKS = 2048;                            // = SORT_BLOCK * 8: one rowscan sweep

struct SortGeom {
    uint32_t nb;
    uint64_t chunk;  // records per block, multiple of SORT_TILE
};
static SortGeom sort_geom(uint64_t n) {
    uint64_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t nb = (uint32_t)(tiles < SORT_MAX_BLOCKS ? tiles : SORT_MAX_BLOCKS);
    if (nb == 0) nb = 1;
    uint64_t tiles_per_block = (tiles + nb - 1) / nb;
    return {nb, tiles_per_block * SORT_TILE};
}
size_t sort_table_elems(uint64_t n) { return 256ull * (SORT_MAX_BLOCKS + 1); }

__global__ void __launch_bounds__(SORT_BLOCK) sort_hist_kernel(const uint64_t *__restrict__ key, int shift, uint64_t n,
                                                               uint64_t chunk, uint32_t nb, uint32_t *__restrict__ table,
                                                               const uint32_t *__restrict__ d_n) {
    __shared__ uint32_t s_hist[256];
    if (d_n) n = *d_n;   // actual record count lives on the device; n passed by the host is only the geometry bound
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    uint64_t b = (uint64_t)blockIdx.x * chunk, e = b + chunk;
    if (e > n) e = n;
    for (uint64_t i = b + threadIdx.x; i < e; i += SORT_BLOCK) atomicAdd(&s_hist[(key[i] >> shift) & 0xFF], 1u);
    __syncthreads();
    table[(uint32_t)threadIdx.x * nb + blockIdx.x] = s_hist[threadIdx.x];
}

// One workgroup per digit: exclusive scan of that digit's row of per-block counts (nb <= 2048 = one 8-item sweep)
// and the digit total; the scatter kernel turns the 256 totals into digit bases itself.  Two launches fewer per
// pass than a generic scan of the whole table.
__global__ void __launch_bounds__(SORT_BLOCK) sort_rowscan_kernel(uint32_t *__restrict__ table, uint32_t rows, uint32_t nb) {
    __shared__ uint32_t s_wave[SORT_BLOCK / 64];
    uint32_t *row = table + (size_t)blockIdx.x * nb;
    const uint32_t b = threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (b + i < nb) ? row[b + i] : 0; s += v[i]; }
    uint32_t tot;
    uint32_t off = block_excl_scan<SORT_BLOCK>(s, s_wave, &tot);
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (b + i < nb) row[b + i] = off; off += v[i]; }
    if (threadIdx.x == 0) table[(size_t)rows * nb + blockIdx.x] = tot;
}
void sort_rowscan_launch(Ctx *ctx, uint32_t *d_table, uint32_t rows, uint32_t nb) {
    hipLaunchKernelGGL(sort_rowscan_kernel, dim3(rows), dim3(SORT_BLOCK), 0, ctx->stream, d_table, rows, nb);
}

template <int NW, bool HASV>
__global__ void __launch_bounds__(SORT_BLOCK) sort_scatter_kernel(SortBufs in, SortBufs out, int word, int shift, uint64_t n,
                                                                  uint64_t chunk, uint32_t nb, const uint32_t *__restrict__ table,
                                                                  const uint32_t *__restrict__ d_n) {
    __shared__ uint32_t s_base[256];
    if (d_n) n = *d_n;
    __shared__ uint32_t s_slot[SORT_SLOTS][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    {
        __shared__ uint32_t s_wave[SORT_BLOCK / 64];
        uint32_t tot;
        uint32_t digit_base = block_excl_scan<SORT_BLOCK>(table[(size_t)256 * nb + threadIdx.x], s_wave, &tot);
        s_base[threadIdx.x] = digit_base + table[(uint32_t)threadIdx.x * nb + blockIdx.x];
    }
    uint64_t b = (uint64_t)blockIdx.x * chunk, e = b + chunk;
    if (e > n) e = n;
    if (b > e) b = e;
    for (uint64_t tile = b; tile < e; tile += SORT_TILE) {
#pragma unroll
        for (int s = 0; s < SORT_SLOTS; ++s) s_slot[s][threadIdx.x] = 0;
        __syncthreads();
        uint64_t k[SORT_ROUNDS][NW];
        uint32_t v[SORT_ROUNDS];
        uint32_t dig[SORT_ROUNDS], rank[SORT_ROUNDS];
        bool valid[SORT_ROUNDS];
#pragma unroll
        for (int j = 0; j < SORT_ROUNDS; ++j) {
            uint64_t idx = tile + (uint64_t)(wave * SORT_ROUNDS + j) * 64 + lane;
            valid[j] = idx < e;
            dig[j] = 0;
            if (valid[j]) {
#pragma unroll
                for (int w = 0; w < NW; ++w) k[j][w] = in.k[w][idx];
                if (HASV) v[j] = in.v[idx];
                uint64_t kw = k[j][0];  // static indices only: a runtime index would spill k[][] to scratch
                if (NW > 1 && word == 1) kw = k[j][1];
                if (NW > 2 && word == 2) kw = k[j][2];
                dig[j] = (uint32_t)(kw >> shift) & 0xFF;
            }
            // lanes of this wave holding the same digit (wave64 ballots, one per digit bit)
            uint64_t peers = __ballot(valid[j]);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                bool one = (dig[j] >> bit) & 1;
                uint64_t bm = __ballot(one);
                peers &= one ? bm : ~bm;
            }
            rank[j] = __popcll(peers & lt);
            if (valid[j] && rank[j] == 0) s_slot[wave * SORT_ROUNDS + j][dig[j]] = __popcll(peers);
        }
        __syncthreads();
        {   // thread d turns the per-slot counts of digit d into output offsets, in record order
            uint32_t run = s_base[threadIdx.x];
#pragma unroll
            for (int s = 0; s < SORT_SLOTS; ++s) {
                uint32_t c = s_slot[s][threadIdx.x];
                s_slot[s][threadIdx.x] = run;
                run += c;
            }
            s_base[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SORT_ROUNDS; ++j) {
            if (valid[j]) {
                uint32_t pos = s_slot[wave * SORT_ROUNDS + j][dig[j]] + rank[j];
#pragma unroll
                for (int w = 0; w < NW; ++w) out.k[w][pos] = k[j][w];
                if (HASV) out.v[pos] = v[j];
            }
        }
        __syncthreads();
    }
}

template <int NW>
static void launch_scatter(Ctx *ctx, SortBufs in, SortBufs out, int word, int shift, uint64_t n, SortGeom g, const uint32_t *table,
                           const uint32_t *d_n) {
    if (in.v) hipLaunchKernelGGL((sort_scatter_kernel<NW, true>), dim3(g.nb), dim3(SORT_BLOCK), 0, ctx->stream, in, out, word, shift, n, g.chunk, g.nb, table, d_n);
    else hipLaunchKernelGGL((sort_scatter_kernel<NW, false>), dim3(g.nb), dim3(SORT_BLOCK), 0, ctx->stream, in, out, word, shift, n, g.chunk, g.nb, table, d_n);
}

int radix_sort(Ctx *ctx, SortBufs a, SortBufs b, uint64_t n, const SortPass *passes, int n_passes, uint32_t *d_table,
               uint32_t *d_scan_tmp, bool *result_in_b, const uint32_t *d_n) {
    *result_in_b = false;
    if (n == 0 || n_passes == 0) return 0;
    if (n >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "radix_sort: %llu records exceed 32-bit positions", (unsigned long long)n);
    SortGeom g = sort_geom(n);
    SortBufs cur = a, nxt = b;
    bool in_b = false;
    for (int p = 0; p < n_passes; ++p) {
        int word = passes[p].word, shift = passes[p].shift;
        { KTimer t(ctx, "sort_hist_kernel");
          hipLaunchKernelGGL(sort_hist_kernel, dim3(g.nb), dim3(SORT_BLOCK), 0, ctx->stream, cur.k[word], shift, n, g.chunk, g.nb, d_table, d_n); }
        { KTimer t(ctx, "sort_rowscan_kernel");
          sort_rowscan_launch(ctx, d_table, 256, g.nb); }
        KTimer t(ctx, "sort_scatter_kernel");
        switch (cur.nw) {
        case 1: launch_scatter<1>(ctx, cur, nxt, word, shift, n, g, d_table, d_n); break;
        case 2: launch_scatter<2>(ctx, cur, nxt, word, shift, n, g, d_table, d_n); break;
        default: launch_scatter<3>(ctx, cur, nxt, word, shift, n, g, d_table, d_n); break;
        }
        SortBufs tmp = cur; cur = nxt; nxt = tmp;
        in_b = !in_b;
    }
    PTX_HIP(ctx, hipGetLastError());
    *result_in_b = in_b;
    return 0;
}

}  // namespace ptx
