// scan_chained.hpp -- the single-pass chained scan as a template over a load and a store functor, so that producers of
// the scanned values and consumers of the prefixes fuse into the one launch (device code + its host launcher).
#pragma once
#include <algorithm>
#include "common.hpp"
#include "wave.hpp"

namespace ptx {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// Single-pass chained scan (decoupled look-back): ONE launch per scan.  A workgroup takes a ticket (tiles are
// therefore started in order, so the tiles it waits for are already running), scans its tile of SCAN_TILE items,
// publishes {epoch, flag, value} of the tile in one 64-bit word -- first its aggregate, later its inclusive
// prefix -- and wave 0 looks back over the predecessors 64 tiles at a time until it meets a published prefix.
// The epoch (one per scan call) makes words left by earlier scans read as "not ready", so the workspace is never
// cleared; the last ticket holder resets the ticket counter.  State words are agent-scope atomics: they are
// served below the per-XCD L2s, which is what makes the hand-off visible across XCDs.
constexpr uint64_t ST_AGG = 1, ST_PREFIX = 2;
__device__ __forceinline__ uint64_t st_pack(uint32_t epoch, uint64_t flag, uint32_t v) { return ((uint64_t)epoch << 34) | (flag << 32) | v; }

// load(i) -> value of item i (i < n); store(i, exclusive prefix, value) consumes it.  Loads of a tile happen before
// its stores, so in-place scans are fine.  Both functors are called in STRIPED order (consecutive lanes = consecutive
// items): whatever they touch in memory is coalesced, also 16-byte records; the blocked order the scan itself wants
// (8 consecutive items per thread) is reached through two padded LDS transposes.
template <class Load, class Store>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_chained_kernel(Load load, Store store, uint64_t n, uint32_t *__restrict__ ws, uint32_t epoch,
                                                                  uint32_t *__restrict__ total) {
    __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
    __shared__ uint32_t s_tile, s_excl;
    __shared__ uint32_t s_v[SCAN_BLOCK * (SCAN_ITEMS + 1)], s_p[SCAN_BLOCK * (SCAN_ITEMS + 1)];   // one pad word per thread row: no bank conflicts
    uint32_t *ticket = ws;
    uint64_t *state = reinterpret_cast<uint64_t *>(ws + 2);
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, nb = gridDim.x;
    const int lane = threadIdx.x & 63;
    const uint64_t tile_base = (uint64_t)tile * SCAN_TILE;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {      // striped: item j = k * 256 + thread
        const uint32_t j = k * SCAN_BLOCK + threadIdx.x;
        const uint64_t idx = tile_base + j;
        s_v[j + j / SCAN_ITEMS] = idx < n ? load(idx) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {      // blocked: thread t owns items 8t .. 8t+7
        v[i] = s_v[threadIdx.x * (SCAN_ITEMS + 1) + i];
        s += v[i];
    }
    uint32_t tot;
    uint32_t off = block_excl_scan<SCAN_BLOCK>(s, s_wave, &tot);
    if (threadIdx.x < 64) {
        uint32_t excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&state[0], st_pack(epoch, ST_PREFIX, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&state[tile], st_pack(epoch, ST_AGG, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int look = (int)tile - 1;
            while (true) {
                const int idx = look - lane;
                uint64_t st;
                bool ready;
                do {
                    st = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : st_pack(epoch, ST_PREFIX, 0);
                    ready = (uint32_t)(st >> 34) == epoch && ((st >> 32) & 3) != 0;
                } while (!__all(ready));
                const unsigned long long pm = __ballot(((st >> 32) & 3) == ST_PREFIX);
                const int first = pm ? __ffsll((long long)pm) - 1 : 64;
                excl += wave_reduce(lane <= first ? (uint32_t)st : 0u, [](uint32_t x, uint32_t y) { return x + y; });
                if (first < 64) break;
                look -= 64;
            }
            if (lane == 0) __hip_atomic_store(&state[tile], st_pack(epoch, ST_PREFIX, excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    off += s_excl;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        s_p[threadIdx.x * (SCAN_ITEMS + 1) + i] = off;
        off += v[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t j = k * SCAN_BLOCK + threadIdx.x;
        const uint64_t idx = tile_base + j;
        if (idx < n) store(idx, s_p[j + j / SCAN_ITEMS], s_v[j + j / SCAN_ITEMS]);
    }
    if (tile == nb - 1 && threadIdx.x == 0) {
        if (total) *total = s_excl + tot;
        *ticket = 0;   // every ticket of this launch has been taken
    }
}


template <class Load, class Store>
int exclusive_scan_fn(Ctx *ctx, Load load, Store store, uint64_t n, uint32_t *d_total, const char *timer_name) {
    if (n == 0) {
        if (d_total) PTX_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
        return 0;
    }
    const uint32_t nb = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
    const size_t need = 2 + 2 * (size_t)nb;   // u32 words: ticket, pad, one u64 per tile
    if (ctx->d_scan_ws.n < need) {
        PTX_HIP(ctx, ctx->d_scan_ws.alloc(std::max<size_t>(need, 1u << 16)));
        PTX_HIP(ctx, hipMemsetAsync(ctx->d_scan_ws.p, 0, ctx->d_scan_ws.bytes(), ctx->stream));
        ctx->scan_epoch = 0;
    }
    if (++ctx->scan_epoch >= (1u << 30)) {   // epoch field wrapped: start over with a clean workspace
        PTX_HIP(ctx, hipMemsetAsync(ctx->d_scan_ws.p, 0, ctx->d_scan_ws.bytes(), ctx->stream));
        ctx->scan_epoch = 1;
    }
    KTimer t(ctx, timer_name);
    hipLaunchKernelGGL((scan_chained_kernel<Load, Store>), dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, load, store, n, ctx->d_scan_ws.p, ctx->scan_epoch,
                       d_total);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx
