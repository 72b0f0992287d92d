// scan_chained.hpp -- the single-pass chained scan as a template over a load and a store functor, so that producers of
// the scanned values and consumers of the prefixes fuse into the one launch (device code + its host launcher).
#pragma once
#include <algorithm>
#include <cstdlib>
#include "common.hpp"
#include "wave.hpp"

namespace ptx {

// two tile shapes: 256 threads x 8 items for small inputs (many workgroups of little work), 512 x 16 above SCAN_BIG_N items --
// every tile costs one ticket (a same-address atomic, ~12-25 ns each, served one after the other) and one hop of the
// look-back chain, so on tens of millions of items 2048-item tiles make the scan ticket-bound (0.40 ms for 3.2e7 items,
// 0.65 TB/s) where 8192-item tiles stream
constexpr uint64_t SCAN_BIG_N = 1u << 22, SCAN_HUGE_N = 1u << 26;
constexpr int SCAN_TILE_SMALL = 256 * 8, SCAN_TILE_BIG = 512 * 16, SCAN_TILE_HUGE = 1024 * 16;   // 16384-item tiles from 2^26 items on (round 3: 3.2e8-item scans at cfg4)

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
// (SCAN_ITEMS consecutive items per thread) is reached through two padded LDS transposes of one buffer; the loaded
// values stay in registers for the store.
template <int SCAN_BLOCK, int SCAN_ITEMS, class Load, class Store>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_chained_kernel(Load load, Store store, uint64_t n, uint32_t *__restrict__ ws, uint32_t epoch,
                                                                  uint32_t *__restrict__ total) {
    constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;
    __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
    __shared__ uint32_t s_tile, s_excl;
    __shared__ uint32_t s_v[SCAN_BLOCK * (SCAN_ITEMS + 1)];   // one pad word per thread row: no bank conflicts
    uint32_t *ticket = ws;
    uint64_t *state = reinterpret_cast<uint64_t *>(ws + 2);
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, nb = gridDim.x;
    const int lane = threadIdx.x & 63;
    const uint64_t tile_base = (uint64_t)tile * SCAN_TILE;
    uint32_t raw[SCAN_ITEMS], v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {      // striped: item j = k * SCAN_BLOCK + thread
        const uint64_t idx = tile_base + (uint32_t)(k * SCAN_BLOCK) + threadIdx.x;
        raw[k] = idx < n ? load(idx) : 0u;
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t j = k * SCAN_BLOCK + threadIdx.x;
        s_v[j + j / SCAN_ITEMS] = raw[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {      // blocked: thread t owns items ITEMS * t .. ITEMS * t + ITEMS - 1
        v[i] = s_v[threadIdx.x * (SCAN_ITEMS + 1) + i];
        s += v[i];
    }
    uint32_t tot;
    uint32_t off = block_excl_scan<SCAN_BLOCK>(s, s_wave, &tot);   // its barriers also end every thread's reads of s_v
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
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {      // prefixes relative to the tile, blocked, into the same buffer
        s_v[threadIdx.x * (SCAN_ITEMS + 1) + i] = off;
        off += v[i];
    }
    __syncthreads();
    const uint32_t excl = s_excl;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t j = k * SCAN_BLOCK + threadIdx.x;
        const uint64_t idx = tile_base + j;
        if (idx < n) store(idx, excl + s_v[j + j / SCAN_ITEMS], raw[k]);
    }
    if (tile == nb - 1 && threadIdx.x == 0) {
        if (total) *total = excl + tot;
        *ticket = 0;   // every ticket of this launch has been taken
    }
}

template <class Load, class Store>
int exclusive_scan_fn(Ctx *ctx, Load load, Store store, uint64_t n, uint32_t *d_total, const char *timer_name) {
    if (n == 0) {
        if (d_total) PTX_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
        return 0;
    }
    const bool big = n >= SCAN_BIG_N, huge = n >= SCAN_HUGE_N && !ctx->cfg.scan_no_huge;
    const uint32_t tile_items = huge ? SCAN_TILE_HUGE : big ? SCAN_TILE_BIG : SCAN_TILE_SMALL;
    const uint32_t nb = (uint32_t)((n + tile_items - 1) / tile_items);
    const size_t need = 2 + 2 * (size_t)nb;   // u32 words: ticket, pad, one u64 per tile
    // one workspace per stream: launches of one stream follow each other, a scan of the side stream may run beside one of the main stream
    const bool side = ctx->stream == ctx->stream2;
    DevBuf<uint32_t> &ws = side ? ctx->d_scan_ws2 : ctx->d_scan_ws;
    uint32_t &epoch = side ? ctx->scan_epoch2 : ctx->scan_epoch;
    if (ws.n < need) {
        PTX_HIP(ctx, ws.alloc(std::max<size_t>(need, 1u << 16)));
        PTX_HIP(ctx, hipMemsetAsync(ws.p, 0, ws.bytes(), ctx->stream));
        epoch = 0;
    }
    if (++epoch >= (1u << 30)) {   // epoch field wrapped: start over with a clean workspace
        PTX_HIP(ctx, hipMemsetAsync(ws.p, 0, ws.bytes(), ctx->stream));
        epoch = 1;
    }
    KTimer t(ctx, timer_name);
    if (huge) hipLaunchKernelGGL((scan_chained_kernel<1024, 16, Load, Store>), dim3(nb), dim3(1024), 0, ctx->stream, load, store, n, ws.p, epoch, d_total);
    else if (big) hipLaunchKernelGGL((scan_chained_kernel<512, 16, Load, Store>), dim3(nb), dim3(512), 0, ctx->stream, load, store, n, ws.p, epoch, d_total);
    else hipLaunchKernelGGL((scan_chained_kernel<256, 8, Load, Store>), dim3(nb), dim3(256), 0, ctx->stream, load, store, n, ws.p, epoch, d_total);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx
