// wave.hpp -- wave64 cross-lane reductions for gfx950 (device code only)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ptx {

// Wave64 reductions on the VALU cross-lane path (DPP row operations + readlane): a few dozen cycles, where a
// __shfl_down ladder (ds_bpermute, two per step for a double) costs over a thousand -- the solver's line search is
// a serial chain of such reductions.  All 64 lanes must be active.  The combining order is fixed (deterministic).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
template <int CTRL, class T>
__device__ __forceinline__ T dpp(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit lanes");
    if constexpr (sizeof(T) == 4) {
        uint32_t u; __builtin_memcpy(&u, &v, 4); u = dpp32<CTRL>(u);
        T r; __builtin_memcpy(&r, &u, 4); return r;
    } else {
        uint64_t u; __builtin_memcpy(&u, &v, 8);
        uint32_t lo = dpp32<CTRL>((uint32_t)u), hi = dpp32<CTRL>((uint32_t)(u >> 32));
        u = ((uint64_t)hi << 32) | lo;
        T r; __builtin_memcpy(&r, &u, 8); return r;
    }
}
template <class T>
__device__ __forceinline__ T lane_get(T v, int lane) {
    if constexpr (sizeof(T) == 4) {
        uint32_t u; __builtin_memcpy(&u, &v, 4); u = (uint32_t)__builtin_amdgcn_readlane((int)u, lane);
        T r; __builtin_memcpy(&r, &u, 4); return r;
    } else {
        uint64_t u; __builtin_memcpy(&u, &v, 8);
        uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
        u = ((uint64_t)hi << 32) | lo;
        T r; __builtin_memcpy(&r, &u, 8); return r;
    }
}
// quad swap, quad-pair swap, row rotate by 4 and 8: every lane of a 16-lane row holds the row's result; the four
// row results are combined from scalar registers, so every lane returns the wave's result
template <class T, class Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
    v = op(v, dpp<0xB1>(v)); v = op(v, dpp<0x4E>(v)); v = op(v, dpp<0x124>(v)); v = op(v, dpp<0x128>(v));
    return op(op(lane_get(v, 0), lane_get(v, 16)), op(lane_get(v, 32), lane_get(v, 48)));
}
// the first four steps alone: every lane holds the result of its own 16-lane row
template <class T, class Op>
__device__ __forceinline__ T row_reduce(T v, Op op) {
    v = op(v, dpp<0xB1>(v)); v = op(v, dpp<0x4E>(v)); v = op(v, dpp<0x124>(v)); v = op(v, dpp<0x128>(v));
    return v;
}
// lexicographic pair reductions: (a, b) "better" as decided by `better(a2, b2, a, b)`
template <class A, class B, class Better>
__device__ __forceinline__ void wave_reduce_pair(A &a, B &b, Better better) {
#define PTX_PAIR_STEP(CTRL) { A a2 = dpp<CTRL>(a); B b2 = dpp<CTRL>(b); if (better(a2, b2, a, b)) { a = a2; b = b2; } }
    PTX_PAIR_STEP(0xB1) PTX_PAIR_STEP(0x4E) PTX_PAIR_STEP(0x124) PTX_PAIR_STEP(0x128)
#undef PTX_PAIR_STEP
    A ra = lane_get(a, 0); B rb = lane_get(b, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) { A a2 = lane_get(a, r); B b2 = lane_get(b, r); if (better(a2, b2, ra, rb)) { ra = a2; rb = b2; } }
    a = ra; b = rb;
}

// whole-wave shift towards higher lanes by one: lane i receives lane i-1's value, lane 0 `fill` (DPP wave_shr:1 -- a VALU
// move, where __shfl_up is a round trip through the LDS crossbar)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill = 0u) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}
// the same with a zero for lane 0 through bound_ctrl: no `old` operand, so a chain of shifts needs no register copies
__device__ __forceinline__ uint32_t wave_shr1z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); }
// inclusive prefix sum over the 64 lanes on the DPP path: row_shr 1/2/4/8 inside the 16-lane rows (zero fill), then the
// row totals travel by row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3).  Wraps modulo 2^32 like the adds do.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a block of NT threads (NT multiple of 64, <= 1024)
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave /*[NT/64]*/, uint32_t *block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        uint32_t t = s_wave[w];
        if (w < wave) woff += t;
        tot += t;
    }
    __syncthreads();
    *block_total = tot;
    return woff + incl - v;
}


}  // namespace ptx
