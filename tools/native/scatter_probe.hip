// scatter_probe.hip -- measurement only (tools/ssn_scatter_probe.sh): what does a bucket scatter of 16-byte rows cost on MI355X as a
// function of the number of buckets a workgroup feeds, written row by row (one 16-byte store per row at the bucket's cursor) or staged
// (a tile's rows ordered by bucket in LDS first, then written run by run by neighbouring lanes)?  Geometry of the LP row sort at cfg4:
// 1000 segments x 200 000 rows, 8 workgroups per segment, bucket ids pseudo-random.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITEMS = 8, TILE = 256 * ITEMS;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// direct: every row one store
__global__ void __launch_bounds__(256) direct_kernel(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, uint32_t n_seg, uint32_t G, uint32_t B, uint32_t cap) {
    extern __shared__ uint32_t s_cnt[];
    const uint32_t s = blockIdx.y, g = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < B; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const uint32_t per = (n_seg + G - 1) / G, r0 = g * per, r1 = min(n_seg, r0 + per);
    const size_t so = (size_t)s * n_seg;
    ulonglong2 *ob = out + (size_t)s * B * G * cap;
    for (uint32_t base = r0; base < r1; base += TILE) {
        ulonglong2 row[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) { const uint32_t i = base + r * 256 + threadIdx.x; row[r] = i < r1 ? in[so + i] : make_ulonglong2(0, 0); }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = base + r * 256 + threadIdx.x;
            if (i >= r1) continue;
            const uint32_t id = mix(i * 2654435761u + s) % B;
            const uint32_t pos = atomicAdd(&s_cnt[id], 1u);
            if (pos < cap) ob[((size_t)id * G + g) * cap + pos] = row[r];
        }
    }
}
// staged: the tile ordered by bucket in LDS, then runs
__global__ void __launch_bounds__(256) staged_kernel(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, uint32_t n_seg, uint32_t G, uint32_t B, uint32_t cap) {
    extern __shared__ uint32_t smem[];
    uint32_t *s_cnt = smem, *s_loc = smem + B, *s_start = smem + 2 * B;       // bucket cursors of the WG; tile-local counts / starts
    ulonglong2 *s_rows = reinterpret_cast<ulonglong2 *>(smem + 3 * B + (B & 1 ? 1 : 0) + 2);   // [TILE] (16-byte aligned below)
    s_rows = reinterpret_cast<ulonglong2 *>(((uintptr_t)s_rows + 15) & ~(uintptr_t)15);
    uint16_t *s_id = reinterpret_cast<uint16_t *>(s_rows + TILE);            // [TILE]
    __shared__ uint32_t s_wave[4];
    const uint32_t s = blockIdx.y, g = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < B; i += 256) s_cnt[i] = 0;
    const uint32_t per = (n_seg + G - 1) / G, r0 = g * per, r1 = min(n_seg, r0 + per);
    const size_t so = (size_t)s * n_seg;
    ulonglong2 *ob = out + (size_t)s * B * G * cap;
    for (uint32_t base = r0; base < r1; base += TILE) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < B; i += 256) s_loc[i] = 0;
        __syncthreads();
        ulonglong2 row[ITEMS];
        uint32_t id[ITEMS], rk[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) { const uint32_t i = base + r * 256 + threadIdx.x; row[r] = i < r1 ? in[so + i] : make_ulonglong2(0, 0); }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = base + r * 256 + threadIdx.x;
            id[r] = 0xFFFFFFFFu;
            if (i < r1) { id[r] = mix(i * 2654435761u + s) % B; rk[r] = atomicAdd(&s_loc[id[r]], 1u); }
        }
        __syncthreads();
        {   // exclusive scan of the B local counts (B <= 2048: 8 per thread)
            const uint32_t per_t = (B + 255) / 256, b0 = threadIdx.x * per_t;
            uint32_t sum = 0;
            for (uint32_t i = 0; i < per_t; ++i) if (b0 + i < B) sum += s_loc[b0 + i];
            uint32_t incl = sum;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            uint32_t off = incl - sum;
            for (int w = 0; w < wave; ++w) off += s_wave[w];
            for (uint32_t i = 0; i < per_t; ++i) if (b0 + i < B) { s_start[b0 + i] = off; off += s_loc[b0 + i]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            if (id[r] != 0xFFFFFFFFu) { const uint32_t p = s_start[id[r]] + rk[r]; s_rows[p] = row[r]; s_id[p] = (uint16_t)id[r]; }
        __syncthreads();
        const uint32_t nrow = min((uint32_t)TILE, r1 - base);
        for (uint32_t p = threadIdx.x; p < nrow; p += 256) {
            const uint32_t b = s_id[p], pos = s_cnt[b] + (p - s_start[b]);
            if (pos < cap) ob[((size_t)b * G + g) * cap + pos] = s_rows[p];
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < B; i += 256) s_cnt[i] += s_loc[i];
    }
}
int main(int argc, char **argv) {
    const uint32_t S = 1000, n_seg = 200000, G = 8;
    const size_t n = (size_t)S * n_seg;
    ulonglong2 *in, *out;
    CK(hipMalloc(&in, n * 16));
    CK(hipMemset(in, 1, n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t B : {32u, 64u, 128u, 256u, 512u, 1024u, 2048u}) {
        const uint32_t per = (n_seg + G - 1) / G, cap = (uint32_t)(per / B * 1.3) + 24;
        const size_t out_rows = (size_t)S * B * G * cap;
        CK(hipMalloc(&out, out_rows * 16));
        for (int staged = 0; staged < 2; ++staged) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (!staged) hipLaunchKernelGGL(direct_kernel, dim3(G, S), dim3(256), B * 4, 0, in, out, n_seg, G, B, cap);
                else hipLaunchKernelGGL(staged_kernel, dim3(G, S), dim3(256), 3 * B * 4 + 64 + TILE * 16 + TILE * 2, 0, in, out, n_seg, G, B, cap);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("B=%4u %s  %.3f ms  (%.2f TB/s of 2 x 16 B x %zu rows)\n", B, staged ? "staged" : "direct", best, 2.0 * 16 * n / best / 1e9, n);
        }
        CK(hipFree(out));
    }
    return 0;
}
