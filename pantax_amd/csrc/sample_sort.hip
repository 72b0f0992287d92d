// sample_sort.hip -- sort of the LP rows (three u64 key words, no payload) when they are few (<= SS_MAX_N).
//
// An LSD radix sort of 63+ key bits is 10 passes x 3 dependent launches; at a few 1e5 rows every launch is
// ~5 us of dispatch latency and the data would fit the LDS of the chip many times over.  Sample sort needs
// five launches in all:
//   1. ss_sample_kernel  : 4096 evenly spaced rows are ranked against each other (256 workgroups x 16 samples,
//                        all-pairs compares on an LDS copy of the sample); every 4th one in rank order is a
//                        splitter (1023 splitters).  Inputs of <= 4096 rows are ranked completely right here.
//   2. ss_hist_kernel    : bucket of every row by binary search over the splitters (LDS); bucket ids
//                        2j   = strictly between splitter j-1 and j        (needs sorting)
//                        2j+1 = equal to splitter j                         (all keys identical: no sorting;
//                        coverage values tie massively, so heavy keys become splitters and land here)
//                        per-workgroup bucket counts -> table[bucket][workgroup]
//   3. ss_rowscan_kernel : prefix over the workgroups for every bucket + bucket totals
//   4. ss_scatter_kernel : rows to their bucket (order inside a bucket is arbitrary)
//   5. ss_local_kernel<1024> (one workgroup per bucket): bitonic sort of the bucket in LDS, written back in place;
//      ss_local_kernel<4096> afterwards for the few buckets above 1024 rows (24 KB of LDS for the common case
//      keeps six workgroups per CU in flight).
// With 4x oversampling a "between" bucket holds n/1024 rows on average; the LDS path takes 4096, i.e. >= 7x
// the mean at SS_MAX_N -- exceeding it has probability < 1e-6 per sort, and a bucket that does is still
// sorted correctly (rank sort through memory, slow).  Keys are compared as (k0, k1, k2) tuples.
#include <algorithm>
#include "primitives.hpp"
#include "wave.hpp"

namespace ptx {

constexpr int SS_SAMPLE = 4096;
constexpr int SS_NSPLIT = SS_SAMPLE / 4 - 1;   // 1023
constexpr int SS_NBUCKET = 2 * (SS_NSPLIT + 1);   // 2048 ids (the last odd one stays empty)
constexpr int SS_TILE = 512;                   // rows per workgroup of the partition kernels
constexpr int SS_CAP = 4096;                   // rows a bucket may hold to be sorted in LDS (second local kernel)
constexpr int SS_CAP1 = 1024;                  // ... by the first local kernel

struct Key3 { uint64_t a, b, c; };
// branch-free on purpose (bitwise & |): these sit in loops whose loads should be issued back to back
__device__ __forceinline__ bool less3(const Key3 &x, const Key3 &y) {
    return (x.a < y.a) | ((x.a == y.a) & ((x.b < y.b) | ((x.b == y.b) & (x.c < y.c))));
}
__device__ __forceinline__ bool eq3(const Key3 &x, const Key3 &y) { return (x.a == y.a) & (x.b == y.b) & (x.c == y.c); }

// bitonic sort of N (power of two) keys held in three LDS arrays, NT threads, ascending
template <int NT>
__device__ __forceinline__ void bitonic_lds(uint64_t *ka, uint64_t *kb, uint64_t *kc, uint32_t N) {
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const Key3 x{ka[i], kb[i], kc[i]}, y{ka[l], kb[l], kc[l]};
                const bool up = (i & k) == 0;
                if (up ? less3(y, x) : less3(x, y)) {
                    ka[i] = y.a; kb[i] = y.b; kc[i] = y.c;
                    ka[l] = x.a; kb[l] = x.b; kc[l] = x.c;
                }
            }
            __syncthreads();
        }
    }
}

// the same network on one key word (buckets whose rows share the two leading words)
template <int NT>
__device__ __forceinline__ void bitonic_lds1(uint64_t *kc, uint32_t N) {
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint64_t x = kc[i], y = kc[l];
                const bool up = (i & k) == 0;
                if (up ? (y < x) : (x < y)) { kc[i] = y; kc[l] = x; }
            }
            __syncthreads();
        }
    }
}

// workspace (u32 words)
struct SsWs {
    uint32_t *flags;          // [0] = 1: the whole input was sorted by the sample kernel
    uint64_t *spl;            // [3][1024]
    uint64_t *samp;           // [3][SS_SAMPLE] the sampled rows, contiguous
    uint32_t *bucket_start;   // [SS_NBUCKET + 1]
    uint32_t *table;          // [nb][SS_NBUCKET] (workgroup-major: coalesced on both sides) + [SS_NBUCKET] totals
    uint16_t *ids;            // [n_bound]
    uint32_t *big_list;       // [SS_NBUCKET] buckets above SS_CAP1 rows
};
static SsWs ss_layout(uint32_t *ws, uint32_t nb) {
    SsWs w;
    w.flags = ws;
    w.spl = reinterpret_cast<uint64_t *>(ws + 4);
    w.samp = w.spl + 3 * 1024;
    w.bucket_start = ws + 4 + 2 * 3 * 1024 + 2 * 3 * SS_SAMPLE;
    w.big_list = w.bucket_start + (SS_NBUCKET + 4);
    w.table = w.big_list + SS_NBUCKET;
    w.ids = reinterpret_cast<uint16_t *>(w.table + (size_t)SS_NBUCKET * (nb + 1));
    return w;
}
size_t sample_sort_ws_elems(uint64_t n_bound) {
    const uint64_t nb = (n_bound + SS_TILE - 1) / SS_TILE + 1;
    return 4 + 2 * 3 * 1024 + 2 * 3 * SS_SAMPLE + (SS_NBUCKET + 4) + SS_NBUCKET + (size_t)SS_NBUCKET * (nb + 1) + (n_bound + 1) / 2 + 8;
}

// 256 workgroups x 16 samples, 16 lanes per sample: every lane counts the samples that precede its own among a
// 1/16 interleaved slice of all 4096 (LDS reads of a 16-lane row are consecutive, the four rows of a wave
// read the same 16 elements: no bank conflicts), a 16-lane DPP reduction gives the rank.
__global__ void __launch_bounds__(256) ss_gather_kernel(SortBufs a, const uint32_t *__restrict__ d_n, uint64_t *__restrict__ samp, uint32_t *__restrict__ flags) {
    const uint32_t n = *d_n;
    const bool small = n <= (uint32_t)SS_SAMPLE;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;   // grid covers SS_SAMPLE
    const uint64_t pos = small ? i : ((uint64_t)i * n) / SS_SAMPLE;
    const bool ok = pos < n;
    samp[i] = ok ? a.k[0][pos] : ~0ull; samp[SS_SAMPLE + i] = ok ? a.k[1][pos] : ~0ull; samp[2 * SS_SAMPLE + i] = ok ? a.k[2][pos] : ~0ull;
    if (i == 0) flags[1] = 0;   // number of buckets left to the large local kernel
}
__global__ void __launch_bounds__(256) ss_sample_kernel(SortBufs b, const uint32_t *__restrict__ d_n, const uint64_t *__restrict__ samp,
                                                        uint32_t *__restrict__ flags, uint64_t *__restrict__ spl) {
    __shared__ uint64_t ka[SS_SAMPLE], kb[SS_SAMPLE], kc[SS_SAMPLE];
    const uint32_t n = *d_n;
    const bool small = n <= (uint32_t)SS_SAMPLE;
    for (uint32_t i = threadIdx.x; i < (uint32_t)SS_SAMPLE; i += 256) { ka[i] = samp[i]; kb[i] = samp[SS_SAMPLE + i]; kc[i] = samp[2 * SS_SAMPLE + i]; }
    __syncthreads();
    const uint32_t s_idx = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
    const Key3 me{ka[s_idx], kb[s_idx], kc[s_idx]};
    uint32_t cnt = 0;
#pragma unroll 8
    for (uint32_t it = 0; it < (uint32_t)SS_SAMPLE / 16; ++it) {
        const uint32_t j = it * 16 + part;
        const Key3 o{ka[j], kb[j], kc[j]};
        cnt += ((int)less3(o, me) | ((int)eq3(o, me) & (int)(j < s_idx))) ? 1u : 0u;
    }
    // sum over the 16 lanes of the row
    cnt += dpp<0xB1>(cnt); cnt += dpp<0x4E>(cnt); cnt += dpp<0x124>(cnt); cnt += dpp<0x128>(cnt);
    if (part == 0) {
        if (small) {
            if (s_idx < n) { b.k[0][cnt] = me.a; b.k[1][cnt] = me.b; b.k[2][cnt] = me.c; }   // copied back by the local kernel
        } else if ((cnt & 3u) == 3u && (cnt >> 2) < (uint32_t)SS_NSPLIT) {
            spl[cnt >> 2] = me.a; spl[1024 + (cnt >> 2)] = me.b; spl[2048 + (cnt >> 2)] = me.c;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = small ? 1u : 0u;
}

__global__ void __launch_bounds__(256) ss_hist_kernel(SortBufs a, const uint32_t *__restrict__ d_n, const uint32_t *__restrict__ flags,
                                                      const uint64_t *__restrict__ spl, uint32_t nb, uint32_t *__restrict__ table,
                                                      uint16_t *__restrict__ ids) {
    __shared__ uint64_t sa[1024], sb[1024], sc[1024];
    __shared__ uint32_t s_hist[SS_NBUCKET];
    for (int i = threadIdx.x; i < SS_NBUCKET; i += 256) s_hist[i] = 0;
    const bool skip = flags[0] != 0;
    const uint32_t n = skip ? 0u : *d_n;
    for (int i = threadIdx.x; i < SS_NSPLIT; i += 256) { sa[i] = spl[i]; sb[i] = spl[1024 + i]; sc[i] = spl[2048 + i]; }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * SS_TILE;
    for (int r = 0; r < SS_TILE / 256; ++r) {
        const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        if (i >= n) continue;
        const Key3 key{a.k[0][i], a.k[1][i], a.k[2][i]};
        uint32_t lo = 0, hi = SS_NSPLIT;   // first splitter that is not less than the key
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (less3(Key3{sa[mid], sb[mid], sc[mid]}, key)) lo = mid + 1; else hi = mid;
        }
        const uint32_t bid = (lo < (uint32_t)SS_NSPLIT && eq3(Key3{sa[lo], sb[lo], sc[lo]}, key)) ? 2 * lo + 1 : 2 * lo;
        ids[i] = (uint16_t)bid;
        atomicAdd(&s_hist[bid], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_NBUCKET; i += 256) table[(size_t)blockIdx.x * SS_NBUCKET + i] = s_hist[i];
}

// prefix over the workgroups for every bucket, in place, + bucket totals.  A workgroup owns 8 buckets; thread
// (bucket, slice) walks a contiguous slice of the workgroups (8 threads read one 32-byte sector per row).
__global__ void __launch_bounds__(256) ss_rowscan_kernel(uint32_t *__restrict__ table, uint32_t nb) {
    __shared__ uint32_t s_tot[32][8];
    const uint32_t bl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const uint32_t bucket = blockIdx.x * 8 + bl;
    const uint32_t per = (nb + 31) / 32;
    const uint32_t lo = sl * per, hi = (lo + per < nb) ? lo + per : nb;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += table[(size_t)i * SS_NBUCKET + bucket];
    s_tot[sl][bl] = s;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int l = 0; l < 32; ++l) { const uint32_t t = s_tot[l][bl]; if ((uint32_t)l < sl) pre += t; tot += t; }
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = table[(size_t)i * SS_NBUCKET + bucket]; table[(size_t)i * SS_NBUCKET + bucket] = pre; pre += v; }
    if (sl == 0) table[(size_t)nb * SS_NBUCKET + bucket] = tot;
}

__global__ void __launch_bounds__(256) ss_scatter_kernel(SortBufs a, SortBufs b, const uint32_t *__restrict__ d_n, const uint32_t *__restrict__ flags,
                                                         uint32_t nb, const uint32_t *__restrict__ table, const uint16_t *__restrict__ ids,
                                                         uint32_t *__restrict__ bucket_start) {
    __shared__ uint32_t s_base[SS_NBUCKET], s_cnt[SS_NBUCKET];
    __shared__ uint32_t s_wave[4];
    if (flags[0] != 0) return;
    const uint32_t n = *d_n;
    {   // exclusive scan of the bucket totals (8 consecutive buckets per thread) + this workgroup's prefix inside each bucket
        const uint32_t *tot = table + (size_t)SS_NBUCKET * nb;
        const uint32_t b0 = threadIdx.x * 8;
        uint32_t v[8], s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = tot[b0 + i]; s += v[i]; }
        uint32_t total;
        uint32_t off = block_excl_scan<256>(s, s_wave, &total);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (blockIdx.x == 0) bucket_start[b0 + i] = off;
            s_base[b0 + i] = off + table[(size_t)blockIdx.x * SS_NBUCKET + b0 + i];
            s_cnt[b0 + i] = 0;
            off += v[i];
        }
        if (blockIdx.x == 0 && threadIdx.x == 255) bucket_start[SS_NBUCKET] = off;
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * SS_TILE;
    for (int r = 0; r < SS_TILE / 256; ++r) {
        const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t bid = ids[i];
        const uint32_t pos = s_base[bid] + atomicAdd(&s_cnt[bid], 1u);
        b.k[0][pos] = a.k[0][i]; b.k[1][pos] = a.k[1][i]; b.k[2][pos] = a.k[2][i];
    }
}

// CAP = SS_CAP1: buckets of 2..SS_CAP1 rows (+ the copies of single-key buckets, + the small-input copy);
// CAP = SS_CAP : the rest.
template <int CAP>
__global__ void __launch_bounds__(256) ss_local_kernel(SortBufs a, SortBufs b, const uint32_t *__restrict__ d_n, uint32_t *__restrict__ flags,
                                                       const uint32_t *__restrict__ bucket_start, uint32_t *__restrict__ big_list,
                                                       const uint64_t *__restrict__ spl) {
    __shared__ uint64_t ka[CAP], kb[CAP], kc[CAP];
    if (flags[0] != 0) {
        const uint32_t bid = blockIdx.x;   // small input: the sample kernel ranked every row into b
        if (CAP == SS_CAP1) {
            const uint32_t n = *d_n;
            for (uint32_t i = bid * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { a.k[0][i] = b.k[0][i]; a.k[1][i] = b.k[1][i]; a.k[2][i] = b.k[2][i]; }
        }
        return;
    }
    // the small instantiation sees every bucket and lists the ones above its capacity in big_list (flags[1] of them)
    const uint32_t n_work = CAP == SS_CAP1 ? (uint32_t)SS_NBUCKET : flags[1];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t bid = CAP == SS_CAP1 ? wi : big_list[wi];
    const uint32_t st = bucket_start[bid], en = bucket_start[bid + 1];
    const uint32_t m = en - st;
    if (m == 0) continue;
    if ((bid & 1u) || m == 1) {   // identical keys (or a single row): nothing to sort
        if (CAP == SS_CAP1)
            for (uint32_t i = threadIdx.x; i < m; i += 256) { a.k[0][st + i] = b.k[0][st + i]; a.k[1][st + i] = b.k[1][st + i]; a.k[2][st + i] = b.k[2][st + i]; }
        continue;
    }
    if (CAP == SS_CAP1 && m > (uint32_t)SS_CAP1) {   // left to the large instantiation
        if (threadIdx.x == 0) big_list[atomicAdd(&flags[1], 1u)] = bid;
        continue;
    }
    __syncthreads();   // LDS reuse across the buckets of this workgroup
    if (m <= (uint32_t)CAP) {
        uint32_t N = 2;
        while (N < m) N <<= 1;
        // a bucket between two splitters that agree in the two leading words holds rows that agree in them too
        // (the usual case: one species, one mask): only the third word moves through the network
        const uint32_t sj = bid >> 1;
        if (sj > 0 && sj < (uint32_t)SS_NSPLIT && spl[sj - 1] == spl[sj] && spl[1024 + sj - 1] == spl[1024 + sj]) {
            const uint64_t k0v = spl[sj], k1v = spl[1024 + sj];
            for (uint32_t i = threadIdx.x; i < N; i += 256) kc[i] = i < m ? b.k[2][st + i] : ~0ull;
            __syncthreads();
            bitonic_lds1<256>(kc, N);
            for (uint32_t i = threadIdx.x; i < m; i += 256) { a.k[0][st + i] = k0v; a.k[1][st + i] = k1v; a.k[2][st + i] = kc[i]; }
            continue;
        }
        for (uint32_t i = threadIdx.x; i < N; i += 256) {
            if (i < m) { ka[i] = b.k[0][st + i]; kb[i] = b.k[1][st + i]; kc[i] = b.k[2][st + i]; }
            else { ka[i] = ~0ull; kb[i] = ~0ull; kc[i] = ~0ull; }
        }
        __syncthreads();
        bitonic_lds<256>(ka, kb, kc, N);
        for (uint32_t i = threadIdx.x; i < m; i += 256) { a.k[0][st + i] = ka[i]; a.k[1][st + i] = kb[i]; a.k[2][st + i] = kc[i]; }
        continue;
    }
    // oversized bucket (practically never): rank every row against the whole bucket through memory
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
        const Key3 key{b.k[0][st + i], b.k[1][st + i], b.k[2][st + i]};
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) {
            const Key3 o{b.k[0][st + j], b.k[1][st + j], b.k[2][st + j]};
            if (less3(o, key) || (eq3(o, key) && j < i)) ++rank;
        }
        a.k[0][st + rank] = key.a; a.k[1][st + rank] = key.b; a.k[2][st + rank] = key.c;
    }
    }   // buckets of this workgroup
}

int sample_sort3(Ctx *ctx, SortBufs a, SortBufs b, uint64_t n_bound, uint32_t *d_ws, const uint32_t *d_n) {
    if (a.nw != 3 || a.v) return fail(ctx, PANTAX_HIP_E_INVALID, "sample_sort3: three key words, no payload");
    if (n_bound == 0) return 0;
    if (n_bound > SS_MAX_N) return fail(ctx, PANTAX_HIP_E_LIMIT, "sample_sort3: %llu rows exceed %llu", (unsigned long long)n_bound, (unsigned long long)SS_MAX_N);
    const uint32_t nb = (uint32_t)((n_bound + SS_TILE - 1) / SS_TILE);
    const SsWs w = ss_layout(d_ws, nb);
    { KTimer t(ctx, "ss_sample_kernel");
      hipLaunchKernelGGL(ss_gather_kernel, dim3(SS_SAMPLE / 256), dim3(256), 0, ctx->stream, a, d_n, w.samp, w.flags);
      hipLaunchKernelGGL(ss_sample_kernel, dim3(SS_SAMPLE / 16), dim3(256), 0, ctx->stream, b, d_n, w.samp, w.flags, w.spl); }
    { KTimer t(ctx, "ss_hist_kernel");
      hipLaunchKernelGGL(ss_hist_kernel, dim3(nb), dim3(256), 0, ctx->stream, a, d_n, w.flags, w.spl, nb, w.table, w.ids); }
    { KTimer t(ctx, "ss_rowscan_kernel");
      hipLaunchKernelGGL(ss_rowscan_kernel, dim3(SS_NBUCKET / 8), dim3(256), 0, ctx->stream, w.table, nb); }
    { KTimer t(ctx, "ss_scatter_kernel");
      hipLaunchKernelGGL(ss_scatter_kernel, dim3(nb), dim3(256), 0, ctx->stream, a, b, d_n, w.flags, nb, w.table, w.ids, w.bucket_start); }
    { KTimer t(ctx, "ss_local_kernel");
      hipLaunchKernelGGL((ss_local_kernel<SS_CAP1>), dim3(SS_NBUCKET), dim3(256), 0, ctx->stream, a, b, d_n, w.flags, w.bucket_start, w.big_list, w.spl);
      hipLaunchKernelGGL((ss_local_kernel<SS_CAP>), dim3(64), dim3(256), 0, ctx->stream, a, b, d_n, w.flags, w.bucket_start, w.big_list, w.spl); }
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx
