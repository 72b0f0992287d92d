// sample_sort_seg.hip -- the LP rows of MANY species sorted in one batch: every species' rows form one contiguous
// segment (they are emitted species by species), and inside a segment the key is (mask, a) -- two u64 words; the
// species word is constant per segment and never moves.  At 100 species x 2e5 rows the LSD radix sort needs 11 passes
// over 2e7 16-byte records (33 launches, 2.5 ms on MI355X); sample sort touches every row three times.
//
// Same scheme as sample_sort.hip, with blockIdx.y = segment:
//   1. ssg_gather / ssg_sample : 4096 evenly spaced rows of the segment sorted in LDS -> 1023 splitters
//                                (a segment of <= 4096 rows is sorted completely right there)
//   2. ssg_hist                : bucket id of every row (binary search over the splitters in LDS; "equal to splitter j"
//                                is its own bucket 2j+1 whose rows need no sorting -- coverage values tie massively);
//                                bucket totals by one global atomic per (workgroup, non-empty bucket)
//   3. ssg_scatter             : every workgroup scans the 2048 totals itself, claims its range inside every bucket it
//                                feeds (one atomic per non-empty bucket) and moves its rows
//   4. ssg_local<1024>, <4096> : bitonic sort of every bucket in LDS, written back in place (oversized buckets:
//                                rank sort through memory, slow but exact)
// The number of rows of a segment is only known on the device (seg_cnt); launch geometry comes from the host-side bound.
#include <algorithm>
#include <cstdlib>
#include "primitives.hpp"
#include "wave.hpp"

namespace ptx {

namespace {
constexpr int SG_SAMPLE = 4096;
constexpr int SG_NSPLIT = SG_SAMPLE / 4 - 1;     // 1023
constexpr int SG_NBUCKET = 2 * (SG_NSPLIT + 1);  // 2048 ids (the last odd one stays empty)
constexpr int SG_ITEMS = 8;                      // rows per thread of the partition kernels
constexpr int SG_TILE = 256 * SG_ITEMS;
constexpr int SG_CAP = 4096, SG_CAP1 = 1024;
constexpr int SG_LOCAL_GRID = 256;               // workgroups per segment of the first local kernel (8 buckets each)

struct Key2 { uint64_t m, a; };
__device__ __forceinline__ bool less2(const Key2 &x, const Key2 &y) { return (x.m < y.m) | ((x.m == y.m) & (x.a < y.a)); }
__device__ __forceinline__ bool eq2(const Key2 &x, const Key2 &y) { return (x.m == y.m) & (x.a == y.a); }

// per-segment workspace (u32 words), SG_WS_WORDS apart
constexpr size_t SG_OFF_FLAGS = 0;                                  // [0] small segment, [1] #big buckets, [2] #medium buckets (wave path)
constexpr size_t SG_OFF_SPL = 4;                                    // u64 [2][1024]
constexpr size_t SG_OFF_SAMP = SG_OFF_SPL + 2 * 2 * 1024;           // u64 [2][4096]
constexpr size_t SG_OFF_CNT = SG_OFF_SAMP + 2 * 2 * SG_SAMPLE;      // [2048] bucket totals      } zeroed at the start of every sort
constexpr size_t SG_OFF_CUR = SG_OFF_CNT + SG_NBUCKET;              // [2048] bucket cursors     }  (by ssg_gather_kernel)
constexpr size_t SG_OFF_START = SG_OFF_CUR + SG_NBUCKET;            // [2049]
constexpr size_t SG_OFF_BIG = SG_OFF_START + SG_NBUCKET + 4;        // [2048]
constexpr size_t SG_OFF_MED = SG_OFF_BIG + SG_NBUCKET;              // [2048] buckets the wave kernel leaves to the workgroup-wide sort
constexpr size_t SG_WS_WORDS = SG_OFF_MED + SG_NBUCKET;

struct Seg {
    const uint32_t *off, *cnt;   // [S] first row and number of rows of every segment (device)
    uint32_t *ws;                // S x SG_WS_WORDS
    uint16_t *ids;               // one per row (global row index)
    uint32_t wave_rows;          // segments of at most this many rows sort their small buckets a wave each (ssg_local_wave_kernel)
    __device__ __forceinline__ uint32_t *w(uint32_t s) const { return ws + (size_t)s * SG_WS_WORDS; }
};

template <int NT>
__device__ __forceinline__ void bitonic2(uint64_t *km, uint64_t *ka, uint32_t N) {
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const Key2 x{km[i], ka[i]}, y{km[l], ka[l]};
                const bool up = (i & k) == 0;
                if (up ? less2(y, x) : less2(x, y)) { km[i] = y.m; ka[i] = y.a; km[l] = x.m; ka[l] = x.a; }
            }
            __syncthreads();
        }
}
template <int NT>
__device__ __forceinline__ void bitonic1(uint64_t *ka, uint32_t N) {
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint64_t x = ka[i], y = ka[l];
                const bool up = (i & k) == 0;
                if (up ? (y < x) : (x < y)) { ka[i] = y; ka[l] = x; }
            }
            __syncthreads();
        }
}

// The same networks run by ONE wave on its own LDS rows (buckets of up to SG_WAVE_CAP rows): no workgroup barrier between the
// steps, the four waves of a workgroup sort four buckets side by side.  A wave's LDS operations complete in order; the fences keep
// the compiler from moving them across a step.
constexpr int SG_WAVE_CAP = 256;
__device__ __forceinline__ void wave_lds_step() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void bitonic2_wave(uint64_t *km, uint64_t *ka, uint32_t N) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < N / 2; t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const Key2 x{km[i], ka[i]}, y{km[l], ka[l]};
                const bool up = (i & k) == 0;
                if (up ? less2(y, x) : less2(x, y)) { km[i] = y.m; ka[i] = y.a; km[l] = x.m; ka[l] = x.a; }
            }
            wave_lds_step();
        }
}
__device__ __forceinline__ void bitonic1_wave(uint64_t *ka, uint32_t N) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < N / 2; t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint64_t x = ka[i], y = ka[l];
                const bool up = (i & k) == 0;
                if (up ? (y < x) : (x < y)) { ka[i] = y; ka[l] = x; }
            }
            wave_lds_step();
        }
}

__global__ void __launch_bounds__(256) ssg_gather_kernel(Seg sg, const uint64_t *__restrict__ km, const uint64_t *__restrict__ ka) {
    const uint32_t s = blockIdx.y, n = sg.cnt[s], o = sg.off[s];
    uint32_t *w = sg.w(s);
    uint64_t *samp = reinterpret_cast<uint64_t *>(w + SG_OFF_SAMP);
    const bool small = n <= (uint32_t)SG_SAMPLE;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;   // grid.x covers SG_SAMPLE
    const uint64_t pos = small ? i : ((uint64_t)i * n) / SG_SAMPLE;
    const bool ok = pos < n;
    samp[i] = ok ? km[o + pos] : ~0ull;
    samp[SG_SAMPLE + i] = ok ? ka[o + pos] : ~0ull;
    if (i == 0) { w[SG_OFF_FLAGS] = small ? 1u : 0u; w[SG_OFF_FLAGS + 1] = 0; w[SG_OFF_FLAGS + 2] = 0; }
    w[SG_OFF_CNT + i] = 0;   // bucket totals and cursors (2 x 2048 words = the 4096 threads of this segment's gather)
}
// One 1024-thread workgroup per segment sorts its 4096 samples in LDS (bitonic network: 78 steps of 2 compare-exchanges
// per thread); every 4th one in rank order is a splitter.  Ranking by all-pairs compares, as the single-segment sort does,
// costs 4096^2 LDS reads per segment -- 0.77 ms of LDS bandwidth for 100 segments; the segments' networks run side by side.
__global__ void __launch_bounds__(1024) ssg_sample_kernel(Seg sg, uint64_t *__restrict__ bm, uint64_t *__restrict__ ba) {
    __shared__ uint64_t km[SG_SAMPLE], ka[SG_SAMPLE];
    const uint32_t s = blockIdx.x, n = sg.cnt[s], o = sg.off[s];
    if (n == 0) return;
    uint32_t *w = sg.w(s);
    const uint64_t *samp = reinterpret_cast<const uint64_t *>(w + SG_OFF_SAMP);
    uint64_t *spl = reinterpret_cast<uint64_t *>(w + SG_OFF_SPL);
    const bool small = n <= (uint32_t)SG_SAMPLE;
    for (uint32_t i = threadIdx.x; i < (uint32_t)SG_SAMPLE; i += 1024) { km[i] = samp[i]; ka[i] = samp[SG_SAMPLE + i]; }
    __syncthreads();
    bitonic2<1024>(km, ka, SG_SAMPLE);
    if (small) {
        for (uint32_t i = threadIdx.x; i < n; i += 1024) { bm[o + i] = km[i]; ba[o + i] = ka[i]; }   // every row, sorted; copied back by the local kernel
    } else {
        for (uint32_t r = threadIdx.x; r < (uint32_t)SG_SAMPLE; r += 1024)
            if ((r & 3u) == 3u && (r >> 2) < (uint32_t)SG_NSPLIT) { spl[r >> 2] = km[r]; spl[1024 + (r >> 2)] = ka[r]; }
    }
}

__global__ void __launch_bounds__(256) ssg_hist_kernel(Seg sg, const uint64_t *__restrict__ km, const uint64_t *__restrict__ ka) {
    __shared__ uint64_t sm[1024], sa[1024];
    __shared__ uint32_t s_hist[SG_NBUCKET];
    const uint32_t s = blockIdx.y, n = sg.cnt[s], o = sg.off[s];
    uint32_t *w = sg.w(s);
    const uint64_t base = (uint64_t)blockIdx.x * SG_TILE;
    if (base >= n || w[SG_OFF_FLAGS] != 0) return;
    const uint64_t *spl = reinterpret_cast<const uint64_t *>(w + SG_OFF_SPL);
    for (int i = threadIdx.x; i < SG_NBUCKET; i += 256) s_hist[i] = 0;
    for (int i = threadIdx.x; i < SG_NSPLIT; i += 256) { sm[i] = spl[i]; sa[i] = spl[1024 + i]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SG_ITEMS; ++r) {
        const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        if (i >= n) continue;
        const Key2 key{km[o + i], ka[o + i]};
        uint32_t lo = 0, hi = SG_NSPLIT;   // first splitter that is not less than the key
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (less2(Key2{sm[mid], sa[mid]}, key)) lo = mid + 1; else hi = mid;
        }
        const uint32_t bid = (lo < (uint32_t)SG_NSPLIT && eq2(Key2{sm[lo], sa[lo]}, key)) ? 2 * lo + 1 : 2 * lo;
        sg.ids[o + i] = (uint16_t)bid;
        atomicAdd(&s_hist[bid], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SG_NBUCKET; i += 256) { const uint32_t c = s_hist[i]; if (c) atomicAdd(&w[SG_OFF_CNT + i], c); }
}

__global__ void __launch_bounds__(256) ssg_scatter_kernel(Seg sg, const uint64_t *__restrict__ km, const uint64_t *__restrict__ ka,
                                                          uint64_t *__restrict__ bm, uint64_t *__restrict__ ba) {
    __shared__ uint32_t s_start[SG_NBUCKET], s_cnt[SG_NBUCKET], s_base[SG_NBUCKET];
    __shared__ uint32_t s_wave[4];
    const uint32_t s = blockIdx.y, n = sg.cnt[s], o = sg.off[s];
    uint32_t *w = sg.w(s);
    const uint64_t base = (uint64_t)blockIdx.x * SG_TILE;
    if (w[SG_OFF_FLAGS] != 0 || (base >= n && blockIdx.x != 0)) return;   // workgroup 0 always publishes the bucket starts
    {   // exclusive scan of the bucket totals (8 consecutive buckets per thread)
        const uint32_t b0 = threadIdx.x * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = w[SG_OFF_CNT + b0 + i]; sum += v[i]; }
        uint32_t total;
        uint32_t off = block_excl_scan<256>(sum, s_wave, &total);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s_start[b0 + i] = off;
            if (blockIdx.x == 0) w[SG_OFF_START + b0 + i] = off;
            s_cnt[b0 + i] = 0;
            off += v[i];
        }
        if (blockIdx.x == 0 && threadIdx.x == 255) w[SG_OFF_START + SG_NBUCKET] = off;
    }
    __syncthreads();
    uint32_t bid[SG_ITEMS];
#pragma unroll
    for (int r = 0; r < SG_ITEMS; ++r) {
        const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        bid[r] = 0xFFFFFFFFu;
        if (i < n) { bid[r] = sg.ids[o + i]; atomicAdd(&s_cnt[bid[r]], 1u); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SG_NBUCKET; i += 256) {
        const uint32_t c = s_cnt[i];
        s_base[i] = c ? s_start[i] + atomicAdd(&w[SG_OFF_CUR + i], c) : 0u;   // this workgroup's range inside the bucket
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SG_NBUCKET; i += 256) s_cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SG_ITEMS; ++r) {
        const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        if (bid[r] == 0xFFFFFFFFu) continue;
        const uint32_t pos = o + s_base[bid[r]] + atomicAdd(&s_cnt[bid[r]], 1u);
        bm[pos] = km[o + i]; ba[pos] = ka[o + i];
    }
}

// Segments of at most sg.wave_rows rows: a wave per bucket -- copies of the single-key buckets, sorts of up to SG_WAVE_CAP rows on the
// wave's own LDS rows; larger buckets go on the segment's list for ssg_local_kernel<SG_CAP1>, which then does nothing else there.
__global__ void __launch_bounds__(256) ssg_local_wave_kernel(Seg sg, uint64_t *__restrict__ am, uint64_t *__restrict__ aa,
                                                             const uint64_t *__restrict__ bm, const uint64_t *__restrict__ ba) {
    __shared__ uint64_t km[4 * SG_WAVE_CAP], ka[4 * SG_WAVE_CAP];
    const uint32_t s = blockIdx.y, n = sg.cnt[s], o = sg.off[s];
    uint32_t *w = sg.w(s);
    if (n == 0 || n > sg.wave_rows || w[SG_OFF_FLAGS] != 0) return;
    const uint32_t *bucket_start = w + SG_OFF_START;
    uint32_t *med_list = w + SG_OFF_MED;
    const uint64_t *spl = reinterpret_cast<const uint64_t *>(w + SG_OFF_SPL);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *wm = km + wave * SG_WAVE_CAP, *wa = ka + wave * SG_WAVE_CAP;
    for (uint32_t bid = blockIdx.x * 4 + wave; bid < (uint32_t)SG_NBUCKET; bid += gridDim.x * 4) {
        const uint32_t st = o + bucket_start[bid], m = bucket_start[bid + 1] - bucket_start[bid];
        if (m == 0) continue;
        if ((bid & 1u) || m == 1) {   // identical keys (or a single row): nothing to sort
            for (uint32_t i = lane; i < m; i += 64) { am[st + i] = bm[st + i]; aa[st + i] = ba[st + i]; }
            continue;
        }
        if (m > (uint32_t)SG_WAVE_CAP) { if (lane == 0) med_list[atomicAdd(&w[SG_OFF_FLAGS + 2], 1u)] = bid; continue; }
        uint32_t N = 2;
        while (N < m) N <<= 1;
        wave_lds_step();              // the wave's rows are reused from its previous bucket
        const uint32_t sj = bid >> 1;
        if (sj > 0 && sj < (uint32_t)SG_NSPLIT && spl[sj - 1] == spl[sj]) {   // one mask between the two splitters: only `a` moves
            const uint64_t mv = spl[sj];
            for (uint32_t i = lane; i < N; i += 64) wa[i] = i < m ? ba[st + i] : ~0ull;
            wave_lds_step();
            bitonic1_wave(wa, N);
            for (uint32_t i = lane; i < m; i += 64) { am[st + i] = mv; aa[st + i] = wa[i]; }
            continue;
        }
        for (uint32_t i = lane; i < N; i += 64) {
            if (i < m) { wm[i] = bm[st + i]; wa[i] = ba[st + i]; } else { wm[i] = ~0ull; wa[i] = ~0ull; }
        }
        wave_lds_step();
        bitonic2_wave(wm, wa, N);
        for (uint32_t i = lane; i < m; i += 64) { am[st + i] = wm[i]; aa[st + i] = wa[i]; }
    }
}

// CAP = SG_CAP1: every bucket of 2..SG_CAP1 rows (+ the copies of single-key buckets, + the small-segment copy);
// CAP = SG_CAP : the rest.  Rows come from b and end in a.
template <int CAP>
__global__ void __launch_bounds__(256) ssg_local_kernel(Seg sg, uint64_t *__restrict__ am, uint64_t *__restrict__ aa,
                                                        const uint64_t *__restrict__ bm, const uint64_t *__restrict__ ba) {
    __shared__ uint64_t km[CAP], ka[CAP];
    const uint32_t s = blockIdx.y, n = sg.cnt[s], o = sg.off[s];
    if (n == 0) return;
    uint32_t *w = sg.w(s);
    if (w[SG_OFF_FLAGS] != 0) {      // small segment: the sample kernel ranked every row into b
        if (CAP == SG_CAP1)
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { am[o + i] = bm[o + i]; aa[o + i] = ba[o + i]; }
        return;
    }
    const uint32_t *bucket_start = w + SG_OFF_START;
    uint32_t *big_list = w + SG_OFF_BIG;
    const uint64_t *spl = reinterpret_cast<const uint64_t *>(w + SG_OFF_SPL);
    const bool from_wave = CAP == SG_CAP1 && n <= sg.wave_rows;    // the wave kernel did the small buckets: only its list is left
    const uint32_t n_work = from_wave ? w[SG_OFF_FLAGS + 2] : CAP == SG_CAP1 ? (uint32_t)SG_NBUCKET : w[SG_OFF_FLAGS + 1];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint32_t bid = from_wave ? w[SG_OFF_MED + wi] : CAP == SG_CAP1 ? wi : big_list[wi];
        const uint32_t st = o + bucket_start[bid], m = bucket_start[bid + 1] - bucket_start[bid];
        if (m == 0) continue;
        if ((bid & 1u) || m == 1) {   // identical keys (or a single row): nothing to sort
            if (CAP == SG_CAP1)
                for (uint32_t i = threadIdx.x; i < m; i += 256) { am[st + i] = bm[st + i]; aa[st + i] = ba[st + i]; }
            continue;
        }
        if (CAP == SG_CAP1 && m > (uint32_t)SG_CAP1) {   // left to the large instantiation
            if (threadIdx.x == 0) big_list[atomicAdd(&w[SG_OFF_FLAGS + 1], 1u)] = bid;
            continue;
        }
        __syncthreads();   // LDS reuse across the buckets of this workgroup
        if (m <= (uint32_t)CAP) {
            uint32_t N = 2;
            while (N < m) N <<= 1;
            // between two splitters with the same mask every row has that mask: only `a` moves through the network
            const uint32_t sj = bid >> 1;
            if (sj > 0 && sj < (uint32_t)SG_NSPLIT && spl[sj - 1] == spl[sj]) {
                const uint64_t mv = spl[sj];
                for (uint32_t i = threadIdx.x; i < N; i += 256) ka[i] = i < m ? ba[st + i] : ~0ull;
                __syncthreads();
                bitonic1<256>(ka, N);
                for (uint32_t i = threadIdx.x; i < m; i += 256) { am[st + i] = mv; aa[st + i] = ka[i]; }
                continue;
            }
            for (uint32_t i = threadIdx.x; i < N; i += 256) {
                if (i < m) { km[i] = bm[st + i]; ka[i] = ba[st + i]; } else { km[i] = ~0ull; ka[i] = ~0ull; }
            }
            __syncthreads();
            bitonic2<256>(km, ka, N);
            for (uint32_t i = threadIdx.x; i < m; i += 256) { am[st + i] = km[i]; aa[st + i] = ka[i]; }
            continue;
        }
        // oversized bucket (practically never): rank every row against the whole bucket through memory
        for (uint32_t i = threadIdx.x; i < m; i += 256) {
            const Key2 key{bm[st + i], ba[st + i]};
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; ++j) {
                const Key2 ot{bm[st + j], ba[st + j]};
                if (less2(ot, key) || (eq2(ot, key) && j < i)) ++rank;
            }
            am[st + rank] = key.m; aa[st + rank] = key.a;
        }
    }
}
}  // namespace

size_t sample_sort_seg_ws_elems(uint32_t S, uint64_t n_total_bound) { return (size_t)S * SG_WS_WORDS + (n_total_bound + 1) / 2 + 8; }

// Rows of segment s: [seg_off[s], seg_off[s] + seg_cnt[s]) of (am, aa); every segment holds at most seg_bound (<= SS_MAX_N)
// rows.  Sorted in place by (am, aa) inside every segment; (bm, ba) is scratch of the same size.
int sample_sort_seg(Ctx *ctx, uint64_t *am, uint64_t *aa, uint64_t *bm, uint64_t *ba, uint32_t S, uint64_t seg_bound, uint64_t n_total_bound,
                    const uint32_t *d_seg_off, const uint32_t *d_seg_cnt, uint32_t *d_ws) {
    if (S == 0 || seg_bound == 0) return 0;
    if (seg_bound > SS_MAX_N) return fail(ctx, PANTAX_HIP_E_LIMIT, "sample_sort_seg: a segment of %llu rows exceeds %llu", (unsigned long long)seg_bound, (unsigned long long)SS_MAX_N);
    if (S > 65535) return fail(ctx, PANTAX_HIP_E_LIMIT, "sample_sort_seg: %u segments exceed the launch grid", S);
    Seg sg{d_seg_off, d_seg_cnt, d_ws, reinterpret_cast<uint16_t *>(d_ws + (size_t)S * SG_WS_WORDS), 0u};
    // A wave's network is a longer chain of dependent LDS steps than the workgroup's; it pays through the number of buckets in flight:
    // 1000 segments 4.1 -> 3.3 ms (cfg4), 100 segments 0.49 -> 0.78 ms (cfg3); 512-row wave buckets: 5.7 ms.  Hence by the number of segments.
    sg.wave_rows = S >= 400 ? (uint32_t)SS_MAX_N : 0u;
    if (ctx->cfg.ssg_wave_rows) sg.wave_rows = ctx->cfg.ssg_wave_rows;   // measurements / tests
    const uint32_t nb = (uint32_t)((seg_bound + SG_TILE - 1) / SG_TILE);
    { KTimer t(ctx, "ss_sample_kernel");
      hipLaunchKernelGGL(ssg_gather_kernel, dim3(SG_SAMPLE / 256, S), dim3(256), 0, ctx->stream, sg, am, aa);
      hipLaunchKernelGGL(ssg_sample_kernel, dim3(S), dim3(1024), 0, ctx->stream, sg, bm, ba); }
    { KTimer t(ctx, "ss_hist_kernel");
      hipLaunchKernelGGL(ssg_hist_kernel, dim3(nb, S), dim3(256), 0, ctx->stream, sg, am, aa); }
    { KTimer t(ctx, "ss_scatter_kernel");
      hipLaunchKernelGGL(ssg_scatter_kernel, dim3(nb, S), dim3(256), 0, ctx->stream, sg, am, aa, bm, ba); }
    { KTimer t(ctx, "ss_local_kernel");
      if (sg.wave_rows) hipLaunchKernelGGL(ssg_local_wave_kernel, dim3(SG_LOCAL_GRID, S), dim3(256), 0, ctx->stream, sg, am, aa, bm, ba);
      hipLaunchKernelGGL((ssg_local_kernel<SG_CAP1>), dim3(SG_LOCAL_GRID, S), dim3(256), 0, ctx->stream, sg, am, aa, bm, ba);
      hipLaunchKernelGGL((ssg_local_kernel<SG_CAP>), dim3(8, S), dim3(256), 0, ctx->stream, sg, am, aa, bm, ba); }
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx
