// sample_sort_nodes.hip -- the LP rows of MANY species sorted straight from the NODE arrays: no compaction pass.
//
// A row of the LP is a node with a_v > 0 and a non-empty membership mask (profile.rs:1380-1385); the solver wants every
// species' rows ordered by (mask, a).  Round 3 sorted rows that a chained scan had compacted first (16V in, 16n out, one more
// pass over everything; deleted in round 5).  Here a species' SEGMENT is its node range [node_base[s], node_base[s+1]) --
// known on the host -- and the sort's own passes skip the nodes that are no rows:
//   1. ssn_gather / ssn_sample : 4096 evenly spaced nodes of the segment, the rows among them sorted in LDS -> 1023 splitters at even
//                                ranks of the valid samples, stored as an implicit search tree in breadth-first order (a level's
//                                nodes are neighbours in LDS: the descent of 64 lanes meets no systematic bank conflict, where the
//                                upper levels of a binary search over the sorted array all fall on ONE bank)
//                                (a segment of <= 4096 nodes is sorted completely right there)
//   2. ssn_hist                : bucket id of every row (ten tree levels; "equal to splitter j" is its own bucket 2j+1 whose rows need
//                                no sorting -- coverage values tie massively).  A workgroup walks SEVERAL tiles of its segment with one
//                                LDS histogram and stores it as a row of the segment's count matrix: no global atomics
//   3. ssn_offsets / ssn_segscan: column sums of the matrix -> bucket starts, the matrix rewritten as every workgroup's first slot in
//                                every bucket; rows per segment -> first output row of every segment, total row count
//   4. ssn_scatter / ssn_ties  : rows {mask, a} as 16-byte records into their bucket (slots from LDS counters seeded by the matrix row) -- the rows
//                                of the EVEN buckets only: a tie bucket holds copies of one key, so it is written as a fill of the output
//                                (coalesced) and its rows never travel.
//                                One 16-byte store per row is what this pass costs (tools/native/scatter_probe.hip: 2e8 rows into 2048
//                                buckets 4.2 ms, 1024: 3.5, 256: 2.9; the real rows, which tie massively, take 3.15 ms either way)
//   5. ssn_local_wave          : a wave per even bucket: up to 512 rows sorted IN REGISTERS (eight per lane: a bitonic
//                                network whose cross-lane steps are ds_bpermute swaps and whose in-lane steps are plain selects -- no
//                                LDS memory, no barriers), written to the dense output of the segment
//      ssn_local_wave2         : the buckets of 513 .. 1024 rows (a few per cent of them), sixteen rows per lane: a kernel of its own so
//                                that its registers do not cost the first one its waves in flight (2.2 -> 3.1 ms when it was one)
//      ssn_local               : the rare larger buckets through an LDS network; above 4096 rows (every bucket of a species of millions of
//                                nodes) the network runs in place through memory, a workgroup per bucket
// The number of rows of a segment is only known on the device; launch geometry comes from the node counts.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "primitives.hpp"
#include "wave.hpp"

namespace ptx {

namespace {
#ifndef SN_LEVELS
#define SN_LEVELS 10                             // levels of the splitter tree (-DSN_LEVELS=9: measurements)
#endif
constexpr int SN_SAMPLE = 4096;
constexpr int SN_NLEAF = 1 << SN_LEVELS;         // 1024
constexpr int SN_NSPLIT = SN_NLEAF - 1;          // 1023 splitters: three to four valid samples between two of them
constexpr int SN_NBUCKET = 2 * SN_NLEAF;         // 2048 ids (the last odd one stays empty)
#ifndef SN_ITEMS_N
#define SN_ITEMS_N 8
#endif
constexpr int SN_ITEMS = SN_ITEMS_N;                      // nodes per thread and tile
constexpr int SN_TILE = 256 * SN_ITEMS;
constexpr int SN_CAP = 4096;
constexpr int SN_WAVE_CAP = 512;                 // rows a wave of the first local kernel sorts in registers (eight per lane)
constexpr int SN_WAVE_CAP2 = 1024;               // ... of the second one (sixteen per lane: more registers, fewer waves in flight)
constexpr uint32_t SN_TARGET_WGS = 8192;         // workgroups of the two partition kernels over all segments
constexpr uint16_t SN_NO_ROW = 0xFFFFu;

struct Key2 { uint64_t m, a; };
__device__ __forceinline__ bool less2(const Key2 &x, const Key2 &y) { return (x.m < y.m) | ((x.m == y.m) & (x.a < y.a)); }
__device__ __forceinline__ bool eq2(const Key2 &x, const Key2 &y) { return (x.m == y.m) & (x.a == y.a); }

// per-segment workspace (u32 words), SN_WS_WORDS apart (a multiple of four: the tree's 16-byte nodes stay aligned)
constexpr size_t SN_OFF_FLAGS = 0;                                  // [0] small segment, [1] #buckets left to the workgroup-wide sort, [2] # left to the second wave kernel, [3] rows
constexpr size_t SN_OFF_TREE = 4;                                   // {m, a} [SN_NLEAF], node k's children 2k and 2k+1 (node 0 unused)
constexpr size_t SN_OFF_SAMP = SN_OFF_TREE + 4 * SN_NLEAF;          // u64 [2][4096]
constexpr size_t SN_OFF_START = SN_OFF_SAMP + 2 * 2 * SN_SAMPLE;    // [SN_NBUCKET + 1]
constexpr size_t SN_OFF_MED = SN_OFF_START + SN_NBUCKET + 4;        // [SN_NBUCKET] buckets the first wave kernel leaves to the second
constexpr size_t SN_OFF_BIG = SN_OFF_MED + SN_NBUCKET;              // [SN_NBUCKET] buckets of more than SN_WAVE_CAP2 rows
constexpr size_t SN_WS_WORDS = SN_OFF_BIG + SN_NBUCKET;
static_assert(SN_WS_WORDS % 4 == 0, "16-byte tree nodes");

struct Sn {
    const uint32_t *node_base;   // [S + 1] (device)
    const double *ab;            // [V] a_v (0 = no row)
    const uint64_t *mask;        // [V] membership mask (0 = no row); null: formed from the haplotype words (hp)
    RowMaskSource hp;
    uint32_t *ws;                // S x SN_WS_WORDS
    uint32_t *cntm;              // S x G x SN_NBUCKET: counts, then first slots
    uint16_t *ids;               // [V] bucket id of every staged row (same places as `stage`)
    uint32_t *stage_cnt;         // [S x G] rows a partition workgroup staged
    double *c0p, *c0;            // [S x G] / [S] (c0 null: not wanted) sum of the abundances of the nodes with a > 0 and an EMPTY mask: no rows, but |0 - a| of the objective
    uint32_t *seg_n, *seg_out;   // [S] rows of a segment, [S + 1] its first output row
    ulonglong2 *stage;           // [V] scratch: the rows that have to travel (even buckets), compacted per partition workgroup from the node of its first tile on
    ulonglong2 *rows;            // [V] scratch: those rows bucket by bucket, segment s from node_base[s]
    uint64_t *ksp, *km, *ka;     // output: {species, mask, a} (ksp null: species << pack_shift | mask in km)
    int pack_shift;
    uint32_t G, per;             // partition workgroups per segment, tiles each of them walks
    uint32_t skip_empty;         // segments without LP columns are not read by the histogram pass (option no_absent_skip: 0)
    uint32_t ablate;             // -DSSN_ABLATE builds: parts of ssn_hist_kernel left out (measurements; the results are wrong)
    __device__ __forceinline__ uint32_t *w(uint32_t s) const { return ws + (size_t)s * SN_WS_WORDS; }
    __device__ __forceinline__ uint64_t key_word(uint32_t s, uint64_t m) const { return pack_shift >= 0 ? (((uint64_t)s << pack_shift) | m) : m; }
    __device__ __forceinline__ void put(uint32_t s, uint32_t pos, uint64_t m, uint64_t a) const {
        km[pos] = key_word(s, m); ka[pos] = a;
        if (ksp) ksp[pos] = s;
    }
};

// mask == null: the membership mask of node v of segment (= species) s from its haplotype word -- bit k of the mask = some haplotype of
// column k visits the node (what mask_nodes_kernel writes, stage_lad.hip); the plain loop, for the few nodes the samplers look at
__device__ __forceinline__ uint64_t sn_node_mask(const Sn &sn, uint32_t s, uint64_t v) {
    if (sn.mask) return sn.mask[v];
    const int p = sn.hp.sp_p[s];
    const uint64_t h0 = sn.hp.hap_off[s], nh = sn.hp.hap_off[s + 1] - h0;
    if (p <= 0 || p > 64 || nh > 64) return 0ull;
    unsigned long long hm = sn.hp.node_haps[v];
    uint64_t m = 0;
    while (hm) { const int j = __ffsll((long long)hm) - 1; hm &= hm - 1; const int bit = sn.hp.hap_bit[h0 + j]; if (bit >= 0) m |= 1ull << bit; }
    return m;
}

// sorted rank (0-based, among the SN_NSPLIT splitters) of tree node k, and back
__device__ __forceinline__ uint32_t tree_rank(uint32_t k) {
    const uint32_t l = 31u - (uint32_t)__builtin_clz(k), p = k - (1u << l);
    return ((2u * p + 1u) << ((uint32_t)SN_LEVELS - 1u - l)) - 1u;
}
__device__ __forceinline__ uint32_t tree_node(uint32_t rank) {
    const uint32_t q = rank + 1u, tz = (uint32_t)__builtin_ctz(q);
    return (1u << ((uint32_t)SN_LEVELS - 1u - tz)) + ((q >> tz) >> 1);
}

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
// ---------------------------------------------------------------------------------------------
// A wave's register network: 64 * L keys, lane l holds elements l * L .. l * L + L - 1 of the sequence being sorted.
// Step (k, j) of the bitonic network pairs element i with i ^ j; j >= L: the partner sits in lane l ^ (j / L), same register
// -- one ds_bpermute per 32-bit half, then the lane keeps the smaller or the larger key; j < L: both in this lane.
// TWO: keys are (m, a); otherwise `a` alone moves (a bucket between two splitters of one mask).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, int addr /* (partner lane) << 2 */) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <bool TWO>
__device__ __forceinline__ void cmp_swap(uint64_t &m0, uint64_t &a0, uint64_t &m1, uint64_t &a1, bool up) {
    // up: afterwards key0 <= key1; otherwise key0 >= key1.  ONE comparison: equal keys may swap, which changes nothing
    const bool lt10 = TWO ? less2(Key2{m1, a1}, Key2{m0, a0}) : (a1 < a0);
    const bool sw = up == lt10;
    const uint64_t ta = sw ? a1 : a0, tb = sw ? a0 : a1;
    a0 = ta; a1 = tb;
    if (TWO) { const uint64_t tm = sw ? m1 : m0, tn = sw ? m0 : m1; m0 = tm; m1 = tn; }
}
template <int L, bool TWO>
__device__ __forceinline__ void wave_sort_regs(uint64_t (&m)[L], uint64_t (&a)[L]) {
    const uint32_t lane = threadIdx.x & 63;
    // stages whose direction depends on the element's place inside the lane (k < L)
#pragma unroll
    for (int k = 2; k < L; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int e = 0; e < L; ++e)
                if ((e & j) == 0) cmp_swap<TWO>(m[e], a[e], m[e | j], a[e | j], (e & k) == 0);
    // stages k = L .. 64 L: the direction is the lane's
    for (uint32_t kl = 1; kl <= 64; kl <<= 1) {          // kl = k / L
        const bool up = (lane & kl) == 0;
        for (uint32_t jl = kl >> 1; jl > 0; jl >>= 1) {  // cross-lane steps: partner lane ^ jl
            const int addr = (int)((lane ^ jl) << 2);
            const bool keep_min = up == ((lane & jl) == 0);
#pragma unroll
            for (int e = 0; e < L; ++e) {
                const uint64_t oa = lane_xor64(a[e], addr);
                uint64_t om = 0;
                if (TWO) om = lane_xor64(m[e], addr);
                const bool o_lt = TWO ? less2(Key2{om, oa}, Key2{m[e], a[e]}) : (oa < a[e]);
                const bool take = keep_min == o_lt;               // (keeping the larger one: an equal key may be taken, which changes nothing)
                a[e] = take ? oa : a[e];
                if (TWO) m[e] = take ? om : m[e];
            }
        }
#pragma unroll
        for (int j = L >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int e = 0; e < L; ++e)
                if ((e & j) == 0) cmp_swap<TWO>(m[e], a[e], m[e | j], a[e | j], up);
    }
}
// one bucket of n <= 64 L rows: src (16-byte records) -> sorted -> the segment's output at dst
template <int L, bool TWO>
__device__ __forceinline__ void wave_sort_bucket(const Sn &sn, uint32_t s, const ulonglong2 *__restrict__ src, uint32_t n, uint32_t dst, uint64_t mv) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t m[L], a[L];
#pragma unroll
    for (int e = 0; e < L; ++e) {                 // any assignment of rows to elements will do: coalesced loads
        const uint32_t i = (uint32_t)e * 64u + lane;
        m[e] = ~0ull; a[e] = ~0ull;               // pads sort behind every row
        if (i < n) { const ulonglong2 r = src[i]; m[e] = r.x; a[e] = r.y; }
    }
    wave_sort_regs<L, TWO>(m, a);
#pragma unroll
    for (int e = 0; e < L; ++e) {
        const uint32_t i = lane * (uint32_t)L + (uint32_t)e;
        if (i < n) sn.put(s, dst + i, TWO ? m[e] : mv, a[e]);
    }
}

__global__ void __launch_bounds__(256) ssn_gather_kernel(Sn sn) {
    const uint32_t s = blockIdx.y, o = sn.node_base[s], n = sn.node_base[s + 1] - o;
    uint32_t *w = sn.w(s);
    uint64_t *samp = reinterpret_cast<uint64_t *>(w + SN_OFF_SAMP);
    const bool small = n <= (uint32_t)SN_SAMPLE;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;   // grid.x covers SN_SAMPLE
    const uint64_t pos = small ? i : ((uint64_t)i * n) / SN_SAMPLE;
    uint64_t m = ~0ull, a = ~0ull;                       // not a row: sorts last
    const bool dead = !sn.mask && sn.skip_empty && sn.hp.sp_p[s] <= 0;   // a segment without LP columns has no rows: nothing sampled, nothing sorted (round 6)
    if (dead) { if (i == 0) { w[SN_OFF_FLAGS] = small ? 1u : 0u; w[SN_OFF_FLAGS + 1] = 0; w[SN_OFF_FLAGS + 2] = 0; w[SN_OFF_FLAGS + 3] = 0; } return; }
    if (pos < n) {
        const double av = sn.ab[o + pos];
        const uint64_t mv = av > 0.0 ? sn_node_mask(sn, s, o + pos) : 0ull;
        if (av > 0.0 && mv != 0ull) { m = mv; a = (uint64_t)__double_as_longlong(av); }   // positive doubles order like their bit patterns
    }
    samp[i] = m; samp[SN_SAMPLE + i] = a;
    if (i == 0) { w[SN_OFF_FLAGS] = small ? 1u : 0u; w[SN_OFF_FLAGS + 1] = 0; w[SN_OFF_FLAGS + 2] = 0; w[SN_OFF_FLAGS + 3] = 0; }
}
// One 1024-thread workgroup per segment sorts its 4096 samples in LDS; the splitters are the valid samples at even ranks.
__global__ void __launch_bounds__(1024) ssn_sample_kernel(Sn sn) {
    __shared__ uint64_t km[SN_SAMPLE], ka[SN_SAMPLE];
    __shared__ uint32_t s_nv;
    const uint32_t s = blockIdx.x, o = sn.node_base[s], n = sn.node_base[s + 1] - o;
    uint32_t *w = sn.w(s);
    if (n == 0) { if (threadIdx.x == 0) { sn.seg_n[s] = 0; if (sn.c0) sn.c0[s] = 0.0; } return; }
    if (!sn.mask && sn.skip_empty && sn.hp.sp_p[s] <= 0) {   // (see ssn_gather_kernel; the histogram pass writes the empty counts of a large segment)
        if (threadIdx.x == 0) { sn.seg_n[s] = 0; w[SN_OFF_FLAGS + 3] = 0; if (sn.c0) sn.c0[s] = 0.0; }
        return;
    }
    const uint64_t *samp = reinterpret_cast<const uint64_t *>(w + SN_OFF_SAMP);
    const bool small = n <= (uint32_t)SN_SAMPLE;
    if (threadIdx.x == 0) s_nv = 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)SN_SAMPLE; i += 1024) { km[i] = samp[i]; ka[i] = samp[SN_SAMPLE + i]; }
    __syncthreads();
    bitonic2<1024>(km, ka, SN_SAMPLE);
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)SN_SAMPLE; i += 1024) c += ka[i] != ~0ull ? 1u : 0u;
    if (c) atomicAdd(&s_nv, c);
    __syncthreads();
    const uint32_t nv = s_nv;
    if (small) {                                         // every row of the segment, sorted: copied out by the local kernel
        for (uint32_t i = threadIdx.x; i < nv; i += 1024) sn.rows[o + i] = make_ulonglong2(km[i], ka[i]);
        if (threadIdx.x == 0) { sn.seg_n[s] = nv; w[SN_OFF_FLAGS + 3] = nv; }
        if (!sn.mask && sn.hp.ratio) {                   // masks from the haplotype words: this segment's column sums are this kernel's (the histogram pass skips it)
            __shared__ unsigned long long s_r[128];
            if (threadIdx.x < 128) s_r[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += 1024) {
                uint64_t m = sn_node_mask(sn, s, o + i);
                const unsigned long long c = sn.hp.cov[o + i], l = sn.hp.node_len[o + i];
                while (m) { const int k = __ffsll((long long)m) - 1; m &= m - 1; if (c) atomicAdd(&s_r[2 * k], c); atomicAdd(&s_r[2 * k + 1], l); }
            }
            __syncthreads();
            if (threadIdx.x < 128 && s_r[threadIdx.x]) atomicAdd(&sn.hp.ratio[2 * sn.hp.hap_off[s] + threadIdx.x], s_r[threadIdx.x]);
        }
        if (sn.c0) {                                     // the segment's nodes without a column (fixed order: deterministic)
            __shared__ double s_c[16];
            double c = 0.0;
            for (uint32_t i = threadIdx.x; i < n; i += 1024) { const double av = sn.ab[o + i]; if (av > 0.0 && sn_node_mask(sn, s, o + i) == 0ull) c += av; }
            c = wave_reduce(c, [](double x, double y) { return x + y; });
            if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
            __syncthreads();
            if (threadIdx.x == 0) { double t = 0.0; for (int q = 0; q < 16; ++q) t += s_c[q]; sn.c0[s] = t; }
        }
        return;
    }
    ulonglong2 *tree = reinterpret_cast<ulonglong2 *>(w + SN_OFF_TREE);
    for (uint32_t k = threadIdx.x; k < (uint32_t)SN_NLEAF; k += 1024) {
        if (k == 0) { tree[0] = make_ulonglong2(~0ull, ~0ull); continue; }
        const uint32_t j = tree_rank(k);                 // splitter j = the valid sample of rank (j + 1) nv / SN_NLEAF
        uint32_t r = (uint32_t)(((uint64_t)(j + 1) * nv) >> SN_LEVELS);
        if (r >= nv) r = nv ? nv - 1 : 0;
        tree[k] = nv ? make_ulonglong2(km[r], ka[r]) : make_ulonglong2(~0ull, ~0ull);
    }
}

// the tiles [t0, t1) of workgroup g of a segment of n nodes
__device__ __forceinline__ void sn_tiles(const Sn &sn, uint32_t n, uint32_t g, uint32_t &t0, uint32_t &t1) {
    const uint32_t nt = (n + SN_TILE - 1) / SN_TILE;
    t0 = g * sn.per; t1 = t0 + sn.per;
    if (t0 > nt) t0 = nt;
    if (t1 > nt) t1 = nt;
}

// HAPS: no mask array -- the mask of a node is formed here from its haplotype word through byte-wise column tables in (dynamic) LDS, and the
// candidates' covered bases and lengths (path_cov_ratio, profile.rs:1344-1361) are summed while it is in a register: mask_nodes_kernel's
// pass (16V in, 8V out) and this pass's own 8V of masks are gone
// -DSSN_ABLATE + option ssn_ablate (tools/r6_ssn_ablate.sh): 1 no column tables, 2 no column sums, 4 no sums beyond column 8, 8 no tree descent, 16 no
// histogram, 32 nothing staged.  Round 6 at cfg4: 1.98 ms whole, 1.34 ms with ALL of them left out -- the kernel is its four input streams (24 B a node
// at 4.5 TB/s); the LDS conflicts round 5's counters showed cost 0.1 ms (tables), 0.1 (sums), 0 (histogram), and 512-thread workgroups (six waves per
// SIMD behind the same tables instead of four) were slower, 2.06 ms
#ifdef SSN_ABLATE
#define SSN_ABL(b) ((sn.ablate & (b)) != 0u)
#else
#define SSN_ABL(b) false
#endif
template <bool HAPS>
__global__ void __launch_bounds__(256) ssn_hist_kernel(Sn sn) {
    __shared__ ulonglong2 tree[SN_NLEAF];
    __shared__ uint32_t s_hist[SN_NBUCKET];
    extern __shared__ unsigned long long s_dyn_tab[];             // HAPS: [nbyte][256] columns of the haplotypes 8b .. 8b+7 set in a byte value
    __shared__ int s_bit[64];
    __shared__ unsigned long long s_acc[2 * 64];
    const uint32_t s = blockIdx.y, g = blockIdx.x, o = sn.node_base[s], n = sn.node_base[s + 1] - o;
    uint32_t *w = sn.w(s);
    if (n == 0 || w[SN_OFF_FLAGS] != 0) return;
    // a species without LP columns (the species level dropped it, or no haplotype passed the first filter) has no rows: an empty histogram, nothing staged,
    // nothing read (round 6: the work follows the species that are present in the sample; its c0 is never used -- objective_rows_kernel leaves such species out)
    if (HAPS && sn.skip_empty && sn.hp.sp_p[s] <= 0) {           // (workgroup-uniform)
        uint32_t *row0 = sn.cntm + ((size_t)s * sn.G + g) * SN_NBUCKET;
        for (int i = threadIdx.x; i < SN_NBUCKET; i += 256) row0[i] = 0u;
        if (threadIdx.x == 0) { sn.stage_cnt[(size_t)s * sn.G + g] = 0u; if (sn.c0) sn.c0p[(size_t)s * sn.G + g] = 0.0; }
        return;
    }
    uint32_t t0, t1;
    sn_tiles(sn, n, g, t0, t1);
    const ulonglong2 *gt = reinterpret_cast<const ulonglong2 *>(w + SN_OFF_TREE);
    __shared__ uint32_t s_nstage;
    for (int i = threadIdx.x; i < SN_NBUCKET; i += 256) s_hist[i] = 0;
    if (threadIdx.x == 0) s_nstage = 0;
    if (t0 < t1) for (int i = threadIdx.x; i < SN_NLEAF; i += 256) tree[i] = gt[i];
    int p0 = 0, nbyte = 0;
    uint64_t h0 = 0;
    if (HAPS) {
        h0 = sn.hp.hap_off[s];
        const uint64_t nh = sn.hp.hap_off[s + 1] - h0;
        p0 = sn.hp.sp_p[s];
        if (p0 <= 0 || p0 > 64 || nh > 64) p0 = 0;                // no columns (or a species the path walk serves: never with HAPS)
        nbyte = p0 ? (int)((nh + 7) / 8) : 0;
        if (threadIdx.x < 64) s_bit[threadIdx.x] = (p0 && threadIdx.x < nh) ? sn.hp.hap_bit[h0 + threadIdx.x] : -1;
        if (threadIdx.x < 128) s_acc[threadIdx.x] = 0;
    }
    __syncthreads();
    if (HAPS) {
        for (int b = 0; b < nbyte; ++b) {
            unsigned long long e = 0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int bit = s_bit[8 * b + i]; if (((threadIdx.x >> i) & 1u) && bit >= 0) e |= 1ull << bit; }
            s_dyn_tab[b * 256 + threadIdx.x] = e;
        }
        __syncthreads();
    }
    unsigned long long c8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // The rows of the EVEN buckets are the only ones that travel (a tie bucket is written as a fill): they are staged here, compacted and
    // with their bucket id, in the node range of this workgroup's tiles -- the scatter pass reads 18 bytes per such row instead of
    // abundance, mask and an id of EVERY node again
    const uint32_t stage0 = o + t0 * SN_TILE;
    const int lane = threadIdx.x & 63;
    double cacc = 0.0;                                           // abundances of this thread's nodes with an empty mask
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t base = t * SN_TILE + threadIdx.x;
        double av[SN_ITEMS];
        uint64_t mv[SN_ITEMS];
        uint32_t cv[SN_ITEMS], lv[SN_ITEMS];
#pragma unroll
        for (int r = 0; r < SN_ITEMS; ++r) {
            const uint32_t i = base + (uint32_t)r * 256u;
            av[r] = 0.0; mv[r] = 0; cv[r] = 0; lv[r] = 0;
            if (i < n) {
                av[r] = sn.ab[o + i];
                if (HAPS) { mv[r] = sn.hp.node_haps[o + i]; cv[r] = sn.hp.cov[o + i]; lv[r] = sn.hp.node_len[o + i]; }
                else mv[r] = sn.mask[o + i];
            }
        }
#pragma unroll
        for (int r = 0; r < SN_ITEMS; ++r) {
            if (HAPS) {                                           // haplotype word -> columns, and the columns' sums
                const unsigned long long hm = mv[r];
                unsigned long long m = 0ull;
                if (SSN_ABL(1u)) m = hm & ((p0 >= 64 ? 0ull : (1ull << p0)) - 1ull);
                else
                for (int b = 0; b < nbyte; ++b) m |= s_dyn_tab[b * 256 + (int)((hm >> (8 * b)) & 255ull)];
                mv[r] = m;
                if (m && !SSN_ABL(2u)) {
                    const unsigned long long c = cv[r], l = lv[r];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < p0) { const bool on = (m >> k) & 1ull; c8[k] += on ? c : 0ull; l8[k] += on ? l : 0ull; }   // (block-uniform: the columns that exist)
                    unsigned long long rest = SSN_ABL(4u) ? 0ull : m >> 8;
                    while (rest) {
                        const int k = __ffsll((long long)rest) - 1 + 8;
                        rest &= rest - 1;
                        if (c) atomicAdd(&s_acc[2 * k], c);
                        atomicAdd(&s_acc[2 * k + 1], l);
                    }
                }
            }
            uint32_t id = SN_NO_ROW;
            const uint64_t abits = (uint64_t)__double_as_longlong(av[r]);
            if (av[r] > 0.0 && mv[r] == 0ull) cacc += av[r];
            if (av[r] > 0.0 && mv[r] != 0ull) {                  // (nodes behind the segment's end were loaded as zeros)
                const Key2 key{mv[r], abits};
                uint32_t k = 1;
                if (SSN_ABL(8u)) k = (uint32_t)SN_NLEAF + ((uint32_t)(abits >> 30) & (uint32_t)(SN_NLEAF - 1));
                else
#pragma unroll
                for (int l = 0; l < SN_LEVELS; ++l) { const ulonglong2 nd = tree[k]; k = 2u * k + (less2(Key2{nd.x, nd.y}, key) ? 1u : 0u); }
                const uint32_t lo = k - (uint32_t)SN_NLEAF;   // splitters less than the key
                uint32_t eq = 0;
                if (lo < (uint32_t)SN_NSPLIT && !SSN_ABL(8u)) { const ulonglong2 nd = tree[tree_node(lo)]; eq = eq2(Key2{nd.x, nd.y}, key) ? 1u : 0u; }
                id = 2u * lo + eq;
                if (!SSN_ABL(16u)) atomicAdd(&s_hist[id], 1u);
            }
            const bool travels = id != SN_NO_ROW && !(id & 1u) && !SSN_ABL(32u);
            const unsigned long long bal = __ballot(travels);
            if (bal) {                                           // (wave-uniform)
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_nstage, (uint32_t)__popcll(bal));
                wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
                if (travels) {
                    const uint32_t pos = stage0 + wbase + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    sn.stage[pos] = make_ulonglong2(mv[r], abits);
                    sn.ids[pos] = (uint16_t)id;
                }
            }
        }
    }
    __syncthreads();
    uint32_t *row = sn.cntm + ((size_t)s * sn.G + g) * SN_NBUCKET;
    for (int i = threadIdx.x; i < SN_NBUCKET; i += 256) row[i] = s_hist[i];
    if (threadIdx.x == 0) sn.stage_cnt[(size_t)s * sn.G + g] = s_nstage;
    if (sn.c0) {                                                 // (block-uniform) fixed-shape sum: deterministic
        __shared__ double s_c[4];
        cacc = wave_reduce(cacc, [](double x, double y) { return x + y; });
        if (lane == 0) s_c[threadIdx.x >> 6] = cacc;
        __syncthreads();
        if (threadIdx.x == 0) sn.c0p[(size_t)s * sn.G + g] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
    }
    if (HAPS && p0 > 0 && sn.hp.ratio) {                         // (block-uniform) exact integer sums: any order
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= p0) break;
            const unsigned long long cs = wave_reduce(c8[k], [](unsigned long long x, unsigned long long y) { return x + y; });
            const unsigned long long ls = wave_reduce(l8[k], [](unsigned long long x, unsigned long long y) { return x + y; });
            if (lane == 0) {
                if (cs) atomicAdd(&s_acc[2 * k], cs);
                if (ls) atomicAdd(&s_acc[2 * k + 1], ls);
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * p0 && s_acc[threadIdx.x]) atomicAdd(&sn.hp.ratio[2 * h0 + threadIdx.x], s_acc[threadIdx.x]);
    }
}

// bucket starts of a segment; the count matrix becomes the first slot of every workgroup in every bucket
__global__ void __launch_bounds__(256) ssn_offsets_kernel(Sn sn) {
    __shared__ uint32_t s_wave[4];
    const uint32_t s = blockIdx.x, n = sn.node_base[s + 1] - sn.node_base[s];
    uint32_t *w = sn.w(s);
    if (n == 0 || w[SN_OFF_FLAGS] != 0) return;          // (a small segment's row count is the sample kernel's)
    uint32_t *cm = sn.cntm + (size_t)s * sn.G * SN_NBUCKET;
    const uint32_t nt = (n + SN_TILE - 1) / SN_TILE, ng = (nt + sn.per - 1) / sn.per;   // workgroups that hold tiles
    constexpr int BPT = SN_NBUCKET / 256;                 // consecutive buckets per thread (a multiple of four)
    static_assert(BPT % 4 == 0 && BPT >= 4, "16-byte steps");
    const uint32_t b0 = threadIdx.x * BPT;
    uint32_t tot[BPT], sum = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) tot[i] = 0;
    for (uint32_t g = 0; g < ng; ++g) {
#pragma unroll
        for (int q = 0; q < BPT; q += 4) {
            const uint4 c = *reinterpret_cast<const uint4 *>(cm + (size_t)g * SN_NBUCKET + b0 + q);
            tot[q] += c.x; tot[q + 1] += c.y; tot[q + 2] += c.z; tot[q + 3] += c.w;
        }
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) sum += tot[i];
    uint32_t total;
    uint32_t off = block_excl_scan<256>(sum, s_wave, &total);
    uint32_t run[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) { run[i] = off; w[SN_OFF_START + b0 + i] = off; off += tot[i]; }
    if (threadIdx.x == 255) w[SN_OFF_START + SN_NBUCKET] = off;
    for (uint32_t g = 0; g < ng; ++g) {
        uint32_t *p = cm + (size_t)g * SN_NBUCKET + b0;
#pragma unroll
        for (int q = 0; q < BPT; q += 4) {
            const uint4 c = *reinterpret_cast<const uint4 *>(p + q);
            *reinterpret_cast<uint4 *>(p + q) = make_uint4(run[q], run[q + 1], run[q + 2], run[q + 3]);
            run[q] += c.x; run[q + 1] += c.y; run[q + 2] += c.z; run[q + 3] += c.w;
        }
    }
    if (threadIdx.x == 0) {
        sn.seg_n[s] = total; w[SN_OFF_FLAGS + 3] = total;
        if (sn.c0) { double t = 0.0; for (uint32_t g = 0; g < ng; ++g) t += sn.c0p[(size_t)s * sn.G + g]; sn.c0[s] = t; }   // in workgroup order
    }
}
// first output row of every segment (the rows of all segments lie back to back), and the total
__global__ void __launch_bounds__(1024) ssn_segscan_kernel(uint32_t S, const uint32_t *__restrict__ seg_n, uint32_t *__restrict__ seg_out, uint32_t *__restrict__ d_n) {
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < S; base += 1024) {
        const uint32_t i = base + threadIdx.x, v = i < S ? seg_n[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_wave, &tot);
        if (i < S) seg_out[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { seg_out[S] = carry; *d_n = carry; }
}

__global__ void __launch_bounds__(256) ssn_scatter_kernel(Sn sn) {
    __shared__ uint32_t s_slot[SN_NBUCKET];
    const uint32_t s = blockIdx.y, g = blockIdx.x, o = sn.node_base[s], n = sn.node_base[s + 1] - o;
    const uint32_t *w = sn.w(s);
    if (n == 0 || w[SN_OFF_FLAGS] != 0) return;
    uint32_t t0, t1;
    sn_tiles(sn, n, g, t0, t1);
    if (t0 >= t1) return;
    const uint32_t cnt = sn.stage_cnt[(size_t)s * sn.G + g];
    if (cnt == 0) return;
    const uint32_t *row = sn.cntm + ((size_t)s * sn.G + g) * SN_NBUCKET;
    for (int i = threadIdx.x; i < SN_NBUCKET; i += 256) s_slot[i] = row[i];
    __syncthreads();
    // the staged rows of this workgroup (ssn_hist_kernel): the even buckets' rows only -- a row equal to a splitter does not travel at
    // all, its bucket holds copies of ONE key and ssn_ties_kernel writes it as a plain fill (cfg4: 64 % of the rows; long reads, whose
    // coverage values are small integers: nearly all)
    const ulonglong2 *st = sn.stage + o + t0 * SN_TILE;
    const uint16_t *sid = sn.ids + o + t0 * SN_TILE;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 256 * SN_ITEMS) {
        ulonglong2 rec[SN_ITEMS];
        uint32_t id[SN_ITEMS];
#pragma unroll
        for (int r = 0; r < SN_ITEMS; ++r) {
            const uint32_t k = k0 + (uint32_t)r * 256u + threadIdx.x;
            id[r] = SN_NO_ROW; rec[r] = make_ulonglong2(0ull, 0ull);
            if (k < cnt) { id[r] = sid[k]; rec[r] = st[k]; }
        }
#pragma unroll
        for (int r = 0; r < SN_ITEMS; ++r) {
            if (id[r] == SN_NO_ROW) continue;
            const uint32_t pos = atomicAdd(&s_slot[id[r]], 1u);
            sn.rows[o + pos] = rec[r];
        }
    }
}

// The tie buckets (2j + 1: the rows equal to splitter j) as fills of the output: a workgroup takes SN_TIE_ROWS consecutive rows of its
// segment's output and walks the buckets that overlap them (a few large buckets hold most of the rows: by rows, not by buckets)
constexpr uint32_t SN_TIE_ROWS = 8192;
__global__ void __launch_bounds__(256) ssn_ties_kernel(Sn sn) {
    __shared__ uint32_t s_start[SN_NBUCKET + 1];
    const uint32_t s = blockIdx.y, o = sn.node_base[s], nn = sn.node_base[s + 1] - o;
    const uint32_t *w = sn.w(s);
    if (nn == 0 || w[SN_OFF_FLAGS] != 0) return;
    const uint32_t n = w[SN_OFF_FLAGS + 3], r0 = blockIdx.x * SN_TIE_ROWS;
    if (r0 >= n) return;
    const uint32_t r1 = min(n, r0 + SN_TIE_ROWS);
    for (uint32_t i = threadIdx.x; i <= (uint32_t)SN_NBUCKET; i += 256) s_start[i] = w[SN_OFF_START + i];
    __syncthreads();
    uint32_t lo = 0, hi = SN_NBUCKET;                            // first bucket that ends behind r0
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid + 1] <= r0) lo = mid + 1; else hi = mid; }
    const ulonglong2 *tree = reinterpret_cast<const ulonglong2 *>(w + SN_OFF_TREE);
    const uint32_t out = sn.seg_out[s];
    // 256 buckets at a time: the keys of their tie buckets are fetched by all threads at once (a dependent 16-byte load per bucket inside
    // the walk cost 0.1 of the 0.75 ms at cfg4), then the walk writes
    __shared__ ulonglong2 s_key[128];
    for (uint32_t qb = lo & ~1u; qb < (uint32_t)SN_NBUCKET && s_start[qb] < r1; qb += 256) {
        __syncthreads();
        if (threadIdx.x < 128u) {
            const uint32_t q = qb + 2u * threadIdx.x + 1u;       // odd bucket: splitter q >> 1
            if (q < (uint32_t)SN_NBUCKET - 1u && s_start[q + 1] > s_start[q]) s_key[threadIdx.x] = tree[tree_node(q >> 1)];
        }
        __syncthreads();
        for (uint32_t q = qb + 1u; q < qb + 256u && q < (uint32_t)SN_NBUCKET && s_start[q] < r1; q += 2) {
            const uint32_t a = max(s_start[q], r0), e = min(s_start[q + 1], r1);
            if (e <= a) continue;                                // (workgroup-uniform; a non-empty odd bucket has j < SN_NSPLIT)
            const ulonglong2 key = s_key[(q - qb) >> 1];
            for (uint32_t i = a + threadIdx.x; i < e; i += 256) sn.put(s, out + i, key.x, key.y);
        }
    }
}

// A wave per even bucket 2j, sorted in registers (more than SN_WAVE_CAP rows: left on the segment's list for the second kernel).
__global__ void __launch_bounds__(256) ssn_local_wave_kernel(Sn sn) {
    // (flat grids whose waves / workgroups walk several (segment, bucket) items were measured in round 6 for this kernel and the tie fills: 1.10 -> 1.32 ms here at
    // cfg4, +0.2 ms a step at the reference-DB shape with its 2.3 million mostly idle workgroups -- starting workgroups that find nothing is not the cost)
    const uint32_t s = blockIdx.y, o = sn.node_base[s], nn = sn.node_base[s + 1] - o;
    uint32_t *w = sn.w(s);
    if (nn == 0 || w[SN_OFF_FLAGS] != 0) return;
    const uint32_t *bucket_start = w + SN_OFF_START;
    const ulonglong2 *tree = reinterpret_cast<const ulonglong2 *>(w + SN_OFF_TREE);
    const uint32_t lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);   // grid.x * 4 = SN_NLEAF pairs
    const uint32_t out = sn.seg_out[s];
    const uint32_t st = bucket_start[2 * j], st1 = bucket_start[2 * j + 1];   // (the tie bucket 2j + 1 went to the output in the scatter pass)
    const uint32_t m = st1 - st;
    if (m == 0) return;
    const ulonglong2 *src = sn.rows + o + st;
    if (m == 1) { if (lane == 0) { const ulonglong2 r = src[0]; sn.put(s, out + st, r.x, r.y); } return; }
    if (m > (uint32_t)SN_WAVE_CAP) { if (lane == 0) w[SN_OFF_MED + atomicAdd(&w[SN_OFF_FLAGS + 2], 1u)] = 2 * j; return; }
    // between two splitters with the same mask every row has that mask: only `a` moves through the network
    bool one = false;
    uint64_t mv = 0;
    if (j > 0 && j < (uint32_t)SN_NSPLIT) {
        const uint64_t ma = tree[tree_node(j - 1)].x, mb = tree[tree_node(j)].x;
        one = ma == mb; mv = mb;
    }
    const uint32_t dst = out + st;
    if (one) {
        if (m <= 64) wave_sort_bucket<1, false>(sn, s, src, m, dst, mv);
        else if (m <= 128) wave_sort_bucket<2, false>(sn, s, src, m, dst, mv);
        else if (m <= 256) wave_sort_bucket<4, false>(sn, s, src, m, dst, mv);
        else wave_sort_bucket<8, false>(sn, s, src, m, dst, mv);
    } else {
        if (m <= 64) wave_sort_bucket<1, true>(sn, s, src, m, dst, mv);
        else if (m <= 128) wave_sort_bucket<2, true>(sn, s, src, m, dst, mv);
        else if (m <= 256) wave_sort_bucket<4, true>(sn, s, src, m, dst, mv);
        else wave_sort_bucket<8, true>(sn, s, src, m, dst, mv);
    }
}
// The first kernel's list: a wave per bucket of 513 .. SN_WAVE_CAP2 rows, sixteen per lane; larger ones go on the next list
__global__ void __launch_bounds__(256) ssn_local_wave2_kernel(Sn sn) {
    const uint32_t s = blockIdx.y, o = sn.node_base[s], nn = sn.node_base[s + 1] - o;
    uint32_t *w = sn.w(s);
    if (nn == 0 || w[SN_OFF_FLAGS] != 0) return;
    const uint32_t n_work = w[SN_OFF_FLAGS + 2];
    const uint32_t *bucket_start = w + SN_OFF_START;
    const ulonglong2 *tree = reinterpret_cast<const ulonglong2 *>(w + SN_OFF_TREE);
    const uint32_t lane = threadIdx.x & 63, out = sn.seg_out[s];
    for (uint32_t wi = blockIdx.x * 4 + (threadIdx.x >> 6); wi < n_work; wi += gridDim.x * 4) {
        const uint32_t bid = w[SN_OFF_MED + wi], j = bid >> 1;
        const uint32_t st = bucket_start[bid], m = bucket_start[bid + 1] - st;
        if (m > (uint32_t)SN_WAVE_CAP2) { if (lane == 0) w[SN_OFF_BIG + atomicAdd(&w[SN_OFF_FLAGS + 1], 1u)] = bid; continue; }
        bool one = false;
        uint64_t mv = 0;
        if (j > 0 && j < (uint32_t)SN_NSPLIT) {
            const uint64_t ma = tree[tree_node(j - 1)].x, mb = tree[tree_node(j)].x;
            one = ma == mb; mv = mb;
        }
        if (one) wave_sort_bucket<16, false>(sn, s, sn.rows + o + st, m, out + st, mv);
        else wave_sort_bucket<16, true>(sn, s, sn.rows + o + st, m, out + st, mv);
    }
}

// What the wave kernels leave: buckets of more than SN_WAVE_CAP2 rows (an LDS network up to SN_CAP rows, a rank sort through memory
// above), and the copy of a small segment.
__global__ void __launch_bounds__(256) ssn_local_kernel(Sn sn) {
    __shared__ uint64_t km[SN_CAP], ka[SN_CAP];
    const uint32_t s = blockIdx.y, o = sn.node_base[s], nn = sn.node_base[s + 1] - o;
    if (nn == 0) return;
    uint32_t *w = sn.w(s);
    const uint32_t out = sn.seg_out[s];
    if (w[SN_OFF_FLAGS] != 0) {      // small segment: the sample kernel sorted every row into the scratch
        const uint32_t n = w[SN_OFF_FLAGS + 3];
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { const ulonglong2 r = sn.rows[o + i]; sn.put(s, out + i, r.x, r.y); }
        return;
    }
    const uint32_t *bucket_start = w + SN_OFF_START;
    const uint32_t n_work = w[SN_OFF_FLAGS + 1];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint32_t bid = w[SN_OFF_BIG + wi];
        const uint32_t st = bucket_start[bid], m = bucket_start[bid + 1] - st;
        const ulonglong2 *src = sn.rows + o + st;
        const uint32_t dst = out + st;
        __syncthreads();   // LDS reuse across the buckets of this workgroup
        if (m <= (uint32_t)SN_CAP) {
            uint32_t N = 2;
            while (N < m) N <<= 1;
            for (uint32_t i = threadIdx.x; i < N; i += 256) {
                if (i < m) { const ulonglong2 r = src[i]; km[i] = r.x; ka[i] = r.y; } else { km[i] = ~0ull; ka[i] = ~0ull; }
            }
            __syncthreads();
            bitonic2<256>(km, ka, N);
            for (uint32_t i = threadIdx.x; i < m; i += 256) sn.put(s, dst + i, km[i], ka[i]);
            continue;
        }
        // A bucket of more than SN_CAP rows (an unrepresentative sample; every bucket of a segment of millions of rows): the network runs
        // IN PLACE in the scratch, through memory, by this one workgroup -- O(m log^2 m) where the rank sort it replaces was O(m^2).  The
        // variant whose merges start with a MIRROR step compares upwards only, so the places behind m act as +inf pads without existing.
        ulonglong2 *buf = sn.rows + o + st;
        uint32_t N = 2;
        while (N < m) N <<= 1;
        auto exchange = [&](uint32_t i, uint32_t l) {             // i < l < m: the smaller key to i
            const ulonglong2 x = buf[i], y = buf[l];
            if (less2(Key2{y.x, y.y}, Key2{x.x, x.y})) { buf[i] = y; buf[l] = x; }
        };
        for (uint32_t k = 2; k <= N; k <<= 1) {
            const uint32_t hk = k >> 1;
            for (uint32_t t = threadIdx.x; t < N / 2; t += 256) {
                const uint32_t blk = t / hk, off = t - blk * hk, i = blk * k + off, l = blk * k + (k - 1u - off);
                if (l < m) exchange(i, l);
            }
            __threadfence_block();
            __syncthreads();
            for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < N / 2; t += 256) {
                    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
                    if (l < m) exchange(i, l);
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        for (uint32_t i = threadIdx.x; i < m; i += 256) { const ulonglong2 r = buf[i]; sn.put(s, dst + i, r.x, r.y); }
    }
}

// ---------------------------------------------------------------------------------------------
// Patterns = runs of equal mask in a segment's sorted rows (the solver's groups, lad_prepare).  No pass over the rows: a run can only
// begin where the mask of the SPLITTERS changes -- between two splitters of one mask every row has that mask, and the row in front of
// them is a copy of the lower splitter or a row behind it -- so one wave per segment walks the 1023 splitters and reads the rows of the
// few bucket pairs at a change (and of the first and the last pair).  Heads are collected in order in the segment's part of the row
// scratch, which the local kernels have finished with.
// ---------------------------------------------------------------------------------------------
#ifndef SN_HEAD_PAIRS
#define SN_HEAD_PAIRS 64                          // bucket pairs a wave of ssn_heads_kernel looks at (round 6: 16 and 4 measured slower or equal)
#endif
constexpr int SN_HP = SN_HEAD_PAIRS, SN_NWH = SN_NLEAF / SN_HP;   // ... and the waves per segment
__global__ void __launch_bounds__(256) ssn_heads_kernel(Sn sn, uint32_t *__restrict__ sub_k) {
    // wave w of a segment (four per workgroup): the bucket pairs [HP w, HP w + HP), its heads from slot start[2 HP w] of the scratch on
    // (a range holds no more heads than rows); sub_k[s][w] = how many
    constexpr int NW = SN_NWH, HP = SN_HP;
    const uint32_t s = blockIdx.y, o = sn.node_base[s], nn = sn.node_base[s + 1] - o, lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n = nn ? sn.seg_n[s] : 0u;
    const uint32_t *w = sn.w(s);
    const uint32_t out = n ? sn.seg_out[s] : 0u;
    const bool small = n != 0 && w[SN_OFF_FLAGS] != 0;
    const uint32_t *start = w + SN_OFF_START;
    uint32_t cnt = 0;
    if (n != 0 && (!small || wave == 0)) {
        ulonglong2 *heads = sn.rows + o + (small ? 0u : start[2 * HP * wave]);   // {mask word as stored, first row of the run}
        auto scan_rows = [&](uint32_t r0, uint32_t r1) {      // rows [r0, r1) of the output, in order
            for (uint32_t base = r0; base < r1; base += 64) {
                const uint32_t i = base + lane;
                const bool in = i < r1;
                const uint64_t m = in ? sn.km[i] : 0ull, pm = (in && i > out) ? sn.km[i - 1] : 0ull;
                const bool head = in && (i == out || m != pm);
                const uint64_t bal = __ballot(head);
                if (head) heads[cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = make_ulonglong2(m, (uint64_t)i);
                cnt += (uint32_t)__popcll(bal);
            }
        };
        if (small) scan_rows(out, out + n);                   // a small segment has no splitters
        else {
            const ulonglong2 *tree = reinterpret_cast<const ulonglong2 *>(w + SN_OFF_TREE);
            const uint32_t j = HP * wave + (lane < (uint32_t)HP ? lane : 0u);
            bool c = j == 0 || j == (uint32_t)SN_NLEAF - 1;
            if (!c) c = tree[tree_node(j)].x != tree[tree_node(j - 1)].x;
            uint64_t bal = __ballot(c && lane < (uint32_t)HP);
            while (bal) {
                const uint32_t jj = HP * wave + (uint32_t)__builtin_ctzll(bal);
                bal &= bal - 1;
                // the odd bucket of the pair holds copies of ONE key (splitter jj): a head can only be its first row -- the rest is not read (round 6:
                // with fifty strains nearly every splitter changes the mask, and a tie bucket of 1e5 rows kept one wave reading for the whole 0.43 ms)
                const uint32_t e0 = start[2 * jj + 1], e1 = start[2 * jj + 2];
                scan_rows(out + start[2 * jj], out + (e1 > e0 ? e0 + 1u : e1));
            }
        }
    }
    if (lane == 0) sub_k[(size_t)s * NW + wave] = cnt;
}
// first pattern of every segment, the number of patterns, and the end of the last run
__global__ void __launch_bounds__(1024) ssn_patscan_kernel(uint32_t S, const uint32_t *__restrict__ sub_k, uint32_t *__restrict__ sp_pat_off, uint32_t *__restrict__ d_K,
                                                           const uint32_t *__restrict__ d_n, uint32_t *__restrict__ pat_start) {
    constexpr int NW = SN_NWH;
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < S; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = 0;
        if (i < S) for (int q = 0; q < NW; ++q) v += sub_k[(size_t)i * NW + q];
        uint32_t tot;
        const uint32_t ex = block_excl_scan<1024>(v, s_wave, &tot);
        if (i < S) sp_pat_off[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { sp_pat_off[S] = carry; *d_K = carry; pat_start[carry] = *d_n; }
}
__global__ void __launch_bounds__(256) ssn_patfill_kernel(Sn sn, const uint32_t *__restrict__ sub_k, const uint32_t *__restrict__ sp_pat_off, uint64_t *__restrict__ pat_mask,
                                                          uint32_t *__restrict__ pat_start, uint32_t *__restrict__ pat_species) {
    constexpr int NW = SN_NWH;
    const uint32_t s = blockIdx.x, k0 = sp_pat_off[s];
    const uint32_t *w = sn.w(s);
    const bool small = w[SN_OFF_FLAGS] != 0;
    uint32_t before = 0;
    for (int q = 0; q < NW; before += sub_k[(size_t)s * NW + q], ++q) {
        const uint32_t cnt = sub_k[(size_t)s * NW + q];
        if (cnt == 0) continue;
        const ulonglong2 *heads = sn.rows + sn.node_base[s] + (small ? 0u : w[SN_OFF_START + 2 * SN_HP * q]);
        for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
            const ulonglong2 h = heads[i];
            pat_mask[k0 + before + i] = sn.pack_shift >= 0 ? (h.x & ((1ull << sn.pack_shift) - 1ull)) : h.x;
            pat_start[k0 + before + i] = (uint32_t)h.y;
            pat_species[k0 + before + i] = s;
        }
    }
}

void sn_geometry(uint32_t S, uint64_t seg_bound, uint32_t *G, uint32_t *per) {
    const uint64_t nt = std::max<uint64_t>(1, (seg_bound + SN_TILE - 1) / SN_TILE);   // tiles of the largest segment
    const uint64_t target = SN_TARGET_WGS;
    uint64_t p = (nt * S + target - 1) / target;
    if (p < 1) p = 1;
    if (p > nt) p = nt;
    *per = (uint32_t)p;
    *G = (uint32_t)((nt + p - 1) / p);
}
}  // namespace

size_t sample_sort_nodes_ws_elems(uint32_t S, uint64_t seg_bound, uint64_t V) {
    uint32_t G, per;
    sn_geometry(S, seg_bound, &G, &per);
    return (size_t)S * SN_WS_WORDS + (size_t)S * G * (SN_NBUCKET + 1 + 2) + (V + 1) / 2 + (2 + (size_t)SN_NWH) * (size_t)S + 20;
}

// Nodes of segment s: [node_base[s], node_base[s + 1]) (device array, the host knows that no segment exceeds seg_bound <= SS_MAX_N
// nodes); a node is a row when ab > 0 and mask != 0.  Output: the rows of all segments back to back, every segment sorted by
// (mask, a), in (ksp, km, ka) -- ksp null: species << pack_shift | mask in km; *d_n = the number of rows.  rows16: 4 V words of scratch.
int sample_sort_nodes(Ctx *ctx, const double *ab, const uint64_t *mask, const uint32_t *d_node_base, uint32_t S, uint64_t seg_bound, uint64_t V,
                      uint64_t *rows16, uint64_t *ksp, uint64_t *km, uint64_t *ka, int pack_shift, uint32_t *d_ws, uint32_t *d_n, const RowPatterns *pat,
                      const RowMaskSource *haps) {
    if (S == 0 || V == 0) {
        PTX_HIP(ctx, hipMemsetAsync(d_n, 0, sizeof(uint32_t), ctx->stream));
        if (pat) { PTX_HIP(ctx, hipMemsetAsync(pat->d_K, 0, sizeof(uint32_t), ctx->stream)); PTX_HIP(ctx, hipMemsetAsync(pat->sp_pat_off, 0, (S + 1) * sizeof(uint32_t), ctx->stream));
                   PTX_HIP(ctx, hipMemsetAsync(pat->pat_start, 0, sizeof(uint32_t), ctx->stream)); }
        return 0;
    }
    if (seg_bound > SSN_MAX_SEG) return fail(ctx, PANTAX_HIP_E_LIMIT, "sample_sort_nodes: a segment of %llu nodes exceeds %llu", (unsigned long long)seg_bound, (unsigned long long)SSN_MAX_SEG);
    if (S > 65535) return fail(ctx, PANTAX_HIP_E_LIMIT, "sample_sort_nodes: %u segments exceed the launch grid", S);
    Sn sn;
    sn.node_base = d_node_base; sn.ab = ab; sn.mask = haps ? nullptr : mask; sn.ws = d_ws;
    if (haps) { if (haps->max_haps > 64) return fail(ctx, PANTAX_HIP_E_INVALID, "sample_sort_nodes: masks from haplotype words take species of at most 64 haplotypes"); sn.hp = *haps; }
    else if (!mask) return fail(ctx, PANTAX_HIP_E_INVALID, "sample_sort_nodes: neither a mask array nor haplotype words");
    sn_geometry(S, seg_bound, &sn.G, &sn.per);
    sn.ablate = ctx->cfg.ssn_ablate;
    sn.skip_empty = ctx->cfg.no_absent_skip ? 0u : 1u;
    sn.cntm = d_ws + (size_t)S * SN_WS_WORDS;
    sn.stage_cnt = sn.cntm + (size_t)S * sn.G * SN_NBUCKET;
    uint32_t *cw = sn.stage_cnt + (size_t)S * sn.G;               // [S x G] doubles, 8-byte aligned
    cw += ((reinterpret_cast<uintptr_t>(cw) & 7u) ? 1 : 0);
    sn.c0p = reinterpret_cast<double *>(cw);
    sn.c0 = pat ? pat->c0 : nullptr;
    uint32_t *tail = cw + 2 * (size_t)S * sn.G;
    sn.seg_n = tail; sn.seg_out = tail + S;                       // [S], [S + 1]
    uint32_t *sub_k = tail + 2 * (size_t)S + 4;                   // [S][SN_NWH] patterns found by each wave of ssn_heads_kernel
    sn.ids = reinterpret_cast<uint16_t *>(sub_k + (size_t)SN_NWH * S);
    sn.rows = reinterpret_cast<ulonglong2 *>(rows16);
    sn.stage = reinterpret_cast<ulonglong2 *>(rows16) + V;
    sn.ksp = ksp; sn.km = km; sn.ka = ka; sn.pack_shift = pack_shift;
    { KTimer t(ctx, "ssn_sample_kernel");
      hipLaunchKernelGGL(ssn_gather_kernel, dim3(SN_SAMPLE / 256, S), dim3(256), 0, ctx->stream, sn);
      hipLaunchKernelGGL(ssn_sample_kernel, dim3(S), dim3(1024), 0, ctx->stream, sn); }
    { KTimer t(ctx, "ssn_hist_kernel");
      if (haps) hipLaunchKernelGGL(ssn_hist_kernel<true>, dim3(sn.G, S), dim3(256), (size_t)((haps->max_haps + 7) / 8) * 256 * sizeof(unsigned long long), ctx->stream, sn);
      else hipLaunchKernelGGL(ssn_hist_kernel<false>, dim3(sn.G, S), dim3(256), 0, ctx->stream, sn); }
    { KTimer t(ctx, "ssn_offsets_kernel");
      hipLaunchKernelGGL(ssn_offsets_kernel, dim3(S), dim3(256), 0, ctx->stream, sn);
      hipLaunchKernelGGL(ssn_segscan_kernel, dim3(1), dim3(1024), 0, ctx->stream, S, (const uint32_t *)sn.seg_n, sn.seg_out, d_n); }
    { KTimer t(ctx, "ssn_scatter_kernel");
      hipLaunchKernelGGL(ssn_scatter_kernel, dim3(sn.G, S), dim3(256), 0, ctx->stream, sn); }
    { KTimer t(ctx, "ssn_ties_kernel");
      hipLaunchKernelGGL(ssn_ties_kernel, dim3((uint32_t)((seg_bound + SN_TIE_ROWS - 1) / SN_TIE_ROWS), S), dim3(256), 0, ctx->stream, sn); }
    { KTimer t(ctx, "ssn_local_wave_kernel");
      hipLaunchKernelGGL(ssn_local_wave_kernel, dim3(SN_NLEAF / 4, S), dim3(256), 0, ctx->stream, sn);
      hipLaunchKernelGGL(ssn_local_wave2_kernel, dim3(8, S), dim3(256), 0, ctx->stream, sn);
      hipLaunchKernelGGL(ssn_local_kernel, dim3(8, S), dim3(256), 0, ctx->stream, sn); }
    if (pat) {
        KTimer t(ctx, "ssn_heads_kernel");
        hipLaunchKernelGGL(ssn_heads_kernel, dim3(SN_NWH / 4, S), dim3(256), 0, ctx->stream, sn, sub_k);
        hipLaunchKernelGGL(ssn_patscan_kernel, dim3(1), dim3(1024), 0, ctx->stream, S, (const uint32_t *)sub_k, pat->sp_pat_off, pat->d_K, (const uint32_t *)d_n, pat->pat_start);
        hipLaunchKernelGGL(ssn_patfill_kernel, dim3(S), dim3(256), 0, ctx->stream, sn, (const uint32_t *)sub_k, (const uint32_t *)pat->sp_pat_off, pat->pat_mask, pat->pat_start,
                           pat->pat_species);
    }
    PTX_HIP(ctx, hipGetLastError());
    if (ctx->cfg.ssn_debug) {   // measurements: bucket statistics of this sort on stderr (synchronises)
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> h((size_t)S * SN_WS_WORDS);
        PTX_HIP(ctx, hipMemcpy(h.data(), d_ws, h.size() * 4, hipMemcpyDeviceToHost));
        uint64_t n_small = 0, rows = 0, n_med = 0, n_big = 0, n_over = 0, max_b = 0, max_rows = 0, n512 = 0, n256 = 0, nb = 0;
        for (uint32_t sg = 0; sg < S; ++sg) {
            const uint32_t *w = h.data() + (size_t)sg * SN_WS_WORDS;
            rows += w[SN_OFF_FLAGS + 3]; max_rows = std::max<uint64_t>(max_rows, w[SN_OFF_FLAGS + 3]);
            if (w[SN_OFF_FLAGS]) { ++n_small; continue; }
            n_med += w[SN_OFF_FLAGS + 2]; n_big += w[SN_OFF_FLAGS + 1];
            for (int b = 0; b < SN_NBUCKET; b += 2) {
                const uint32_t m = w[SN_OFF_START + b + 1] - w[SN_OFF_START + b];
                if (!m) continue;
                ++nb; max_b = std::max<uint64_t>(max_b, m);
                if (m > 256) ++n256;
                if (m > 512) ++n512;
                if (m > (uint32_t)SN_CAP) ++n_over;
            }
        }
        std::fprintf(stderr, "[ssn] S=%u small=%llu rows=%llu max rows/segment=%llu | even buckets in use %llu, >256: %llu, >512: %llu (list %llu), >1024: %llu, >4096: %llu, largest %llu | G=%u per=%u\n", S,
                     (unsigned long long)n_small, (unsigned long long)rows, (unsigned long long)max_rows, (unsigned long long)nb, (unsigned long long)n256, (unsigned long long)n512,
                     (unsigned long long)n_med, (unsigned long long)n_big, (unsigned long long)n_over, (unsigned long long)max_b, sn.G, sn.per);
    }
    return 0;
}

}  // namespace ptx
