// primitives.hpp -- device-wide exclusive scan and LSD radix sort used by the trio index (a7)
// and the LP row grouping (a10/a12).  Hand-written for wave64 / gfx950: ranks inside a wave come
// from __ballot matching (64-bit masks), block offsets from an LDS table, one launch per stage.
#pragma once
#include "common.hpp"

namespace ptx {

// out[i] = sum_{j<i} in[j] (u32 arithmetic); *d_total (device, may be null) = sum of all.
// d_tmp must hold scan_tmp_elems(n) uint32.
size_t scan_tmp_elems(uint64_t n);
int exclusive_scan_u32(Ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp, uint32_t *d_total);
int exclusive_scan_u8(Ctx *ctx, const uint8_t *d_in, uint32_t *d_out, uint64_t n, uint32_t *d_tmp, uint32_t *d_total);

// large zero fills (16-byte stores from every CU; small ranges go through hipMemsetAsync), on ctx->stream
int zero_fill(Ctx *ctx, void *ptr, size_t bytes);
int byte_fill(Ctx *ctx, void *ptr, int byte, size_t bytes);   // every byte = `byte`, 16-byte stores from every CU

// Records are structure-of-arrays: up to 3 u64 key words + one u32 payload.
constexpr int SORT_MAX_WORDS = 3;
struct SortBufs {
    int nw = 0;                       // key words in use
    uint64_t *k[SORT_MAX_WORDS] = {}; // d pointers
    uint32_t *v = nullptr;            // payload (may be null)
};
struct SortPass {
    int word;   // which key word supplies the digit
    int shift;  // digit = (k[word] >> shift) & 0xFF
};
// Stable LSD radix sort: passes[0] is the LEAST significant digit.  Ping-pongs between a and b;
// returns (via *result_in_b) which side holds the sorted records.  d_table: >= sort_table_elems(n) u32.
size_t sort_table_elems(uint64_t n);
// d_n (optional): the actual record count on the device (<= n); n then only fixes the launch geometry, so the
// caller never has to read the count back.
int radix_sort(Ctx *ctx, SortBufs a, SortBufs b, uint64_t n, const SortPass *passes, int n_passes, uint32_t *d_table,
               uint32_t *d_scan_tmp, bool *result_in_b, const uint32_t *d_n = nullptr);

// shared by both sorts: exclusive scan of every row of table[rows][nb] (nb <= 2048) + row totals at table[rows*nb + row]
void sort_rowscan_launch(Ctx *ctx, uint32_t *d_table, uint32_t rows, uint32_t nb);

// Sample sort of three-word records without payload for inputs of at most SS_MAX_N rows: five launches instead
// of 3 per radix pass (sample_sort.hip).  The sorted rows end up in `a`; `b` is scratch of the same size.
// d_ws: >= sample_sort_ws_elems(n_bound) u32; d_n: the actual row count on the device (<= n_bound).
constexpr uint64_t SS_MAX_N = 600000;
size_t sample_sort_ws_elems(uint64_t n_bound);
int sample_sort3(Ctx *ctx, SortBufs a, SortBufs b, uint64_t n_bound, uint32_t *d_ws, const uint32_t *d_n);

// The batched sort WITHOUT the compaction in front of it (sample_sort_nodes.hip): segment s = the nodes [node_base[s], node_base[s + 1])
// (at most seg_bound <= SS_MAX_N of them), a node is a row when ab > 0 and mask != 0.  The rows of all segments end up back to back in
// (ksp, km, ka) -- ksp null: species << pack_shift | mask in km -- every segment sorted by (mask, a); *d_n = their number.
// rows16: 4 V words of scratch; d_ws: >= sample_sort_nodes_ws_elems(S, seg_bound, V) u32.
// pat (optional): the runs of equal mask inside every segment, in order -- pat_mask / pat_start (first row) / pat_species of run k,
// sp_pat_off[s] = first run of segment s ([S + 1]), *d_K = their number, pat_start[K] = the row count.  Arrays of >= V (+ 1) entries.
struct RowPatterns {
    uint64_t *pat_mask;
    uint32_t *pat_start, *pat_species, *sp_pat_off, *d_K;
    double *c0;   // [S] or null: per segment the sum of ab over the nodes with ab > 0 and mask == 0 (the rows-free part of the LP's objective)
};
constexpr uint64_t SSN_MAX_SEG = 1ull << 26;   // nodes of one segment (buckets grow with the segment: beyond 4096 rows they are sorted through memory)
// mask == null: the sort forms a node's membership mask itself from its haplotype word (and sums the candidates' covered bases and lengths
// on the way: what mask_nodes_kernel does in a pass of its own) -- the mask array is then neither written nor read
struct RowMaskSource {
    const unsigned long long *node_haps = nullptr;   // [V] bit j: haplotype j of the node's species visits it
    const uint64_t *hap_off = nullptr;               // [S + 1]
    const int32_t *hap_bit = nullptr, *sp_p = nullptr;   // [H] LP column of a haplotype or -1; [S] columns of a species
    const uint32_t *cov = nullptr, *node_len = nullptr;  // [V] covered bases, length
    unsigned long long *ratio = nullptr;             // [2 H] at 2 (hap_off[s] + k): sum cov, sum len of column k
    uint32_t max_haps = 0;                           // most haplotypes of a species (<= 64)
};
size_t sample_sort_nodes_ws_elems(uint32_t S, uint64_t seg_bound, uint64_t V);
int sample_sort_nodes(Ctx *ctx, const double *ab, const uint64_t *mask, const uint32_t *d_node_base, uint32_t S, uint64_t seg_bound, uint64_t V,
                      uint64_t *rows16, uint64_t *ksp, uint64_t *km, uint64_t *ka, int pack_shift, uint32_t *d_ws, uint32_t *d_n, const RowPatterns *pat = nullptr,
                      const RowMaskSource *haps = nullptr);

// helper: passes covering bits [lo,hi) of a word, least significant first, appended to out
inline void add_passes(std::vector<SortPass> &out, int word, int lo, int hi) {
    for (int s = lo; s < hi; s += 8) out.push_back({word, s});
}
inline int bits_for(uint64_t max_value) {
    int b = 0;
    while (b < 64 && (max_value >> b)) ++b;
    return b ? b : 1;
}

}  // namespace ptx
