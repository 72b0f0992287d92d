// stage_trio.hip -- a7: the unique-trio index (trio_nodes_info, profile.rs:658-740) on device.
//
// Reference: every 3-window of every haplotype walk, canonicalised by swapping the ends when
// w[0] > w[2] (:672-678); count_per_trio counts every (hap, position) occurrence (:688-702); a
// trio is strain-specific ("unique") iff that count is exactly 1 (:708-716); its length is the
// sum of its three node lengths (:712).  The reference keeps a dense trio x hap presence matrix;
// a unique trio has exactly one owner, so an owner index per row carries the same information.
//
// Device plan (all species of the db in one batch):
//   1. emit one record per path position q: key = (species, a | b, c), payload q  (sentinel key
//      for positions that start no window)                                        [4P in, 20P out]
//   2. LSD radix sort by (species, a, b, c)                                       [40 B/record/pass]
//   3. flag records whose key differs from both neighbours -> uniq_q[q] = 1
//   4. exclusive scan of uniq_q over q  -> row number in (species, hap, position) order
//   5. compact the sorted unique records into the lookup arrays (CSR over the first node)
//   6. fill the row-order arrays (abc, hap, len) and hap_trio_off
// Row order (species, hap, position) replaces the reference's FxHashSet iteration order, which
// is arbitrary; results are compared as keyed sets.
#include <algorithm>
#include "primitives.hpp"

namespace ptx {

constexpr uint64_t SENTINEL = ~0ull;

__device__ __forceinline__ uint32_t find_hap(const uint64_t *__restrict__ path_off, uint32_t H, uint64_t q) {
    uint32_t lo = 0, hi = H;  // last h with path_off[h] <= q
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (path_off[mid] <= q) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

__global__ void __launch_bounds__(256) trio_emit_kernel(uint64_t P, uint32_t H, const uint64_t *__restrict__ path_off,
                                                        const uint32_t *__restrict__ path_nodes,
                                                        const uint32_t *__restrict__ hap_species, uint64_t *__restrict__ k0,
                                                        uint64_t *__restrict__ k1, uint32_t *__restrict__ val) {
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (uint64_t)gridDim.x * 256) {
        uint32_t h = find_hap(path_off, H, q);
        uint64_t key0 = SENTINEL, key1 = SENTINEL;
        if (q + 2 < path_off[h + 1]) {
            uint32_t a = path_nodes[q], b = path_nodes[q + 1], c = path_nodes[q + 2];
            if (a > c) { uint32_t t = a; a = c; c = t; }   // profile.rs:672-678
            key0 = ((uint64_t)hap_species[h] << 32) | a;
            key1 = ((uint64_t)b << 32) | c;
        }
        k0[q] = key0;
        k1[q] = key1;
        val[q] = (uint32_t)q;
    }
}

__global__ void __launch_bounds__(256) trio_flag_kernel(uint64_t n, const uint64_t *__restrict__ k0, const uint64_t *__restrict__ k1,
                                                        const uint32_t *__restrict__ val, uint8_t *__restrict__ uniq_sorted,
                                                        uint8_t *__restrict__ uniq_q) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint64_t a0 = k0[i], a1 = k1[i];
        bool u = a0 != SENTINEL;
        if (u && i > 0 && k0[i - 1] == a0 && k1[i - 1] == a1) u = false;
        if (u && i + 1 < n && k0[i + 1] == a0 && k1[i + 1] == a1) u = false;   // count == 1 (profile.rs:709)
        uniq_sorted[i] = u;
        if (u) uniq_q[val[i]] = 1;
    }
}

__global__ void __launch_bounds__(256) trio_compact_kernel(uint64_t n, const uint64_t *__restrict__ k0, const uint64_t *__restrict__ k1,
                                                           const uint32_t *__restrict__ val, const uint8_t *__restrict__ uniq_sorted,
                                                           const uint32_t *__restrict__ pos_sorted, const uint32_t *__restrict__ row_of_q,
                                                           const uint32_t *__restrict__ node_base, uint2 *__restrict__ trio_bc,
                                                           uint32_t *__restrict__ trio_row, uint32_t *__restrict__ first_cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (!uniq_sorted[i]) continue;
        uint32_t j = pos_sorted[i];
        uint64_t a0 = k0[i], a1 = k1[i];
        trio_bc[j] = make_uint2((uint32_t)(a1 >> 32), (uint32_t)a1);
        trio_row[j] = row_of_q[val[i]];
        atomicAdd(&first_cnt[node_base[(uint32_t)(a0 >> 32)] + (uint32_t)a0], 1u);
    }
}

__global__ void __launch_bounds__(256) trio_rows_kernel(uint64_t P, uint32_t H, const uint64_t *__restrict__ path_off,
                                                        const uint32_t *__restrict__ path_nodes, const uint32_t *__restrict__ hap_species,
                                                        const uint64_t *__restrict__ hap_off, const uint32_t *__restrict__ node_base,
                                                        const uint64_t *__restrict__ bit_off, const uint8_t *__restrict__ uniq_q,
                                                        const uint32_t *__restrict__ row_of_q, uint32_t *__restrict__ abc,
                                                        uint32_t *__restrict__ hap_out, uint32_t *__restrict__ len_out) {
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (uint64_t)gridDim.x * 256) {
        if (!uniq_q[q]) continue;
        uint32_t h = find_hap(path_off, H, q);
        uint32_t s = hap_species[h];
        uint32_t a = path_nodes[q], b = path_nodes[q + 1], c = path_nodes[q + 2];
        if (a > c) { uint32_t t = a; a = c; c = t; }
        uint32_t row = row_of_q[q];
        abc[3ull * row] = a; abc[3ull * row + 1] = b; abc[3ull * row + 2] = c;
        hap_out[row] = h - (uint32_t)hap_off[s];
        uint32_t nb = node_base[s];
        len_out[row] = (uint32_t)((bit_off[nb + a + 1] - bit_off[nb + a]) + (bit_off[nb + b + 1] - bit_off[nb + b]) +
                                  (bit_off[nb + c + 1] - bit_off[nb + c]));   // profile.rs:712
    }
}

__global__ void __launch_bounds__(256) trio_hapoff_kernel(uint32_t H, uint64_t P, const uint64_t *__restrict__ path_off,
                                                          const uint32_t *__restrict__ row_of_q, const uint32_t *__restrict__ total,
                                                          uint64_t *__restrict__ hap_trio_off) {
    uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h > H) return;
    uint64_t q = path_off[h];
    hap_trio_off[h] = (h == H || q >= P) ? (uint64_t)*total : (uint64_t)row_of_q[q];
}

int trio_index_build(Ctx *ctx, Db *db) {
    const uint64_t P = db->P;
    const uint32_t H = (uint32_t)db->H;
    if (P >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "trio_index: %llu path steps exceed 32-bit positions", (unsigned long long)P);
    DevBuf<uint64_t> k0a, k1a, k0b, k1b;
    DevBuf<uint32_t> va, vb, table, scan_tmp, row_of_q, pos_sorted, first_cnt, d_tot;
    DevBuf<uint8_t> uniq_sorted, uniq_q;
    PTX_HIP(ctx, k0a.alloc(P)); PTX_HIP(ctx, k1a.alloc(P)); PTX_HIP(ctx, k0b.alloc(P)); PTX_HIP(ctx, k1b.alloc(P));
    PTX_HIP(ctx, va.alloc(P)); PTX_HIP(ctx, vb.alloc(P));
    PTX_HIP(ctx, table.alloc(sort_table_elems(P)));
    uint64_t scan_n = std::max<uint64_t>(std::max<uint64_t>(P, db->V + 1), 256ull * 2048);
    PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(scan_n)));
    PTX_HIP(ctx, row_of_q.alloc(P)); PTX_HIP(ctx, pos_sorted.alloc(P));
    PTX_HIP(ctx, uniq_sorted.alloc(P)); PTX_HIP(ctx, uniq_q.alloc(P));
    PTX_HIP(ctx, first_cnt.alloc(db->V + 1));
    PTX_HIP(ctx, d_tot.alloc(2));
    PTX_HIP(ctx, hipMemsetAsync(uniq_q.p, 0, (P ? P : 1), ctx->stream));
    PTX_HIP(ctx, hipMemsetAsync(first_cnt.p, 0, (db->V + 1) * sizeof(uint32_t), ctx->stream));
    PTX_HIP(ctx, db->d_hap_trio_off.alloc(H + 1));
    uint32_t Utot = 0;
    if (P) {
        int grid = grid_for(P, 256, ctx->n_cu * 8);
        {
            KTimer t(ctx, "trio_emit_kernel");
            hipLaunchKernelGGL(trio_emit_kernel, dim3(grid), dim3(256), 0, ctx->stream, P, H, db->d_path_off.p, db->d_path_nodes.p,
                               db->d_hap_species.p, k0a.p, k1a.p, va.p);
        }
        // bits actually populated: local node ids < max species node count; species < S (+ sentinel = all ones)
        uint64_t max_local = 1;
        for (uint32_t s = 0; s < db->S; ++s) max_local = std::max<uint64_t>(max_local, db->h_node_off[s + 1] - db->h_node_off[s]);
        int nb_bits = bits_for(max_local);           // sentinel has all bits set, so it still sorts last
        int sp_bits = bits_for(db->S);               // species ids 0..S-1 and the sentinel's high bits
        std::vector<SortPass> passes;
        add_passes(passes, 1, 0, nb_bits);           // c
        add_passes(passes, 1, 32, 32 + nb_bits);     // b
        add_passes(passes, 0, 0, nb_bits);           // a
        add_passes(passes, 0, 32, 32 + sp_bits + 1); // species (+1 bit so the sentinel's ones outrank S-1)
        SortBufs A, B;
        A.nw = B.nw = 2;
        A.k[0] = k0a.p; A.k[1] = k1a.p; A.v = va.p;
        B.k[0] = k0b.p; B.k[1] = k1b.p; B.v = vb.p;
        bool in_b = false;
        PTX_TRY(radix_sort(ctx, A, B, P, passes.data(), (int)passes.size(), table.p, scan_tmp.p, &in_b));
        SortBufs Sd = in_b ? B : A;
        {
            KTimer t(ctx, "trio_flag_kernel");
            hipLaunchKernelGGL(trio_flag_kernel, dim3(grid), dim3(256), 0, ctx->stream, P, Sd.k[0], Sd.k[1], Sd.v, uniq_sorted.p, uniq_q.p);
        }
        PTX_TRY(exclusive_scan_u8(ctx, uniq_q.p, row_of_q.p, P, scan_tmp.p, d_tot.p));
        PTX_TRY(exclusive_scan_u8(ctx, uniq_sorted.p, pos_sorted.p, P, scan_tmp.p, d_tot.p + 1));
        PTX_TRY(download(ctx, &Utot, d_tot.p, 1));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        db->U = Utot;
        PTX_HIP(ctx, db->d_trio_bc.alloc(Utot)); PTX_HIP(ctx, db->d_trio_row.alloc(Utot));
        PTX_HIP(ctx, db->d_trio_abc.alloc(3ull * Utot)); PTX_HIP(ctx, db->d_trio_hap.alloc(Utot)); PTX_HIP(ctx, db->d_trio_len.alloc(Utot));
        {
            KTimer t(ctx, "trio_compact_kernel");
            hipLaunchKernelGGL(trio_compact_kernel, dim3(grid), dim3(256), 0, ctx->stream, P, Sd.k[0], Sd.k[1], Sd.v, uniq_sorted.p,
                               pos_sorted.p, row_of_q.p, db->d_node_base.p, db->d_trio_bc.p, db->d_trio_row.p, first_cnt.p);
            hipLaunchKernelGGL(trio_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, P, H, db->d_path_off.p, db->d_path_nodes.p,
                               db->d_hap_species.p, db->d_hap_off.p, db->d_node_base.p, db->d_bit_off.p, uniq_q.p, row_of_q.p,
                               db->d_trio_abc.p, db->d_trio_hap.p, db->d_trio_len.p);
            hipLaunchKernelGGL(trio_hapoff_kernel, dim3((H + 1 + 255) / 256), dim3(256), 0, ctx->stream, H, P, db->d_path_off.p,
                               row_of_q.p, d_tot.p, db->d_hap_trio_off.p);
        }
    } else {
        db->U = 0;
        PTX_HIP(ctx, hipMemsetAsync(db->d_hap_trio_off.p, 0, (H + 1) * sizeof(uint64_t), ctx->stream));
    }
    PTX_HIP(ctx, db->d_trio_first.alloc(db->V + 1));
    PTX_TRY(exclusive_scan_u32(ctx, first_cnt.p, db->d_trio_first.p, db->V + 1, scan_tmp.p, nullptr));
    db->h_hap_trio_off.resize(H + 1);
    PTX_TRY(download(ctx, db->h_hap_trio_off.data(), db->d_hap_trio_off.p, H + 1));
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // temporaries are freed on return
    db->trio_built = true;
    db->cov_done = false;
    return 0;
}

}  // namespace ptx

using namespace ptx;
extern "C" {

int pantax_hip_trio_index(pantax_hip_ctx *ctx, pantax_hip_db *db, uint64_t *n_unique_total_out) {
    if (!ctx || !db) return PANTAX_HIP_E_INVALID;
    PTX_HIP(ctx, hipSetDevice(ctx->device));
    if (!db->trio_built) PTX_TRY(trio_index_build(ctx, db));
    if (n_unique_total_out) *n_unique_total_out = db->U;
    return 0;
}

int pantax_hip_trio_get(pantax_hip_ctx *ctx, const pantax_hip_db *db, uint32_t *abc_out, uint32_t *hap_out, int64_t *len_out,
                        uint64_t *hap_trio_off_out) {
    if (!ctx || !db) return PANTAX_HIP_E_INVALID;
    if (!db->trio_built) return fail(ctx, PANTAX_HIP_E_STATE, "trio_get: call pantax_hip_trio_index first");
    PTX_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> len32;
    if (abc_out && db->U) PTX_TRY(download(ctx, abc_out, db->d_trio_abc.p, 3 * db->U));
    if (hap_out && db->U) PTX_TRY(download(ctx, hap_out, db->d_trio_hap.p, db->U));
    if (len_out && db->U) { len32.resize(db->U); PTX_TRY(download(ctx, len32.data(), db->d_trio_len.p, db->U)); }
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (len_out) for (uint64_t u = 0; u < db->U; ++u) len_out[u] = len32[u];
    if (hap_trio_off_out) for (uint64_t h = 0; h <= db->H; ++h) hap_trio_off_out[h] = db->h_hap_trio_off[h];
    return 0;
}
}
