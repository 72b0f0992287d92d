// stage_db.hip -- device side of loading a resident DB from device-ready images (db_image.cpp): the
// per-node tables are derived from the uploaded 32-bit lengths, the walks are range-checked, and the species-local
// row numbers of the stored unique-trio index are moved to their place in the batch.  Nothing here is on the step's
// path; it replaces host passes over V and P at load time.
#include "common.hpp"
#include "primitives.hpp"
#include "wave.hpp"
#include "scan_chained.hpp"

namespace ptx {

namespace {
// exclusive scan of one species' node lengths (its bases fit 32 bits, checked by the caller) -> global bit offsets
struct LenLoad {
    const uint32_t *len;
    uint32_t *zero_flag;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { const uint32_t l = len[i]; if (l == 0) *zero_flag = 1u; return l; }
};
struct NodeTableStore {
    uint64_t base;
    uint64_t *bit_off;
    uint4 *node_rec;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t l) const {
        const uint64_t bo = base + excl;
        bit_off[i] = bo;
        node_rec[i] = nr_make(bo, l);
    }
};

// one workgroup per haplotype: every node of the walk must lie inside its species' graph (profile.rs:849 would panic)
__global__ void __launch_bounds__(256) walk_check_kernel(const uint64_t *__restrict__ path_off, const uint32_t *__restrict__ path_nodes,
                                                         const uint32_t *__restrict__ hap_species, const uint32_t *__restrict__ node_base,
                                                         uint32_t *__restrict__ bad /* [0] = 1 + first offending hap */) {
    const uint32_t h = blockIdx.x;
    const uint32_t s = hap_species[h];
    const uint32_t nv = node_base[s + 1] - node_base[s];
    uint32_t mx = 0;
    for (uint64_t q = path_off[h] + threadIdx.x; q < path_off[h + 1]; q += 256) mx = max(mx, path_nodes[q]);
    mx = wave_reduce(mx, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    if ((threadIdx.x & 63) == 0 && path_off[h + 1] > path_off[h] && mx >= nv) atomicMin(bad, h + 1);
}

// species-local lookup heads / rows -> their place in the batch
__global__ void __launch_bounds__(256) trio_rebase_kernel(uint32_t n_nodes, uint32_t n_rows, uint32_t row_base, uint32_t node_base, uint32_t *__restrict__ first /* [n_nodes] slice */,
                                                          uint32_t next_first_local /* = local first[n_nodes] */, uint4 *__restrict__ node_rec,
                                                          uint4 *__restrict__ ent /* [n_rows] slice */, uint32_t *__restrict__ err) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_nodes) {
        const uint32_t f = first[i], nx = i + 1 < n_nodes ? first[i + 1] : next_first_local;
        if (nx - f >= NODE_REC_MAX_ROWS) atomicAdd(err, 1u);
        uint4 r = node_rec[i];
        r.y = nr_head(r.y, nx - f, 0xFFu); r.w = f + row_base;   // the lookup head rides in the node record (no filter: every row is fetched)
        node_rec[i] = r;
    }
    if (i < n_rows) { uint4 e = ent[i]; e.x += node_base; e.y += node_base; e.z += row_base; ent[i] = e; }   // images hold species-local (b, c, row)
}
__global__ void __launch_bounds__(256) add_u32_kernel(uint32_t n, uint32_t *__restrict__ v, uint32_t add) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] += add;
}
}  // namespace

int node_tables_launch(Ctx *ctx, Db *db, const uint64_t *sp_bits /*[S+1] prefix of the species' bases*/, uint32_t *d_flags /*[2], zeroed*/) {
    for (uint32_t s = 0; s < db->S; ++s) {
        const uint64_t nb = db->h_node_off[s], n = db->h_node_off[s + 1] - nb;
        PTX_TRY(exclusive_scan_fn(ctx, LenLoad{db->d_node_len.p + nb, d_flags}, NodeTableStore{sp_bits[s], db->d_bit_off.p + nb, db->d_node_rec.p + nb}, n,
                                  nullptr, "node_tables_kernel"));
    }
    PTX_HIP(ctx, hipMemcpyAsync(db->d_bit_off.p + db->V, &sp_bits[db->S], sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (db->H) hipLaunchKernelGGL(walk_check_kernel, dim3((uint32_t)db->H), dim3(256), 0, ctx->stream, db->d_path_off.p, db->d_path_nodes.p,
                                  db->d_hap_species.p, db->d_node_base.p, d_flags + 1);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

int trio_rebase_launch(Ctx *ctx, Db *db, uint32_t s, uint64_t row_base, uint64_t n_rows, uint32_t *d_err) {
    const uint64_t nb = db->h_node_off[s], n = db->h_node_off[s + 1] - nb;
    const uint32_t m = (uint32_t)std::max<uint64_t>(n, n_rows);
    if (m == 0) return 0;
    // heads first (they read the local firsts), then the firsts themselves move
    hipLaunchKernelGGL(trio_rebase_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t)n, (uint32_t)n_rows, (uint32_t)row_base, (uint32_t)nb,
                       db->d_trio_first.p + nb, (uint32_t)n_rows, db->d_node_rec.p + nb, db->d_trio_ent.p + row_base, d_err);
    if (n && row_base) hipLaunchKernelGGL(add_u32_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)n, db->d_trio_first.p + nb, (uint32_t)row_base);
    PTX_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace ptx
