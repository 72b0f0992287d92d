// db_image.cpp -- SURVEY 8f-2: a device-ready image of one species' graph beside the reference's containers
// (<db>/species_graph_info/<otu>.bin of zip.rs:171-190 stays the source of truth; the image is a cache this library
// writes and reads).  An image holds what the resident DB holds for the species, in the layouts the kernels use, so
// loading is "map the file, copy to HBM": 32-bit node lengths, the walks as a CSR of 32-bit local node ids, the
// haplotype names, and the unique-trio index of a7 (trio_nodes_info, profile.rs:658-740) -- lookup CSR over the
// smallest end node plus the (hap, position)-ordered rows -- so that a7 is a load-time no-op.
//
// Layout (little endian, every section padded to 16 bytes):
//   header  : "PTXHIPDB", u32 version, u32 flags (bit 0 = all walks identical), u64 V, H, P, U, L, u64 name_bytes
//   node_len u32[V] | path_off u64[H+1] | path_nodes u32[P] | names ('\n'-joined) |
//   trio_first u32[V+1] (local; CSR over the middle node) | trio_ent {smaller end, larger end, local row, 0} u32x4[U] | trio_abc u32[3U] | trio_hap u32[U] |
//   trio_len u32[U] | hap_trio_off u64[H+1] (local) | u64 end marker = header checksum
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "db_image.hpp"
#include "lad.hpp"

namespace ptx {

namespace {
constexpr char MAGIC[8] = {'P', 'T', 'X', 'H', 'I', 'P', 'D', 'B'};
constexpr uint32_t VERSION = 2;   // 2: the lookup rows are filed under the window's MIDDLE node {smaller end, larger end, row} (1: under the smaller end)
struct Header { char magic[8]; uint32_t version, flags; uint64_t V, H, P, U, L, name_bytes; };
inline uint64_t pad16(uint64_t n) { return (n + 15) & ~uint64_t(15); }
inline uint64_t header_sum(const Header &h) {
    uint64_t x = 0xcbf29ce484222325ull;
    const uint8_t *b = reinterpret_cast<const uint8_t *>(&h);
    for (size_t i = 0; i < sizeof(Header); ++i) { x ^= b[i]; x *= 0x100000001b3ull; }
    return x;
}
struct Layout {
    uint64_t node_len, path_off, path_nodes, names, trio_first, trio_ent, trio_abc, trio_hap, trio_len, hto, end, total;
    explicit Layout(const Header &h) {
        uint64_t o = pad16(sizeof(Header));
        auto take = [&](uint64_t bytes) { const uint64_t at = o; o += pad16(bytes); return at; };
        node_len = take(4 * h.V); path_off = take(8 * (h.H + 1)); path_nodes = take(4 * h.P); names = take(h.name_bytes);
        trio_first = take(4 * (h.V + 1)); trio_ent = take(16 * h.U); trio_abc = take(12 * h.U); trio_hap = take(4 * h.U); trio_len = take(4 * h.U);
        hto = take(8 * (h.H + 1)); end = take(8);
        total = o;
    }
};
}  // namespace

std::string SpeciesImage::open(const std::string &path) {
    const std::string e = mf.open(path);
    if (!e.empty()) return e;
    if (mf.size < sizeof(Header)) return path + ": not a pantax-hip graph image (too short)";
    Header h;
    std::memcpy(&h, mf.data, sizeof(h));
    if (std::memcmp(h.magic, MAGIC, 8) != 0) return path + ": not a pantax-hip graph image";
    if (h.version != VERSION) return path + ": graph image version " + std::to_string(h.version) + ", this build reads " + std::to_string(VERSION);
    if (h.V >= 0xFFFFFFFFull || h.P >= 0xFFFFFFFFull || h.U > h.P || h.H > h.P + 1) return path + ": implausible graph image header";
    const Layout L(h);
    if (L.total != mf.size) return path + ": graph image is truncated or has trailing bytes";
    uint64_t endmark = 0;
    std::memcpy(&endmark, mf.data + L.end, 8);
    if (endmark != header_sum(h)) return path + ": graph image end marker does not match its header";
    V = h.V; H = h.H; P = h.P; U = h.U; L_bases = h.L; all_same = (h.flags & 1u) != 0;
    off_node_len = L.node_len; off_path_nodes = L.path_nodes; off_trio_first = L.trio_first; off_trio_ent = L.trio_ent; off_trio_abc = L.trio_abc;
    off_trio_hap = L.trio_hap; off_trio_len = L.trio_len;
    node_len = reinterpret_cast<const uint32_t *>(mf.data + L.node_len);
    path_off = reinterpret_cast<const uint64_t *>(mf.data + L.path_off);
    path_nodes = reinterpret_cast<const uint32_t *>(mf.data + L.path_nodes);
    trio_first = reinterpret_cast<const uint32_t *>(mf.data + L.trio_first);
    trio_ent = reinterpret_cast<const uint4 *>(mf.data + L.trio_ent);
    trio_abc = reinterpret_cast<const uint32_t *>(mf.data + L.trio_abc);
    trio_hap = reinterpret_cast<const uint32_t *>(mf.data + L.trio_hap);
    trio_len = reinterpret_cast<const uint32_t *>(mf.data + L.trio_len);
    hap_trio_off = reinterpret_cast<const uint64_t *>(mf.data + L.hto);
    if (path_off[0] != 0 || path_off[H] != P || hap_trio_off[0] != 0 || hap_trio_off[H] != U || trio_first[0] != 0 || trio_first[V] != U)
        return path + ": graph image offsets are inconsistent";
    hap_names.clear();
    const char *nb = mf.data + L.names, *ne = nb + h.name_bytes;
    for (const char *p = nb; p < ne;) {
        const char *q = static_cast<const char *>(std::memchr(p, '\n', (size_t)(ne - p)));
        if (!q) q = ne;
        hap_names.emplace_back(p, q);
        p = q + 1;
    }
    if (hap_names.size() != H) return path + ": graph image holds " + std::to_string(hap_names.size()) + " names for " + std::to_string(H) + " haplotypes";
    return "";
}

// one species of a resident db (trio index built) -> file
int db_save_image(Ctx *ctx, Db *db, uint32_t s, const std::vector<std::string> &names, const std::string &path) {
    if (s >= db->S) return fail(ctx, PANTAX_HIP_E_INVALID, "db_save_image: species %u of %u", s, db->S);
    if (!db->trio_built) PTX_TRY(trio_index_build(ctx, db));
    PTX_TRY(trio_keys_ensure(ctx, db));
    PTX_TRY(trio_first_ensure(ctx, db));
    const uint64_t nb = db->h_node_off[s], ne = db->h_node_off[s + 1], h0 = db->h_hap_off[s], h1 = db->h_hap_off[s + 1];
    const uint64_t q0 = db->h_path_off[h0], q1 = db->h_path_off[h1], u0 = db->h_hap_trio_off[h0], u1 = db->h_hap_trio_off[h1];
    if (names.size() != h1 - h0) return fail(ctx, PANTAX_HIP_E_INVALID, "db_save_image: %zu names for %llu haplotypes", names.size(), (unsigned long long)(h1 - h0));
    Header h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, MAGIC, 8);
    h.version = VERSION; h.flags = db->h_all_same[s] ? 1u : 0u;
    h.V = ne - nb; h.H = h1 - h0; h.P = q1 - q0; h.U = u1 - u0;
    std::string joined;
    for (size_t i = 0; i < names.size(); ++i) {
        if (names[i].find('\n') != std::string::npos) return fail(ctx, PANTAX_HIP_E_INVALID, "db_save_image: haplotype name with a line break");
        joined += names[i];
        if (i + 1 < names.size()) joined += '\n';
    }
    h.name_bytes = joined.size();
    std::vector<uint64_t> bo(2);
    PTX_TRY(download(ctx, &bo[0], db->d_bit_off.p + nb, 1)); PTX_TRY(download(ctx, &bo[1], db->d_bit_off.p + ne, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h.L = bo[1] - bo[0];
    const Layout L(h);
    std::vector<uint8_t> img(L.total, 0);
    std::memcpy(img.data(), &h, sizeof(h));
    uint32_t *tf = reinterpret_cast<uint32_t *>(img.data() + L.trio_first);
    uint4 *te = reinterpret_cast<uint4 *>(img.data() + L.trio_ent);
    uint64_t *po = reinterpret_cast<uint64_t *>(img.data() + L.path_off), *hto = reinterpret_cast<uint64_t *>(img.data() + L.hto);
    PTX_TRY(download(ctx, reinterpret_cast<uint32_t *>(img.data() + L.node_len), db->d_node_len.p + nb, h.V));
    PTX_TRY(download(ctx, reinterpret_cast<uint32_t *>(img.data() + L.path_nodes), db->d_path_nodes.p + q0, h.P));
    PTX_TRY(download(ctx, tf, db->d_trio_first.p + nb, h.V + 1));
    if (h.U) {
        PTX_TRY(download(ctx, te, db->d_trio_ent.p + u0, h.U));
        PTX_TRY(download(ctx, reinterpret_cast<uint32_t *>(img.data() + L.trio_abc), db->d_trio_abc.p + 3 * u0, 3 * h.U));
        PTX_TRY(download(ctx, reinterpret_cast<uint32_t *>(img.data() + L.trio_hap), db->d_trio_hap.p + u0, h.U));
        PTX_TRY(download(ctx, reinterpret_cast<uint32_t *>(img.data() + L.trio_len), db->d_trio_len.p + u0, h.U));
    }
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t i = 0; i <= h.H; ++i) { po[i] = db->h_path_off[h0 + i] - q0; hto[i] = db->h_hap_trio_off[h0 + i] - u0; }
    if (tf[0] != (uint32_t)u0 || tf[h.V] != (uint32_t)u1) return fail(ctx, PANTAX_HIP_E_STATE, "db_save_image: lookup rows of species %u are not the contiguous block its table rows are", s);
    for (uint64_t i = 0; i <= h.V; ++i) tf[i] -= (uint32_t)u0;
    for (uint64_t i = 0; i < h.U; ++i) { te[i].x -= (uint32_t)nb; te[i].y -= (uint32_t)nb; te[i].z -= (uint32_t)u0; }   // species-local (b, c, row) in the file
    std::memcpy(img.data() + L.names, joined.data(), joined.size());
    const uint64_t endmark = header_sum(h);
    std::memcpy(img.data() + L.end, &endmark, 8);
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", tmp.c_str());
    const bool ok = std::fwrite(img.data(), 1, img.size(), f) == img.size();
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return fail(ctx, PANTAX_HIP_E_IO, "short write to %s", path.c_str()); }
    return 0;
}

// images -> resident db with its trio index in place.  The big arrays stream from the files (pread into pinned chunks,
// DMA); the node tables, the walk check and the move of the species-local trio rows to their place in the batch run
// on the device (stage_db.hip) -- the host touches the headers, the walk offsets and the names only.
int db_from_images(Ctx *ctx, uint32_t S, const SpeciesImage *const *im, const int64_t *range_start, const int64_t *range_end, pantax_hip_db **out) {
    *out = nullptr;
    std::vector<GraphPart> parts(S);
    for (uint32_t s = 0; s < S; ++s) {
        parts[s] = GraphPart{nullptr, im[s]->V, im[s]->H, im[s]->path_off, im[s]->path_nodes};
        parts[s].node_len32 = im[s]->node_len;
        parts[s].fd = im[s]->mf.fd; parts[s].off_node_len = im[s]->off_node_len; parts[s].off_path_nodes = im[s]->off_path_nodes;
        parts[s].n_bases = im[s]->L_bases; parts[s].all_same = im[s]->all_same ? 1 : 0;
    }
    pantax_hip_db *raw = nullptr;
    PTX_TRY(db_upload_parts(ctx, S, range_start, range_end, parts.data(), &raw));
    std::unique_ptr<pantax_hip_db> db(raw);
    const uint64_t V = db->V, H = db->H;
    std::vector<uint64_t> ubase(S + 1, 0);
    for (uint32_t s = 0; s < S; ++s) ubase[s + 1] = ubase[s] + im[s]->U;
    const uint64_t Utot = ubase[S];
    if (Utot >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "graph images: %llu unique trios exceed 32-bit rows", (unsigned long long)Utot);
    db->h_hap_trio_off.assign(H + 1, 0);
    for (uint32_t s = 0; s < S; ++s)
        for (uint64_t h = 0; h < im[s]->H; ++h) db->h_hap_trio_off[db->h_hap_off[s] + h] = im[s]->hap_trio_off[h] + ubase[s];
    db->h_hap_trio_off[H] = Utot;
    PTX_HIP(ctx, db->d_trio_first.alloc(V + 1)); PTX_HIP(ctx, db->d_trio_ent.alloc(Utot ? Utot : 1));
    DevBuf<uint32_t> d_err;
    PTX_HIP(ctx, d_err.alloc(1));
    PTX_HIP(ctx, hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), ctx->stream));
    PTX_HIP(ctx, db->d_trio_abc.alloc(3 * (Utot ? Utot : 1))); PTX_HIP(ctx, db->d_trio_hap.alloc(Utot ? Utot : 1)); PTX_HIP(ctx, db->d_trio_len.alloc(Utot ? Utot : 1));
    for (uint32_t s = 0; s < S; ++s) {
        const SpeciesImage &g = *im[s];
        const uint64_t nb = db->h_node_off[s];
        PTX_TRY(upload_file(ctx, db->d_trio_first.p + nb, g.mf.fd, g.off_trio_first, g.V * sizeof(uint32_t)));   // entry V of the image = its U
        PTX_TRY(upload_file(ctx, db->d_trio_ent.p + ubase[s], g.mf.fd, g.off_trio_ent, g.U * sizeof(uint4)));
        PTX_TRY(upload_file(ctx, db->d_trio_abc.p + 3 * ubase[s], g.mf.fd, g.off_trio_abc, 3 * g.U * sizeof(uint32_t)));
        PTX_TRY(upload_file(ctx, db->d_trio_hap.p + ubase[s], g.mf.fd, g.off_trio_hap, g.U * sizeof(uint32_t)));
        PTX_TRY(upload_file(ctx, db->d_trio_len.p + ubase[s], g.mf.fd, g.off_trio_len, g.U * sizeof(uint32_t)));
        PTX_TRY(trio_rebase_launch(ctx, db.get(), s, ubase[s], g.U, d_err.p));
    }
    const uint32_t u32tot = (uint32_t)Utot;
    PTX_HIP(ctx, hipMemcpyAsync(db->d_trio_first.p + V, &u32tot, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PTX_TRY(upload(ctx, db->d_hap_trio_off, db->h_hap_trio_off.data(), H + 1));
    uint32_t h_err = 0;
    PTX_TRY(download(ctx, &h_err, d_err.p, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err) return fail(ctx, PANTAX_HIP_E_LIMIT, "graph images: %u nodes head 2^24 or more unique-trio rows", h_err);
    db->U = db->U_known = Utot;
    db->trio_sizes_known = true;
    db->trio_built = true;
    db->trio_first_valid = true;
    db->trio_keys_built = true;
    db->cov_done = false;
    *out = db.release();
    return 0;
}

}  // namespace ptx

using namespace ptx;

extern "C" int pantax_hip_db_save_images(pantax_hip_ctx *ctx, pantax_hip_db *db, const char *const *paths, const char *const *hap_names) {
    if (!ctx || !db || !paths || !hap_names) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    if (db->d_node_rec.p == nullptr) return fail(ctx, PANTAX_HIP_E_STATE, "db_save_images: the db was uploaded without graphs (ranges only)");
    for (uint32_t s = 0; s < db->S; ++s) {
        std::vector<std::string> names;
        for (uint64_t h = db->h_hap_off[s]; h < db->h_hap_off[s + 1]; ++h) names.emplace_back(hap_names[h] ? hap_names[h] : "");
        PTX_TRY(db_save_image(ctx, db, s, names, paths[s]));
    }
    return 0;
}

extern "C" int pantax_hip_db_load_images(pantax_hip_ctx *ctx, uint32_t n_species, const char *const *paths, const int64_t *range_start,
                                         const int64_t *range_end, pantax_hip_db **out) {
    if (!ctx || !paths || !range_start || !range_end || !out || n_species == 0) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    std::vector<std::unique_ptr<SpeciesImage>> im(n_species);
    std::vector<const SpeciesImage *> ptr(n_species);
    for (uint32_t s = 0; s < n_species; ++s) {
        im[s].reset(new SpeciesImage());
        const std::string e = im[s]->open(paths[s] ? paths[s] : "");
        if (!e.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", e.c_str());
        ptr[s] = im[s].get();
    }
    return db_from_images(ctx, n_species, ptr.data(), range_start, range_end, out);
}
