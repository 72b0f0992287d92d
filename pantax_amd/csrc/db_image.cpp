// db_image.cpp -- SURVEY 8f-2: a device-ready image of one species' graph beside the reference's containers
// (<db>/species_graph_info/<otu>.bin of zip.rs:171-190 stays the source of truth; the image is a cache this library
// writes and reads).  An image holds the graph in the layouts the kernels use -- 32-bit node lengths, the walks as a CSR of
// 32-bit local node ids, the haplotype names -- so loading is "pread into the pinned ring, DMA": half the bytes of the `.bin`
// (usize ids are 64-bit there) and no narrowing on the way.
//
// Version 3 (round 5) no longer stores the unique-trio index of a7.  Measured at 10 000 strains (cfg4): the stored index is
// 7.8 GB = 0.14 s of PCIe at the box's 56 GB/s, the device REBUILDS it from the walks in 16 ms (trio_visit_kernel +
// trio_rows_kernel) on top of a visit table built in 75 ms -- and that work can run beside the next transfer, the PCIe bytes cannot.
// The same holds for the visit table itself (4 bytes per path step: 8.9 GB = 0.16 s against 75 ms of kernels).  So the image is
// the graph alone and a7 stays a per-run device build, as in the reference (profile.rs:2936).
//
// Version 4 (round 6) packs the image for the trip over PCIe, which is what bounds a warm load: the walks as blocks of 256 positions of
// zigzag DELTAS (1, 2 or 4 bytes per step, chosen per block; common.hpp PackedWalks) -- node ids run along a walk in steps of one or two,
// so 97 % of the blocks take one byte per step -- and the node lengths as u16 when every length of the species is below 2^16.  At 1e4
// strains: 10.2 GB -> 3.0 GB per db, unpacked in HBM in a few milliseconds (stage_db.hip).
//
// Layout (little endian, every section padded to 16 bytes):
//   header  : "PTXHIPDB", u32 version, u32 flags (bit 0: u16 node lengths), u64 V, H, P, L, u64 name_bytes, u64 n_blocks, u64 payload_bytes
//   node_len u16/u32[V] | path_off u64[H+1] | blk_first u32[n_blocks] | blk_off u32[n_blocks+1] (units of 256 bytes) | payload |
//   names ('\n'-joined) | u64 end marker = header checksum
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
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
constexpr uint32_t VERSION = 4;   // 4: packed walks, 16-bit lengths; 3: the graph alone (2: + the unique-trio index filed under the middle node; 1: under the smaller end)
constexpr uint32_t FLAG_LEN16 = 1u;
struct Header { char magic[8]; uint32_t version, flags; uint64_t V, H, P, L, name_bytes, n_blocks, payload_bytes; };
inline uint64_t pad16(uint64_t n) { return (n + 15) & ~uint64_t(15); }
inline uint64_t header_sum(const Header &h) {
    uint64_t x = 0xcbf29ce484222325ull;
    const uint8_t *b = reinterpret_cast<const uint8_t *>(&h);
    for (size_t i = 0; i < sizeof(Header); ++i) { x ^= b[i]; x *= 0x100000001b3ull; }
    return x;
}
struct Layout {
    uint64_t node_len, path_off, blk_first, blk_off, payload, names, end, total;
    explicit Layout(const Header &h) {
        uint64_t o = pad16(sizeof(Header));
        auto take = [&](uint64_t bytes) { const uint64_t at = o; o += pad16(bytes); return at; };
        node_len = take(((h.flags & FLAG_LEN16) ? 2 : 4) * h.V); path_off = take(8 * (h.H + 1));
        blk_first = take(4 * h.n_blocks); blk_off = take(4 * (h.n_blocks + 1)); payload = take(h.payload_bytes); names = take(h.name_bytes);
        end = take(8);
        total = o;
    }
};
struct Fd {
    int fd = -1;
    ~Fd() { if (fd >= 0) ::close(fd); }
};
bool pread_all(int fd, void *dst, uint64_t n, uint64_t off) {
    uint8_t *d = static_cast<uint8_t *>(dst);
    uint64_t at = 0;
    while (at < n) {
        const ssize_t r = ::pread(fd, d + at, n - at, (off_t)(off + at));
        if (r <= 0) return false;
        at += (uint64_t)r;
    }
    return true;
}
}  // namespace

std::string SpeciesImage::open(const std::string &p) {
    path = p;
    Fd f;
    f.fd = ::open(p.c_str(), O_RDONLY);
    if (f.fd < 0) return "cannot open graph image " + p;
    struct stat st;
    if (fstat(f.fd, &st) != 0) return "cannot stat " + p;
    const uint64_t size = (uint64_t)st.st_size;
    Header h;
    if (size < sizeof(Header) || !pread_all(f.fd, &h, sizeof(h), 0)) return p + ": not a pantax-hip graph image (too short)";
    if (std::memcmp(h.magic, MAGIC, 8) != 0) return p + ": not a pantax-hip graph image";
    if (h.version != VERSION) return p + ": graph image version " + std::to_string(h.version) + ", this build reads " + std::to_string(VERSION);
    if (h.V >= 0xFFFFFFFFull || h.P >= 0xFFFFFFFFull || h.H > h.P + 1 || h.name_bytes > (1ull << 32) || h.n_blocks != (h.P + PK_BLOCK - 1) / PK_BLOCK ||
        h.payload_bytes % PK_UNIT || h.payload_bytes > 4ull * PK_BLOCK * h.n_blocks || h.payload_bytes < (uint64_t)PK_BLOCK * h.n_blocks)
        return p + ": implausible graph image header";
    const Layout L(h);
    if (L.total != size) return p + ": graph image is truncated or has trailing bytes";
    uint64_t endmark = 0;
    if (!pread_all(f.fd, &endmark, 8, L.end) || endmark != header_sum(h)) return p + ": graph image end marker does not match its header";
    V = h.V; H = h.H; P = h.P; L_bases = h.L;
    off_node_len = L.node_len; len16 = (h.flags & FLAG_LEN16) != 0;
    n_blocks = h.n_blocks; payload_bytes = h.payload_bytes; off_blk_first = L.blk_first; off_blk_off = L.blk_off; off_payload = L.payload;
    path_off.assign(H + 1, 0);
    if (!pread_all(f.fd, path_off.data(), 8 * (H + 1), L.path_off)) return p + ": cannot read the walk offsets";
    if (path_off[0] != 0 || path_off[H] != P) return p + ": graph image offsets are inconsistent";
    for (uint64_t i = 0; i < H; ++i) if (path_off[i] > path_off[i + 1]) return p + ": graph image offsets are inconsistent";
    std::string names(h.name_bytes, '\0');
    if (h.name_bytes && !pread_all(f.fd, &names[0], h.name_bytes, L.names)) return p + ": cannot read the haplotype names";
    hap_names.clear();
    const char *nb = names.data(), *ne = nb + names.size();
    for (const char *q = nb; q < ne;) {
        const char *e = static_cast<const char *>(std::memchr(q, '\n', (size_t)(ne - q)));
        if (!e) e = ne;
        hap_names.emplace_back(q, e);
        q = e + 1;
    }
    if (H && names.empty()) hap_names.assign(1, "");     // one haplotype with an empty name
    if (hap_names.size() != H) return p + ": graph image holds " + std::to_string(hap_names.size()) + " names for " + std::to_string(H) + " haplotypes";
    return "";
}

// where the image's sections lie, as the db upload takes them (file = index of the image among the upload's files)
void SpeciesImage::fill_part(GraphPart &pt, int32_t file) const {
    pt.n_nodes = V; pt.n_haps = H; pt.path_off = path_off.data();
    pt.len_seg = UploadSeg(); pt.len_seg.file = file; pt.len_seg.file_off = off_node_len;
    pt.len16 = len16;
    pt.len_seg.out_bytes = len16 ? ((2 * V + 3) & ~3ull) : 4 * V;      // (sections are padded to 16 bytes in the file: the two bytes behind an odd stretch are there)
    pt.walk_segs.clear();
    pt.packed = true;
    pt.pk.n_blocks = n_blocks; pt.pk.payload_bytes = payload_bytes;
    pt.pk.first_seg = UploadSeg(); pt.pk.first_seg.file = file; pt.pk.first_seg.file_off = off_blk_first; pt.pk.first_seg.out_bytes = 4 * n_blocks;
    pt.pk.off_seg = UploadSeg(); pt.pk.off_seg.file = file; pt.pk.off_seg.file_off = off_blk_off; pt.pk.off_seg.out_bytes = 4 * (n_blocks + 1);
    pt.pk.payload_seg = UploadSeg(); pt.pk.payload_seg.file = file; pt.pk.payload_seg.file_off = off_payload; pt.pk.payload_seg.out_bytes = payload_bytes;
}

// one species of a resident db -> file
int db_save_image(Ctx *ctx, Db *db, uint32_t s, const std::vector<std::string> &names, const std::string &path) {
    if (s >= db->S) return fail(ctx, PANTAX_HIP_E_INVALID, "db_save_image: species %u of %u", s, db->S);
    const uint64_t nb = db->h_node_off[s], ne = db->h_node_off[s + 1], h0 = db->h_hap_off[s], h1 = db->h_hap_off[s + 1];
    const uint64_t q0 = db->h_path_off[h0], q1 = db->h_path_off[h1];
    if (names.size() != h1 - h0) return fail(ctx, PANTAX_HIP_E_INVALID, "db_save_image: %zu names for %llu haplotypes", names.size(), (unsigned long long)(h1 - h0));
    Header h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, MAGIC, 8);
    h.version = VERSION; h.flags = 0;
    h.V = ne - nb; h.H = h1 - h0; h.P = q1 - q0;
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
    // node lengths and walks from the device, packed on the host (once per species and database)
    std::vector<uint32_t> lens(h.V), walk(h.P);
    PTX_TRY(download(ctx, lens.data(), db->d_node_len.p + nb, h.V));
    PTX_TRY(download(ctx, walk.data(), db->d_path_nodes.p + q0, h.P));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t lmax = 0;
    for (uint32_t l : lens) lmax = std::max(lmax, l);
    if (lmax < 65536u) h.flags |= FLAG_LEN16;
    h.n_blocks = (h.P + PK_BLOCK - 1) / PK_BLOCK;
    std::vector<uint32_t> blk_first(h.n_blocks), blk_off(h.n_blocks + 1, 0);
    std::vector<uint8_t> payload;
    payload.reserve((size_t)h.P + (size_t)h.P / 16 + 1024);
    {
        uint32_t zz[PK_BLOCK];
        for (uint64_t b = 0; b < h.n_blocks; ++b) {
            const uint64_t p0 = b * PK_BLOCK, n = std::min<uint64_t>(PK_BLOCK, h.P - p0);
            blk_first[b] = walk[p0];
            uint32_t mx = 0;
            zz[0] = 0;
            for (uint64_t i = 1; i < n; ++i) {
                const int64_t d = (int64_t)walk[p0 + i] - (int64_t)walk[p0 + i - 1];   // |d| < 2^32: zigzag of the 32-bit difference (wraps like the decoder's adds)
                const uint32_t d32 = (uint32_t)d;
                zz[i] = (d32 << 1) ^ (uint32_t)((int32_t)d32 >> 31);
                mx = std::max(mx, zz[i]);
            }
            // a difference of 2^31 and more does not survive the 32-bit zigzag: cannot occur (node ids below 2^32 - 1 differ by less than 2^32, and the
            // decoder adds modulo 2^32 -- the round trip is exact for every pair of 32-bit ids)
            for (uint64_t i = n; i < PK_BLOCK; ++i) zz[i] = 0;
            const uint32_t w = mx < 256u ? 1u : mx < 65536u ? 2u : 4u;
            const size_t at = payload.size();
            payload.resize(at + (size_t)PK_BLOCK * w);
            if (w == 1) for (uint32_t i = 0; i < PK_BLOCK; ++i) payload[at + i] = (uint8_t)zz[i];
            else if (w == 2) for (uint32_t i = 0; i < PK_BLOCK; ++i) { const uint16_t x = (uint16_t)zz[i]; std::memcpy(&payload[at + 2 * i], &x, 2); }
            else std::memcpy(&payload[at], zz, sizeof(zz));
            blk_off[b + 1] = blk_off[b] + w * (PK_BLOCK / PK_UNIT);
        }
    }
    h.payload_bytes = payload.size();
    const Layout L(h);
    std::vector<uint8_t> img(L.total, 0);
    std::memcpy(img.data(), &h, sizeof(h));
    uint64_t *po = reinterpret_cast<uint64_t *>(img.data() + L.path_off);
    if (h.flags & FLAG_LEN16) { uint16_t *o = reinterpret_cast<uint16_t *>(img.data() + L.node_len); for (uint64_t v = 0; v < h.V; ++v) o[v] = (uint16_t)lens[v]; }
    else std::memcpy(img.data() + L.node_len, lens.data(), 4 * h.V);
    std::memcpy(img.data() + L.blk_first, blk_first.data(), 4 * h.n_blocks);
    std::memcpy(img.data() + L.blk_off, blk_off.data(), 4 * (h.n_blocks + 1));
    if (!payload.empty()) std::memcpy(img.data() + L.payload, payload.data(), payload.size());
    for (uint64_t i = 0; i <= h.H; ++i) po[i] = db->h_path_off[h0 + i] - q0;
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

// images -> resident db: the two arrays of every species stream from the files (pread into the pinned ring, DMA); the node tables and
// the checks run on the device (stage_db.hip) -- the host touches the headers, the walk offsets and the names only
int db_from_images(Ctx *ctx, uint32_t S, const SpeciesImage *const *im, const int64_t *range_start, const int64_t *range_end, pantax_hip_db **out) {
    *out = nullptr;
    std::vector<GraphPart> parts(S);
    std::vector<std::string> files(S);
    for (uint32_t s = 0; s < S; ++s) {
        files[s] = im[s]->path;
        GraphPart &pt = parts[s];
        pt.n_nodes = im[s]->V; pt.n_haps = im[s]->H; pt.path_off = im[s]->path_off.data();
        im[s]->fill_part(pt, (int32_t)s);
    }
    return db_upload_parts(ctx, S, range_start, range_end, parts.data(), files.data(), out);
}

// ---- bincode-1 `Graph` (types.rs:51-55): where its arrays lie ----
std::string scan_graph_bin(const std::string &path, BinIndex &out) {
    out = BinIndex();
    Fd f;
    f.fd = ::open(path.c_str(), O_RDONLY);
    if (f.fd < 0) return "cannot open serialized graph " + path;
    struct stat st;
    if (fstat(f.fd, &st) != 0) return "cannot stat " + path;
    const uint64_t n = (uint64_t)st.st_size;
    uint64_t off = 0;
    auto rd64 = [&](uint64_t &v) { if (off + 8 > n || !pread_all(f.fd, &v, 8, off)) return false; off += 8; return true; };
    uint64_t nn = 0;
    if (!rd64(nn) || nn > (n - off) / 8) return "truncated graph file " + path;
    out.V = nn; out.off_node_len = off;
    off += nn * 8;
    uint64_t m = 0;
    if (!rd64(m)) return "truncated graph file " + path;
    if (m > (n - off) / 16) return "corrupt graph file " + path;
    // one small read per haplotype carries its key length, its key and -- for keys of the usual size -- the length of its walk
    std::vector<char> buf(4096);
    for (uint64_t i = 0; i < m; ++i) {
        if (off + 8 > n) return "truncated graph file " + path;
        const uint64_t got = std::min<uint64_t>(buf.size(), n - off);
        if (!pread_all(f.fd, buf.data(), got, off)) return "cannot read " + path;
        uint64_t kl = 0, vl = 0;
        std::memcpy(&kl, buf.data(), 8);
        if (kl > (1u << 20) || off + 8 + kl + 8 > n) return "corrupt graph file " + path;
        std::string key(kl, '\0');
        if (8 + kl + 8 <= got) { std::memcpy(&key[0], buf.data() + 8, kl); std::memcpy(&vl, buf.data() + 8 + kl, 8); }
        else if (!pread_all(f.fd, &key[0], kl, off + 8) || !pread_all(f.fd, &vl, 8, off + 8 + kl)) return "cannot read " + path;
        off += 8 + kl + 8;
        if (vl > (n - off) / 8) return "truncated graph file " + path;
        if (!out.hap_names.empty() && !(out.hap_names.back() < key)) out.names_ascending = false;
        out.hap_names.push_back(std::move(key));
        out.walk_off.push_back(off); out.walk_len.push_back(vl);
        off += vl * 8;
    }
    return "";
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
    std::vector<std::string> errs(n_species);
    parallel_for(n_species, 16, [&](uint64_t i0, uint64_t i1) {
        for (uint64_t s = i0; s < i1; ++s) { im[s].reset(new SpeciesImage()); errs[s] = im[s]->open(paths[s] ? paths[s] : ""); ptr[s] = im[s].get(); }
    });
    for (uint32_t s = 0; s < n_species; ++s) if (!errs[s].empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", errs[s].c_str());
    return db_from_images(ctx, n_species, ptr.data(), range_start, range_end, out);
}
