// db_image.hpp -- device-ready per-species graph images; format and rationale in db_image.cpp
#pragma once
#include <string>
#include <vector>
#include "common.hpp"
#include "host_io.hpp"

namespace ptx {

struct SpeciesImage {   // the header, walk offsets and names of an image file; the big arrays stay in the file until the upload streams them
    std::string path;
    uint64_t V = 0, H = 0, P = 0, L_bases = 0;
    std::vector<uint64_t> path_off;       // [H+1]
    std::vector<std::string> hap_names;
    uint64_t off_node_len = 0;                        // byte offsets in the file
    bool len16 = false;                               // node lengths as u16
    uint64_t n_blocks = 0, payload_bytes = 0, off_blk_first = 0, off_blk_off = 0, off_payload = 0;   // the packed walks (common.hpp PackedWalks)
    std::string open(const std::string &path);   // "" or an error text
    void fill_part(GraphPart &pt, int32_t file) const;
};

int db_save_image(Ctx *ctx, Db *db, uint32_t species, const std::vector<std::string> &hap_names, const std::string &path);
int db_from_images(Ctx *ctx, uint32_t S, const SpeciesImage *const *images, const int64_t *range_start, const int64_t *range_end, pantax_hip_db **out);

// where the arrays of a bincode-1 `Graph` file lie (zip.rs:171-190: u64 len + i64s; u64 map len; per entry u64 key len + bytes + u64
// vec len + u64s), read with a handful of small preads: the file seam streams the arrays from there straight into the pinned upload
// ring (64-bit values narrowed on the way) instead of parsing the file into host vectors first
struct BinIndex {
    uint64_t V = 0, off_node_len = 0;
    std::vector<std::string> hap_names;           // in file order = BTreeMap order
    std::vector<uint64_t> walk_off, walk_len;     // per haplotype: byte offset of its u64 node ids, their number
    bool names_ascending = true;                  // byte-wise strictly ascending, as a BTreeMap serialises
};
std::string scan_graph_bin(const std::string &path, BinIndex &out);   // "" or an error text

}  // namespace ptx
