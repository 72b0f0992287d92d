// db_image.hpp -- device-ready per-species graph images; format and rationale in db_image.cpp
#pragma once
#include <string>
#include <vector>
#include "common.hpp"
#include "host_io.hpp"

namespace ptx {

struct SpeciesImage {   // a mapped image file; the pointers look into the mapping
    MappedFile mf;
    uint64_t V = 0, H = 0, P = 0, U = 0, L_bases = 0;
    bool all_same = false;
    const uint32_t *node_len = nullptr, *path_nodes = nullptr, *trio_first = nullptr, *trio_abc = nullptr, *trio_hap = nullptr, *trio_len = nullptr;
    const uint64_t *path_off = nullptr, *hap_trio_off = nullptr;
    const uint4 *trio_ent = nullptr;
    std::vector<std::string> hap_names;
    uint64_t off_node_len = 0, off_path_nodes = 0, off_trio_first = 0, off_trio_ent = 0, off_trio_abc = 0, off_trio_hap = 0, off_trio_len = 0;   // byte offsets in the file
    std::string open(const std::string &path);   // "" or an error text
};

int db_save_image(Ctx *ctx, Db *db, uint32_t species, const std::vector<std::string> &hap_names, const std::string &path);
int db_from_images(Ctx *ctx, uint32_t S, const SpeciesImage *const *images, const int64_t *range_start, const int64_t *range_end, pantax_hip_db **out);

}  // namespace ptx
