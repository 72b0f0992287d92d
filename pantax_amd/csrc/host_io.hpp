// host_io.hpp -- host-side readers/writers of the reference's file contracts (pipeline seam):
//   species_range.txt (sort_range.rs:25-40), species_genomes_stats.txt (stat.rs:127-135),
//   genomes_info.txt (profile.rs:3092-3146), species_gfa/<sp>.gfa (profile.rs:466-545),
//   species_graph_info/<sp>.bin (bincode-1 `Graph`, zip.rs:171-190 / :236-247),
//   GAF (rcls.rs:119-146), and the TSV outputs (rcls.rs:409-420).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ptx {

struct RangeRow { std::string species; int64_t start, end; int32_t is_pan; };
// returns "" on success, else an error message
std::string read_species_range(const std::string &path, std::vector<RangeRow> &out);
std::string read_species_len(const std::string &path, std::vector<std::pair<std::string, double>> &out);

struct GenomeRow { std::string genome_id, strain_taxid, species_taxid, hap_id; };
std::string read_genomes_info(const std::string &path, std::vector<GenomeRow> &out);

struct HostGraph {
    std::vector<int64_t> node_len;
    std::vector<std::string> hap_names;   // byte-wise sorted (BTreeMap order)
    std::vector<uint64_t> path_off;       // [H+1]
    std::vector<uint32_t> path_nodes;
};
std::string read_gfa(const std::string &path, HostGraph &g);        // profile.rs:466-545 (no walk reversal)
std::string read_graph_bin(const std::string &path, HostGraph &g);
// codec 2 = <otu>.bin.lz4 (LZ4 frame), 3 = <otu>.bin.zst: the same bincode image behind a stream codec (zip.rs:191-223, :250-265)
std::string read_graph_zip(const std::string &path, int codec, HostGraph &g);  // bincode 1.3 fixed-int little-endian

struct HostReads {
    std::vector<uint32_t> step_off{0}, node_id, pstart, pend, qlen;
    std::vector<uint8_t> mapq, flags;
    std::vector<uint64_t> id_hash;        // 64-bit hash of read_id (duplicate detection, profile.rs:369-378)
    std::vector<std::pair<uint64_t, uint32_t>> id_span;   // offset/len of read_id in the mapped file (binning report)
    uint64_t n_lines = 0;
    int ids_distinct = -1;                // device tokenizer: 1 = no two reads share an id hash, 0 = some do; -1 = not checked
};
// tokenises the GAF columns rcls.rs:127-137 selects; keeps the mapping alive in `keep` for id_span
struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    int fd = -1;
    ~MappedFile();
    std::string open(const std::string &path);
};
std::string parse_gaf(const MappedFile &mf, HostReads &out, int n_threads);

}  // namespace ptx
// the handle behind pantax_hip_gaf_load / pantax_hip_gaf_load_device
struct pantax_hip_gaf { ptx::MappedFile mf; ptx::HostReads reads; };
namespace ptx {


// polars CsvWriter-style float text: shortest round-trip digits, integral values keep ".0"
std::string fmt_f64(double v);

}  // namespace ptx
