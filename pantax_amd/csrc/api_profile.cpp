// api_profile.cpp -- the pipeline seam: pantax_hip_profile(cfg) == profile::profile(ProfilingConfig)
// (profile.rs:3325-3436): files in (GAF + DB files), files out (species_abundance.txt,
// strain_abundance.txt, ori_strain_abundance.txt, optional reads_classification.tsv).
// Host orchestration only; every per-read / per-node computation goes through the device stages.
#include <sys/stat.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include "common.hpp"
#include "db_image.hpp"
#include "host_io.hpp"

using namespace ptx;

namespace {

bool is_file(const std::string &p) { struct stat st; return !p.empty() && stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
// modification time in ns, 0 if the file is missing
int64_t file_mtime(const std::string &p) { struct stat st; return (!p.empty() && stat(p.c_str(), &st) == 0) ? (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec : 0; }
bool is_dir(const std::string &p) { struct stat st; return !p.empty() && stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
std::string join(const std::string &a, const std::string &b) { return a.empty() ? b : (a.back() == '/' ? a + b : a + "/" + b); }
std::string opt(const char *s) { return s ? std::string(s) : std::string(); }

// choose_existing_file_from_two_files (profile.rs:107-134): explicit path wins, else the DB default
std::string choose(const std::string &a, const std::string &b) { return is_file(a) ? a : (is_file(b) ? b : std::string()); }

struct SpeciesProfileRow { std::string species; double abundance, coverage; };

struct DbHolder {
    pantax_hip_ctx *ctx;
    pantax_hip_db *db = nullptr;
    ~DbHolder() { if (db) pantax_hip_db_free(ctx, db); }
};
struct ReadsHolder {
    pantax_hip_ctx *ctx;
    pantax_hip_reads *rd = nullptr;
    ~ReadsHolder() { if (rd) pantax_hip_reads_free(ctx, rd); }
};

inline double round2(double x) { return std::round(x * 100.0) / 100.0; }
std::string cell(bool has, double v, bool rnd = false) { return has ? fmt_f64(rnd ? round2(v) : v) : std::string(); }

}  // namespace

extern "C" int pantax_hip_profile(pantax_hip_ctx *ctx, const pantax_hip_profiling_config *cfg) {
    if (!ctx || !cfg) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    // ---- check_args_valid (profile.rs:71-199)
    if (!cfg->species && !cfg->strain) return fail(ctx, PANTAX_HIP_E_INVALID, "Please choose profiling level with --species or/and --strain.");
    // one process per GPU: rank r takes the selected species i with i % world_size == r; the two global sums of the strain
    // table (profile.rs:3198, :3243) and the hand-over of the rows go through the caller's all-reduce (RCCL / MPI / ...)
    const int W = cfg->world_size > 1 ? cfg->world_size : 1;
    const int rk = W > 1 ? cfg->rank : 0;
    if (W > 1 && (!cfg->allreduce_sum || rk < 0 || rk >= W))
        return fail(ctx, PANTAX_HIP_E_INVALID, "profile: world_size %d needs 0 <= rank < world_size and an allreduce_sum callback", W);
    auto allreduce = [&](double *buf, uint64_t n) -> int {
        if (W == 1) return 0;
        const int rc = cfg->allreduce_sum(cfg->comm_user, buf, n);
        return rc == 0 ? 0 : fail(ctx, PANTAX_HIP_E_STATE, "profile: the caller's allreduce_sum returned %d", rc);
    };
    const std::string db_dir = opt(cfg->db), wd = opt(cfg->wd);
    std::string out_dir = opt(cfg->output_dir);
    if (out_dir.empty()) out_dir = wd;
    if (!is_dir(db_dir)) return fail(ctx, PANTAX_HIP_E_IO, "Specified PanTax database directory '%s' is not a valid directory path", db_dir.c_str());
    if (!is_dir(wd)) return fail(ctx, PANTAX_HIP_E_IO, "Specified PanTax work directory '%s' is not a valid directory path", wd.c_str());
    if (cfg->sample_nodes < 0) return fail(ctx, PANTAX_HIP_E_INVALID, "profile: --sample %d", cfg->sample_nodes);
    const std::string zip = opt(cfg->zip);
    if (zip == "h5")
        return fail(ctx, PANTAX_HIP_E_LIMIT, "profile: graph container '%s' is not available in this build (the reference gates it behind a cargo feature); use serialize / lz / zstd or GFA", zip.c_str());
    const std::string species_file = join(wd, "species_abundance.txt"), strain_file = join(wd, "strain_abundance.txt");
    const bool species_exists = !cfg->force && is_file(species_file);
    const bool strain_exists = !cfg->force && is_file(strain_file);
    bool full_path = cfg->species && !species_exists;
    bool strain_only = !full_path && cfg->strain && !strain_exists;
    bool strain_done = strain_exists;
    if (W > 1) {   // rank 0 looked at the work directory before anybody wrote to it: every rank follows its decision
        double d[3] = {rk == 0 && full_path ? 1.0 : 0.0, rk == 0 && strain_only ? 1.0 : 0.0, rk == 0 && strain_done ? 1.0 : 0.0};
        PTX_TRY(allreduce(d, 3));
        full_path = d[0] != 0.0; strain_only = d[1] != 0.0; strain_done = d[2] != 0.0;
    }
    if (!full_path && !strain_only) return 0;   // profile.rs:3419-3427: outputs already present
    mkdir(out_dir.c_str(), 0777);

    const std::string gaf_path = opt(cfg->input_aln_file);
    if (!is_file(gaf_path)) return fail(ctx, PANTAX_HIP_E_IO, "Specified GAF mapping file '%s' is not a valid file path", gaf_path.c_str());
    const std::string range_path = choose(opt(cfg->range_file), join(db_dir, "species_range.txt"));
    if (range_path.empty()) return fail(ctx, PANTAX_HIP_E_IO, "Neither species range file '%s' nor '%s' is a valid file path", opt(cfg->range_file).c_str(), join(db_dir, "species_range.txt").c_str());

    // PANTAX_HIP_TRACE=1: wall time of each phase on stderr (the reference logs its phases through env_logger)
    const bool trace = std::getenv("PANTAX_HIP_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pantax_hip_profile] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };

    std::vector<RangeRow> ranges;
    std::string err = read_species_range(range_path, ranges);
    if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
    const uint32_t S = (uint32_t)ranges.size();
    if (S == 0) return fail(ctx, PANTAX_HIP_E_IO, "species range file %s is empty", range_path.c_str());

    // ---- a1: GAF -> packed reads (rcls.rs:119-146)
    MappedFile mf;
    err = mf.open(gaf_path);
    if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
    // tokenised on the device (stage_gaf.hip; host_io.cpp:parse_gaf is its checker): the packed reads stay in HBM,
    // only read_len / mapq / flags / id hashes come back for the report and the duplicate-id rule
    HostReads hr;
    ReadsHolder reads{ctx};
    reads.rd = new pantax_hip_reads();
    PTX_TRY(gaf_tokenize_device(ctx, mf.data, mf.size, hr, reads.rd, mf.fd));
    const uint64_t R = hr.qlen.size();
    lap("ranges + GAF tokenise");

    // ---- a2/a3: binning against ALL species ranges (ranges-only db), counters on device
    std::vector<int64_t> rs(S), re(S);
    for (uint32_t s = 0; s < S; ++s) { rs[s] = ranges[s].start; re[s] = ranges[s].end; }
    DbHolder bin_db{ctx};
    {
        pantax_hip_graphs g{};
        g.n_species = S; g.range_start = rs.data(); g.range_end = re.data();
        PTX_TRY(pantax_hip_db_upload(ctx, &g, &bin_db.db));
    }
    std::vector<int32_t> sp_idx(R);
    std::vector<int64_t> rc(S), bs(S), lm(S), uq(S);
    PTX_TRY(pantax_hip_bin_reads(ctx, bin_db.db, reads.rd, sp_idx.data(), rc.data(), bs.data(), lm.data(), uq.data()));

    lap("bin all species");
    std::vector<SpeciesProfileRow> sp_profile;   // species_taxid, predicted_abundance, predicted_coverage
    if (full_path) {
        // optional binning report: read_id, mapq, species, read_len; no header (profile.rs:3337-3351)
        const std::string report = opt(cfg->out_binning_file);
        if (rk == 0 && !report.empty() && report != "None") {
            std::ofstream f(report);
            if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", report.c_str());
            for (uint64_t r = 0; r < R; ++r) {
                f.write(mf.data + hr.id_span[r].first, hr.id_span[r].second);
                f << '\t';
                if (hr.mapq[r] != 255) f << (int)hr.mapq[r];
                f << '\t' << (sp_idx[r] >= 0 ? ranges[sp_idx[r]].species : std::string("U")) << '\t' << hr.qlen[r] << '\n';
            }
        }
        const std::string len_path = choose(opt(cfg->species_len_file), join(db_dir, "species_genomes_stats.txt"));
        if (len_path.empty()) return fail(ctx, PANTAX_HIP_E_IO, "Neither species length file '%s' nor '%s' is a valid file path", opt(cfg->species_len_file).c_str(), join(db_dir, "species_genomes_stats.txt").c_str());
        std::vector<std::pair<std::string, double>> lens;
        err = read_species_len(len_path, lens);
        if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
        std::unordered_map<std::string, double> len_of(lens.begin(), lens.end());
        std::vector<double> avg(S, 0.0);
        for (uint32_t s = 0; s < S; ++s) { auto it = len_of.find(ranges[s].species); if (it != len_of.end()) avg[s] = it->second; }
        std::vector<uint8_t> keep(S);
        std::vector<double> absolute(S), abundance(S);
        PTX_TRY(pantax_hip_species_profile(ctx, bin_db.db, reads.rd, rc.data(), bs.data(), lm.data(), uq.data(), avg.data(), cfg->filtered,
                                           keep.data(), absolute.data(), abundance.data()));
        for (uint32_t s = 0; s < S; ++s) if (keep[s]) sp_profile.push_back({ranges[s].species, abundance[s], absolute[s]});
        std::stable_sort(sp_profile.begin(), sp_profile.end(), [](const SpeciesProfileRow &a, const SpeciesProfileRow &b) { return a.abundance > b.abundance; });   // :344
        if (rk == 0) {
            std::ofstream f(join(out_dir, "species_abundance.txt"));
            if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", join(out_dir, "species_abundance.txt").c_str());
            f << "species_taxid\tpredicted_abundance\tpredicted_coverage\n";
            for (auto &r : sp_profile) f << r.species << '\t' << fmt_f64(r.abundance) << '\t' << fmt_f64(r.coverage) << '\n';
        }
        if (!cfg->strain || strain_done) return 0;
    } else {
        // strain only (profile.rs:3365-3417): species column comes from the saved binning file (positional join)
        std::string rb = choose(opt(cfg->reads_binning_file), join(wd, "reads_classification.tsv"));   // profile.rs:179-182
        if (rb.empty()) return fail(ctx, PANTAX_HIP_E_IO, "reads binning file '%s' is not a valid file path", join(wd, "reads_classification.tsv").c_str());
        std::unordered_map<std::string, int32_t> idx_of;
        for (uint32_t s = 0; s < S; ++s) idx_of.emplace(ranges[s].species, (int32_t)s);
        std::ifstream f(rb);
        std::string line;
        uint64_t r = 0;
        while (std::getline(f, line)) {
            if (r >= R) return fail(ctx, PANTAX_HIP_E_IO, "%s has more rows than the GAF (%llu)", rb.c_str(), (unsigned long long)R);
            size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1), t3 = t2 == std::string::npos ? t2 : line.find('\t', t2 + 1);
            if (t2 == std::string::npos) return fail(ctx, PANTAX_HIP_E_IO, "malformed row in %s", rb.c_str());
            std::string spn = line.substr(t2 + 1, t3 == std::string::npos ? std::string::npos : t3 - t2 - 1);
            auto it = idx_of.find(spn);
            sp_idx[r++] = it == idx_of.end() ? -1 : it->second;
        }
        if (r != R) return fail(ctx, PANTAX_HIP_E_IO, "%s has %llu rows but the GAF has %llu (the join is positional, profile.rs:3381-3384)", rb.c_str(), (unsigned long long)r, (unsigned long long)R);
        if (!is_file(species_file)) return fail(ctx, PANTAX_HIP_E_IO, "species abundance file '%s' is not a valid file path", species_file.c_str());
        std::ifstream sf(species_file);
        bool header = true;
        while (std::getline(sf, line)) {
            if (header) { header = false; continue; }
            size_t t1 = line.find('\t'), t2 = line.find('\t', t1 + 1);
            if (t1 == std::string::npos || t2 == std::string::npos) continue;
            sp_profile.push_back({line.substr(0, t1), std::stod(line.substr(t1 + 1, t2 - t1 - 1)), std::stod(line.substr(t2 + 1))});
        }
    }

    lap("species table / report");
    // ---- a4: load_species_range (profile.rs:553-656)
    std::set<std::string> ds;
    const std::string ds_s = opt(cfg->designated_species);
    if (!ds_s.empty() && ds_s != "None") {
        std::stringstream ss(ds_s);
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            size_t b = tok.find_first_not_of(" \t"), e = tok.find_last_not_of(" \t");
            if (b != std::string::npos) ds.insert(tok.substr(b, e - b + 1));
        }
    }
    std::unordered_map<std::string, uint32_t> range_idx;
    for (uint32_t s = 0; s < S; ++s) range_idx.emplace(ranges[s].species, s);
    std::vector<uint32_t> sel;          // indices into `ranges`, in species-profile order
    std::vector<double> sel_cov;
    bool any_after_ds = false;
    for (auto &row : ranges) {
        if ((cfg->mode == 0 && row.is_pan != 0) || (cfg->mode == 1 && row.is_pan != 1)) continue;
        if (!ds.empty() && !ds.count(row.species)) continue;
        any_after_ds = true;
    }
    if (!any_after_ds) return 0;        // reference: warn + exit(0) (profile.rs:595-598)
    for (auto &row : sp_profile) {
        if (!(row.abundance > cfg->min_species_abundance)) continue;                 // :602
        auto it = range_idx.find(row.species);
        if (it == range_idx.end()) continue;                                         // inner join :604-605
        const RangeRow &rr = ranges[it->second];
        if ((cfg->mode == 0 && rr.is_pan != 0) || (cfg->mode == 1 && rr.is_pan != 1)) continue;
        if (!ds.empty() && !ds.count(rr.species)) continue;
        sel.push_back(it->second);
        sel_cov.push_back(row.coverage);
    }

    // ---- a5: rows with a null field are dropped; duplicate read ids (profile.rs:361-463)
    std::vector<uint8_t> flags(hr.flags);
    {
        std::unordered_set<uint64_t> seen;
        if (hr.ids_distinct != 1) seen.reserve(R * 2);
        bool unique = true;
        // the device tokenizer has already sorted the id hashes: when no two reads share one, nothing can repeat
        if (hr.ids_distinct != 1)
            for (uint64_t r = 0; r < R && unique; ++r) if (sp_idx[r] >= 0 && !seen.insert(hr.id_hash[r]).second) unique = false;
        if (!unique) {   // process_with_duplicates: keep an id only if all of its (complete) alignments sit in one species
            std::unordered_map<uint64_t, int32_t> first;
            std::unordered_set<uint64_t> mixed;
            for (uint64_t r = 0; r < R; ++r) {
                if (sp_idx[r] < 0 || flags[r]) continue;
                auto ins = first.emplace(hr.id_hash[r], sp_idx[r]);
                if (!ins.second && ins.first->second != sp_idx[r]) mixed.insert(hr.id_hash[r]);
            }
            for (uint64_t r = 0; r < R; ++r) if (sp_idx[r] >= 0 && mixed.count(hr.id_hash[r])) flags[r] |= PANTAX_HIP_READ_DUPDROP;
        }
    }

    lap("select + duplicate ids");
    // ---- a6: graphs of the selected species (optimize_otu file choice, profile.rs:2888-2932)
    const uint32_t Ss = (uint32_t)sel.size();
    std::vector<HostGraph> graphs(Ss);
    std::vector<uint8_t> loaded(Ss, 1);
    std::vector<uint32_t> use;   // this rank's selected species with a loaded graph (indices into sel)
    uint32_t Su = 0;
    std::vector<pantax_hip_hap_metrics> met;
    std::vector<pantax_hip_solve_info> info;
    std::vector<uint64_t> hap_off(1, 0);
    std::vector<std::string> hap_names;
    // which rank takes which selected species: longest-processing-time packing on (reads binned to the species + its graph
    // nodes), heaviest first onto the least loaded rank; every rank computes the same table from the same inputs
    std::vector<int> owner(Ss, 0);
    if (W > 1) {
        std::vector<uint32_t> by_weight(Ss);
        std::vector<double> weight(Ss), load(W, 0.0);
        for (uint32_t i = 0; i < Ss; ++i) {
            by_weight[i] = i;
            weight[i] = (double)rc[sel[i]] * 8.0 + (double)(ranges[sel[i]].end - ranges[sel[i]].start + 1);   // ~8 walk steps per read
        }
        std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
        for (uint32_t i : by_weight) {
            int r = 0;
            for (int q = 1; q < W; ++q) if (load[q] < load[r]) r = q;
            owner[i] = r; load[r] += weight[i];
        }
    }
    // everything a rank does on its own shard; a failure here must not leave the other ranks waiting in the exchange below
    auto shard = [&]() -> int {
    // image_cache >= 1: device-ready images <db>/species_graph_info/<otu>.hipdb (db_image.cpp) stand in for
    // the graph files AND the per-run unique-trio index when every selected species has one that is not older than its
    // source; otherwise the graphs are parsed as usual (and, with image_cache == 2, the images are written afterwards)
    auto source_of = [&](const std::string &otu) {
        const std::string bin = join(join(db_dir, "species_graph_info"), otu + ".bin");
        if (zip == "serialize" && is_file(bin)) return bin;
        if (zip == "lz" && is_file(bin + ".lz4")) return bin + ".lz4";
        if (zip == "zstd" && is_file(bin + ".zst")) return bin + ".zst";
        return join(join(db_dir, "species_gfa"), otu + ".gfa");
    };
    auto image_of = [&](const std::string &otu) { return join(join(db_dir, "species_graph_info"), otu + ".hipdb"); };
    std::vector<std::unique_ptr<SpeciesImage>> images(Ss);
    bool use_images = cfg->image_cache >= 1 && Ss > 0;
    if (use_images) {
        std::vector<uint8_t> ok(Ss, 0);
        parallel_for(Ss, 8, [&](uint64_t i0, uint64_t i1) {
            for (uint64_t i = i0; i < i1; ++i) {
                if (owner[i] != rk) { ok[i] = 1; continue; }                                     // another rank's species
                const std::string &otu = ranges[sel[i]].species;
                const std::string img = image_of(otu);
                if (!is_file(img) || file_mtime(img) < file_mtime(source_of(otu))) continue;
                images[i].reset(new SpeciesImage());
                if (!images[i]->open(img).empty()) continue;                                   // unreadable image: parse instead
                ok[i] = (int64_t)images[i]->V == ranges[sel[i]].end - ranges[sel[i]].start + 1;
            }
        });
        for (uint32_t i = 0; i < Ss; ++i) use_images = use_images && ok[i];
        if (!use_images) for (auto &im : images) im.reset();
    }
    if (!use_images) {   // the graph files are independent: parsed by a few threads; the first problem in species order is reported
        std::vector<std::string> hard(Ss);   // errors that end the run
        parallel_for(Ss, 8, [&](uint64_t i0, uint64_t i1) {
            for (uint64_t i = i0; i < i1; ++i) {
                if (owner[i] != rk) { loaded[i] = 0; continue; }                                 // another rank's species
                const std::string &otu = ranges[sel[i]].species;
                std::string gfa = join(join(db_dir, "species_gfa"), otu + ".gfa");
                std::string bin = join(join(db_dir, "species_graph_info"), otu + ".bin");
                const std::string lz = bin + ".lz4", zst = bin + ".zst";
                std::string e2;
                if (zip == "serialize" && is_file(bin)) e2 = read_graph_bin(bin, graphs[i]);
                else if (zip == "lz" && is_file(lz)) e2 = read_graph_zip(lz, 2, graphs[i]);
                else if (zip == "zstd" && is_file(zst)) e2 = read_graph_zip(zst, 3, graphs[i]);
                else if (is_file(gfa)) e2 = read_gfa(gfa, graphs[i]);
                else { hard[i] = "gfa information file " + gfa + " does not exist. Please check database."; continue; }
                if (!e2.empty()) { loaded[i] = 0; continue; }            // "GFA read error" => species skipped (.ok()?)
                const int64_t nvert = ranges[sel[i]].end - ranges[sel[i]].start + 1;
                if ((int64_t)graphs[i].node_len.size() != nvert)
                    hard[i] = "species " + otu + ": graph has " + std::to_string(graphs[i].node_len.size()) + " nodes but its range spans " + std::to_string((long long)nvert);
            }
        });
        for (uint32_t i = 0; i < Ss; ++i) if (!hard[i].empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", hard[i].c_str());
    }
    lap("graph load");
    for (uint32_t i = 0; i < Ss; ++i) if (loaded[i] && owner[i] == rk) use.push_back(i);
    Su = (uint32_t)use.size();
    info.assign(Su, pantax_hip_solve_info{});
    hap_off.assign(Su + 1, 0);
    if (Su) {
        std::vector<int64_t> g_rs(Su), g_re(Su);
        std::vector<GraphPart> parts(Su);
        std::vector<const SpeciesImage *> imp(Su);
        for (uint32_t k = 0; k < Su; ++k) {   // the parsed graphs go to the device as they are: one part per species
            g_rs[k] = ranges[sel[use[k]]].start; g_re[k] = ranges[sel[use[k]]].end;
            const std::vector<std::string> &names = use_images ? images[use[k]]->hap_names : graphs[use[k]].hap_names;
            if (use_images) imp[k] = images[use[k]].get();
            else { const HostGraph &hg = graphs[use[k]]; parts[k] = GraphPart{hg.node_len.data(), hg.node_len.size(), hg.hap_names.size(), hg.path_off.data(), hg.path_nodes.data()}; }
            hap_names.insert(hap_names.end(), names.begin(), names.end());
            hap_off[k + 1] = hap_names.size();
        }
        DbHolder sdb{ctx};
        if (use_images) PTX_TRY(db_from_images(ctx, Su, imp.data(), g_rs.data(), g_re.data(), &sdb.db));   // trio index included
        else PTX_TRY(db_upload_parts(ctx, Su, g_rs.data(), g_re.data(), parts.data(), &sdb.db));
        lap("db upload");
        // the same resident reads with the strain-level drop flags; species binned against the selected ranges
        // (reads of unselected species fall outside every range => "U" => skipped, as in the reference
        // where only selected species are looked up in the per-species read map, profile.rs:3301-3303)
        if (strain_only) {
            // species from the saved report decide membership: a read whose recorded species differs from
            // where its nodes bin now is dropped for that species
            for (uint64_t r = 0; r < R; ++r) if (sp_idx[r] < 0) flags[r] |= PANTAX_HIP_READ_NULLFIELD;
        }
        PTX_TRY(pantax_hip_reads_set_flags(ctx, reads.rd, flags.data()));
        pantax_hip_reads *const sreads_rd = reads.rd;
        PTX_TRY(pantax_hip_bin_reads(ctx, sdb.db, sreads_rd, nullptr, nullptr, nullptr, nullptr, nullptr));
        uint64_t nU = 0, n_abort = 0;
        PTX_TRY(pantax_hip_trio_index(ctx, sdb.db, &nU));
        PTX_TRY(pantax_hip_node_coverage(ctx, sdb.db, sreads_rd, nullptr, nullptr, nullptr, nullptr, &n_abort));
        pantax_hip_strain_config sc{cfg->unique_trio_nodes_fraction, cfg->unique_trio_nodes_mean_count_f, cfg->single_cov_ratio, cfg->min_depth, cfg->shift, cfg->sample_nodes};
        std::vector<double> cov(Su);
        for (uint32_t k = 0; k < Su; ++k) cov[k] = sel_cov[use[k]];
        met.resize(hap_names.size());
        PTX_TRY(pantax_hip_strain_profile(ctx, sdb.db, &sc, nullptr, cov.data(), met.data(), info.data()));
        lap("strain step");
        if (!use_images && cfg->image_cache == 2) {   // leave images behind for the next run
            for (uint32_t k = 0; k < Su; ++k)
                PTX_TRY(db_save_image(ctx, sdb.db, k, graphs[use[k]].hap_names, image_of(ranges[sel[use[k]]].species)));
            lap("graph images written");
        }
    }
    return 0;
    };   // shard
    const int shard_rc = shard();

    // ---- a15: abundance_est (profile.rs:3091-3289)
    std::vector<GenomeRow> genomes;
    err = read_genomes_info(join(db_dir, "genomes_info.txt"), genomes);   // the reference always reads <db>/genomes_info.txt (:3099)
    if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
    std::unordered_multimap<std::string, size_t> by_hap;
    for (size_t i = 0; i < genomes.size(); ++i) by_hap.emplace(genomes[i].hap_id, i);
    std::vector<uint8_t> reported(Su, 0), pass(hap_names.size() ? hap_names.size() : 1, 0);
    double sum_all = 0.0, sum_pass = 0.0;
    int local_rc = shard_rc;
    if (local_rc == 0) {
        for (uint32_t k = 0; k < Su; ++k) reported[k] = (info[k].status1 == 0 && info[k].status2 == 0) ? 1 : 0;
        if (Su) local_rc = pantax_hip_abundance_filter(Su, hap_off.data(), met.data(), reported.data(), cfg->single_cov_diff, cfg->min_cov, pass.data(), &sum_all, &sum_pass, nullptr, nullptr);
    }
    {   // the one exchange of the strain level: did every rank get through, and the two normalisers
        double ex[3] = {local_rc != 0 ? 1.0 : 0.0, sum_all, sum_pass};
        PTX_TRY(allreduce(ex, 3));
        if (ex[0] != 0.0) return local_rc ? local_rc : fail(ctx, PANTAX_HIP_E_STATE, "profile: another rank failed on its species; no strain table was written");
        sum_all = ex[1]; sum_pass = ex[2];
    }
    // rows keep (species position in the selection, running number) so that any merge reproduces the one-process order
    struct OutRow { double key; uint32_t k, seq; std::string line; };
    auto row_text = [&](uint32_t k, uint64_t h, const GenomeRow *gr, double abund, bool has_abund, bool rnd) {
        const pantax_hip_hap_metrics &m = met[h];
        std::string s = ranges[sel[use[k]]].species;
        s += '\t'; if (gr) s += gr->strain_taxid;
        s += '\t'; if (gr) s += gr->genome_id;
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_SECOND, m.second_sol, rnd);
        s += '\t' + (has_abund ? fmt_f64(abund) : std::string());
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_RATIO, m.path_cov_ratio, rnd);
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_FRACTION, m.unique_trio_nodes_fraction, rnd);
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_FREQ_MEAN, m.frequencies_mean, rnd);
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_FIRST, m.first_sol, rnd);
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_DIVERGENCE, m.divergence, rnd);
        s += '\t' + cell(m.has & PANTAX_HIP_HAS_TOTAL_DIFF, m.total_cov_diff, rnd);
        return s;
    };
    const char *header = "species_taxid\tstrain_taxid\tgenome_ID\tpredicted_coverage\tpredicted_abundance\tpath_base_cov\tunique_trio_fraction\tuniq_trio_cov_mean\tfirst_sol\tstrain_cov_diff\ttotal_cov_diff\n";
    std::vector<OutRow> ori_rows, final_rows;
    for (uint32_t k = 0; k < Su; ++k) {
        if (!reported[k]) continue;
        uint32_t seq = 0;
        for (uint64_t h = hap_off[k]; h < hap_off[k + 1]; ++h) {
            auto range = by_hap.equal_range(hap_names[h]);
            std::vector<const GenomeRow *> grs;
            for (auto it = range.first; it != range.second; ++it) grs.push_back(&genomes[it->second]);
            if (grs.empty()) grs.push_back(nullptr);          // left join keeps the row with null metadata
            const bool hs = met[h].has & PANTAX_HIP_HAS_SECOND;
            for (const GenomeRow *gr : grs) {
                ori_rows.push_back({0.0, use[k], seq, row_text(k, h, gr, hs ? met[h].second_sol / sum_all : 0.0, hs, false)});
                if (pass[h]) final_rows.push_back({met[h].second_sol / sum_pass, use[k], seq, row_text(k, h, gr, met[h].second_sol / sum_pass, true, !cfg->full)});   // :3250-3284
                ++seq;
            }
        }
    }
    if (W > 1) {   // rows of the other ranks reach rank 0 through part files in the work directory (one node, one file system)
        auto part_name = [&](const char *what, int r) { return strain_file + "." + what + ".part" + std::to_string(r); };
        auto write_part = [&](const std::string &path, const std::vector<OutRow> &rows) {
            FILE *f = std::fopen(path.c_str(), "wb");
            if (!f) return false;
            bool ok = true;
            for (const OutRow &r : rows) {
                const uint32_t len = (uint32_t)r.line.size();
                ok = ok && std::fwrite(&r.key, 8, 1, f) == 1 && std::fwrite(&r.k, 4, 1, f) == 1 && std::fwrite(&r.seq, 4, 1, f) == 1 && std::fwrite(&len, 4, 1, f) == 1 &&
                     (len == 0 || std::fwrite(r.line.data(), 1, len, f) == len);
            }
            return std::fclose(f) == 0 && ok;
        };
        const bool wrote = write_part(part_name("ori", rk), ori_rows) && write_part(part_name("final", rk), final_rows);
        double bar[1] = {wrote ? 0.0 : 1.0};
        PTX_TRY(allreduce(bar, 1));                      // every part is on disk (or somebody could not write)
        if (bar[0] != 0.0) return fail(ctx, PANTAX_HIP_E_IO, "profile: a rank could not write its part of the strain table under %s", wd.c_str());
        if (rk != 0) { lap("tables"); return 0; }
        auto read_part = [&](const std::string &path, std::vector<OutRow> &rows) {
            FILE *f = std::fopen(path.c_str(), "rb");
            if (!f) return false;
            for (;;) {
                OutRow r; uint32_t len = 0;
                if (std::fread(&r.key, 8, 1, f) != 1) break;
                if (std::fread(&r.k, 4, 1, f) != 1 || std::fread(&r.seq, 4, 1, f) != 1 || std::fread(&len, 4, 1, f) != 1) { std::fclose(f); return false; }
                r.line.resize(len);
                if (len && std::fread(&r.line[0], 1, len, f) != len) { std::fclose(f); return false; }
                rows.push_back(std::move(r));
            }
            std::fclose(f);
            std::remove(path.c_str());
            return true;
        };
        ori_rows.clear(); final_rows.clear();
        for (int r = 0; r < W; ++r)
            if (!read_part(part_name("ori", r), ori_rows) || !read_part(part_name("final", r), final_rows))
                return fail(ctx, PANTAX_HIP_E_IO, "profile: cannot read the part of rank %d under %s", r, wd.c_str());
        auto by_pos = [](const OutRow &a, const OutRow &b) { return a.k != b.k ? a.k < b.k : a.seq < b.seq; };
        std::sort(ori_rows.begin(), ori_rows.end(), by_pos);
        std::sort(final_rows.begin(), final_rows.end(), by_pos);
    }
    {
        std::ofstream ori("ori_strain_abundance.txt");   // written to the current directory (profile.rs:3217)
        if (ori) { ori << header; for (auto &r : ori_rows) ori << r.line << '\n'; }
    }
    std::stable_sort(final_rows.begin(), final_rows.end(), [](const OutRow &a, const OutRow &b) { return a.key > b.key; });   // :3247-3248
    std::ofstream f(strain_file);
    if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", strain_file.c_str());
    f << header;
    for (auto &r : final_rows) f << r.line << '\n';
    lap("tables");
    return 0;
}
