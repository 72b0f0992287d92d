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
#include "primitives.hpp"

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

// first line start at or after byte `c` of the mapped text (a line starts at 0 or right after a '\n')
static uint64_t line_start_at_or_after(const MappedFile &mf, uint64_t c) {
    if (c == 0) return 0;
    if (c >= mf.size) return mf.size;
    const void *nl = std::memchr(mf.data + (c - 1), '\n', mf.size - (c - 1));
    return nl ? (uint64_t)(static_cast<const char *>(nl) - mf.data) + 1 : mf.size;
}

static int profile_impl(pantax_hip_ctx *ctx, const pantax_hip_profiling_config *cfg) {
    PTX_ENTER(ctx);
    // ---- check_args_valid (profile.rs:71-199)
    if (!cfg->species && !cfg->strain) return fail(ctx, PANTAX_HIP_E_INVALID, "Please choose profiling level with --species or/and --strain.");
    // one process per GPU: the selected species are packed onto the ranks by weight; the two global sums of the strain table
    // (profile.rs:3198, :3243) and the hand-over of the rows go through the caller's all-reduce (RCCL / MPI / ...).  With an
    // alltoallv callback the input is sharded as well: every rank tokenises and bins its byte range of the GAF.
    const int W = cfg->world_size > 1 ? cfg->world_size : 1;
    const int rk = W > 1 ? cfg->rank : 0;
    if (W > 1 && (!cfg->allreduce_sum || rk < 0 || rk >= W))
        return fail(ctx, PANTAX_HIP_E_INVALID, "profile: world_size %d needs 0 <= rank < world_size and an allreduce_sum callback", W);
    // a one-rank world that is given the callbacks still goes through them (like an MPI program started on one rank): the
    // whole multi-rank protocol, sharded ingest included, can be exercised on a single GPU
    const bool use_comm = W > 1 || (cfg->world_size == 1 && cfg->allreduce_sum != nullptr);
    const bool sharded = use_comm && cfg->alltoallv != nullptr;
    if (sharded && W > 64) return fail(ctx, PANTAX_HIP_E_LIMIT, "profile: the sharded ingest routes reads to at most 64 ranks (world_size %d)", W);
    auto allreduce = [&](double *buf, uint64_t n) -> int {
        if (!use_comm) return 0;
        const int rc = cfg->allreduce_sum(cfg->comm_user, buf, n);
        return rc == 0 ? 0 : fail(ctx, PANTAX_HIP_E_STATE, "profile: the caller's allreduce_sum returned %d", rc);
    };
    // every rank-local failure travels in a flag of the next collective: either all ranks go on or all return (nobody is
    // left waiting in an exchange); the failing rank reports its own error, the others E_STATE
    auto others_failed = [&]() { return fail(ctx, PANTAX_HIP_E_STATE, "profile: another rank failed; this rank stopped with it"); };
    auto agree = [&](int local_rc) -> int {
        if (!use_comm) return local_rc;
        double f = local_rc != 0 ? 1.0 : 0.0;
        PTX_TRY(allreduce(&f, 1));
        if (f != 0.0) return local_rc ? local_rc : others_failed();
        return 0;
    };
    const std::string db_dir = opt(cfg->db), wd = opt(cfg->wd);
    std::string out_dir = opt(cfg->output_dir);
    if (out_dir.empty()) out_dir = wd;
    if (!is_dir(db_dir)) return fail(ctx, PANTAX_HIP_E_IO, "Specified PanTax database directory '%s' is not a valid directory path", db_dir.c_str());
    if (!is_dir(wd)) return fail(ctx, PANTAX_HIP_E_IO, "Specified PanTax work directory '%s' is not a valid directory path", wd.c_str());
    if (cfg->sample_nodes < 0) return fail(ctx, PANTAX_HIP_E_INVALID, "profile: --sample %d", cfg->sample_nodes);
    if (cfg->solver_semantics != PANTAX_HIP_SEMANTICS_GUROBI && cfg->solver_semantics != PANTAX_HIP_SEMANTICS_HIGHS)
        return fail(ctx, PANTAX_HIP_E_INVALID, "profile: solver_semantics %d", cfg->solver_semantics);
    if (!(cfg->minimization_min_cov >= 0.0) || !std::isfinite(cfg->minimization_min_cov))
        return fail(ctx, PANTAX_HIP_E_INVALID, "profile: minimization_min_cov %g", cfg->minimization_min_cov);
    const std::string zip = opt(cfg->zip);
    if (zip == "h5")
        return fail(ctx, PANTAX_HIP_E_LIMIT, "profile: graph container '%s' is not available in this build (the reference gates it behind a cargo feature); use serialize / lz / zstd or GFA", zip.c_str());
    const std::string species_file = join(wd, "species_abundance.txt"), strain_file = join(wd, "strain_abundance.txt");
    const bool species_exists = !cfg->force && is_file(species_file);
    const bool strain_exists = !cfg->force && is_file(strain_file);
    bool full_path = cfg->species && !species_exists;
    bool strain_only = !full_path && cfg->strain && !strain_exists;
    bool strain_done = strain_exists;
    if (use_comm) {   // rank 0 looked at the work directory before anybody wrote to it: every rank follows its decision
        double d[3] = {rk == 0 && full_path ? 1.0 : 0.0, rk == 0 && strain_only ? 1.0 : 0.0, rk == 0 && strain_done ? 1.0 : 0.0};
        PTX_TRY(allreduce(d, 3));
        full_path = d[0] != 0.0; strain_only = d[1] != 0.0; strain_done = d[2] != 0.0;
    }
    if (!full_path && !strain_only) return 0;   // profile.rs:3419-3427: outputs already present
    mkdir(out_dir.c_str(), 0777);

    // PANTAX_HIP_TRACE=1: wall time of each phase on stderr (the reference logs its phases through env_logger)
    const bool trace = ctx->cfg.trace;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pantax_hip_profile r%d] %-28s %9.3f ms\n", rk, what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };

    // bytes between the ranks (sharded ingest).  The callback takes host or device pointers (comm_device_buffers); both
    // forms are offered here so that neither the small id exchange nor the read payload is staged more than needed.
    const bool dev_comm = cfg->comm_device_buffers != 0;
    auto a2a = [&](const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off) -> int {
        const int rc = cfg->alltoallv(cfg->comm_user, send, send_off, recv, recv_off);
        return rc == 0 ? 0 : fail(ctx, PANTAX_HIP_E_STATE, "profile: the caller's alltoallv returned %d", rc);
    };
    auto a2a_host = [&](const void *send_h, const uint64_t *send_off, std::vector<uint8_t> &recv_h, const uint64_t *recv_off) -> int {
        recv_h.resize(recv_off[W] ? recv_off[W] : 1);
        if (!dev_comm) return a2a(send_h, send_off, recv_h.data(), recv_off);
        DevBuf<uint8_t> ds, dr;
        PTX_HIP(ctx, ds.alloc(send_off[W] ? send_off[W] : 1)); PTX_HIP(ctx, dr.alloc(recv_off[W] ? recv_off[W] : 1));
        if (send_off[W]) PTX_HIP(ctx, hipMemcpyAsync(ds.p, send_h, send_off[W], hipMemcpyHostToDevice, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PTX_TRY(a2a(ds.p, send_off, dr.p, recv_off));
        if (recv_off[W]) PTX_HIP(ctx, hipMemcpyAsync(recv_h.data(), dr.p, recv_off[W], hipMemcpyDeviceToHost, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    auto a2a_dev = [&](const void *send_d, const uint64_t *send_off, void *recv_d, const uint64_t *recv_off) -> int {
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (dev_comm) return a2a(send_d, send_off, recv_d, recv_off);
        std::vector<uint8_t> hs(send_off[W] ? send_off[W] : 1), hrv(recv_off[W] ? recv_off[W] : 1);
        if (send_off[W]) PTX_HIP(ctx, hipMemcpy(hs.data(), send_d, send_off[W], hipMemcpyDeviceToHost));
        PTX_TRY(a2a(hs.data(), send_off, hrv.data(), recv_off));
        if (recv_off[W]) PTX_TRY(upload_big(ctx, recv_d, hrv.data(), recv_off[W]));
        return 0;
    };
    // who sends how many bytes to whom: every rank fills its row of a W x W matrix, one all-reduce (it also carries the
    // failure flag of the phase before).  -> recv_off [W+1] of this rank
    auto exchange_sizes = [&](const uint64_t *send_off, std::vector<uint64_t> &recv_off, int local_rc) -> int {
        std::vector<double> m((size_t)W * W + 1, 0.0);
        if (local_rc == 0) for (int j = 0; j < W; ++j) m[(size_t)rk * W + j] = (double)(send_off[j + 1] - send_off[j]);
        m[(size_t)W * W] = local_rc != 0 ? 1.0 : 0.0;
        PTX_TRY(allreduce(m.data(), m.size()));
        if (m[(size_t)W * W] != 0.0) return local_rc ? local_rc : others_failed();
        recv_off.assign(W + 1, 0);
        for (int i = 0; i < W; ++i) recv_off[i + 1] = recv_off[i] + (uint64_t)m[(size_t)i * W + rk];
        return 0;
    };

    const std::string report = opt(cfg->out_binning_file);
    const bool want_report = full_path && !report.empty() && report != "None";
    // ---- a1 + a2/a3, rank-local: ranges, GAF (this rank's byte range when sharded) -> packed reads in HBM, binned against
    // ALL species ranges (ranges-only db), counters on the device
    const std::string gaf_path = opt(cfg->input_aln_file);
    std::vector<RangeRow> ranges;
    uint32_t S = 0;
    MappedFile mf;
    HostReads hr;
    ReadsHolder reads{ctx};
    DbHolder bin_db{ctx};
    uint64_t R = 0, text_begin = 0;
    std::vector<int32_t> sp_idx;
    std::vector<int64_t> rc, bs, lm, uq;
    std::vector<int64_t> rs, re;
    auto ingest = [&]() -> int {
        if (!is_file(gaf_path)) return fail(ctx, PANTAX_HIP_E_IO, "Specified GAF mapping file '%s' is not a valid file path", gaf_path.c_str());
        const std::string range_path = choose(opt(cfg->range_file), join(db_dir, "species_range.txt"));
        if (range_path.empty()) return fail(ctx, PANTAX_HIP_E_IO, "Neither species range file '%s' nor '%s' is a valid file path", opt(cfg->range_file).c_str(), join(db_dir, "species_range.txt").c_str());
        std::string err = read_species_range(range_path, ranges);
        if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
        S = (uint32_t)ranges.size();
        if (S == 0) return fail(ctx, PANTAX_HIP_E_IO, "species range file %s is empty", range_path.c_str());
        err = mf.open(gaf_path);
        if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
        // tokenised on the device (stage_gaf.hip; host_io.cpp:parse_gaf is its checker): the packed reads stay in HBM,
        // only read_len / mapq / flags / id hashes come back for the report and the duplicate-id rule
        uint64_t text_end = mf.size;
        if (sharded) {
            text_begin = line_start_at_or_after(mf, mf.size / (uint64_t)W * (uint64_t)rk);
            text_end = rk + 1 == W ? mf.size : line_start_at_or_after(mf, mf.size / (uint64_t)W * (uint64_t)(rk + 1));
        }
        reads.rd = new pantax_hip_reads();
        // The per-read host columns (read_len, mapq, flags, id hash: 14 bytes per read) and the species of every read come back over PCIe only
        // for a caller that uses them: the binning report, the strain-only resume, the sharded ingest -- or, later, the duplicate-id rule when
        // two reads do share an id (host_cols below).  A plain run on distinct ids needs the species COUNTERS and the first rows only.
        // (no locus-grouped copy yet: the species decision needs the counters only -- the plain columns are binned in file order --, and the copy is
        // built while the first graphs travel, on an otherwise idle device; round 5 built it here, 27 ms behind the last byte of the GAF at 1e8 reads)
        PTX_TRY(gaf_tokenize_device(ctx, mf.data + text_begin, text_end - text_begin, hr, reads.rd, mf.fd, text_begin, /*group=*/false, /*want_id_spans=*/want_report,
                                    /*want_host_columns=*/false));
        R = reads.rd->R;
        lap("ranges + GAF tokenise");
        rs.resize(S); re.resize(S);
        for (uint32_t s = 0; s < S; ++s) { rs[s] = ranges[s].start; re[s] = ranges[s].end; }
        pantax_hip_graphs g{};
        g.n_species = S; g.range_start = rs.data(); g.range_end = re.data();
        PTX_TRY(pantax_hip_db_upload(ctx, &g, &bin_db.db));
        rc.resize(S); bs.resize(S); lm.resize(S); uq.resize(S);
        PTX_TRY(pantax_hip_bin_reads(ctx, bin_db.db, reads.rd, nullptr, rc.data(), bs.data(), lm.data(), uq.data()));
        lap("bin all species");
        return 0;
    };
    bool have_cols = false;
    auto host_cols = [&]() -> int {   // the host columns + the species of every read (file order), once
        if (have_cols) return 0;
        PTX_TRY(reads_host_columns(ctx, reads.rd, hr));
        sp_idx.resize(R);
        if (R) {
            PTX_TRY(species_ensure(ctx, reads.rd));
            PTX_TRY(download(ctx, sp_idx.data(), reads.rd->d_species.p, R));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        have_cols = true;
        lap("host columns + species of every read");
        return 0;
    };
    int local_rc = ingest();
    if (local_rc == 0 && (want_report || sharded || strain_only)) local_rc = host_cols();
    // the read lengths of the first (up to 1000) binned rows of the FILE decide the equal-length branch (profile.rs:312-319): they are among the
    // first rows the binning pass hands back with its counters, unless those hold fewer than 1000 binned rows of a longer file
    std::vector<uint32_t> head;
    if (local_rc == 0 && !have_cols) {
        const std::vector<int32_t> &ps = reads.rd->h_pre_species;
        for (size_t r = 0; r < ps.size() && head.size() < 1000; ++r) if (ps[r] >= 0) head.push_back(reads.rd->h_pre_qlen[r]);
        if (head.size() < 1000 && (uint64_t)ps.size() < R) { head.clear(); local_rc = host_cols(); }
    }
    if (local_rc == 0 && have_cols) for (uint64_t r = 0; r < R && head.size() < 1000; ++r) if (sp_idx[r] >= 0) head.push_back(hr.qlen[r]);
    uint64_t read_base = 0, R_all = R;   // this rank's first read in file order; reads of the whole file
    if (sharded) {
        // one all-reduce: {failure flag, S (must agree), reads per rank, the four counters per species, every rank's head}
        const size_t o_rank = 2, o_cnt = o_rank + W, o_head = o_cnt + 4 * (size_t)S;
        double s_chk[2] = {local_rc != 0 ? 1.0 : 0.0, 0.0};
        PTX_TRY(allreduce(s_chk, 1));   // S is only known to ranks that got through: settle the failure first
        if (s_chk[0] != 0.0) return local_rc ? local_rc : others_failed();
        std::vector<double> x(o_head + (size_t)W * 1001, 0.0);
        x[o_rank + rk] = (double)R;
        for (uint32_t s = 0; s < S; ++s) { x[o_cnt + s] = (double)rc[s]; x[o_cnt + S + s] = (double)bs[s]; x[o_cnt + 2 * (size_t)S + s] = (double)lm[s]; x[o_cnt + 3 * (size_t)S + s] = (double)uq[s]; }
        x[o_head + (size_t)rk * 1001] = (double)head.size();
        for (size_t i = 0; i < head.size(); ++i) x[o_head + (size_t)rk * 1001 + 1 + i] = (double)head[i];
        PTX_TRY(allreduce(x.data(), x.size()));
        R_all = 0;
        for (int q = 0; q < W; ++q) { if (q < rk) read_base += (uint64_t)x[o_rank + q]; R_all += (uint64_t)x[o_rank + q]; }
        for (uint32_t s = 0; s < S; ++s) { rc[s] = (int64_t)x[o_cnt + s]; bs[s] = (int64_t)x[o_cnt + S + s]; lm[s] = (int64_t)x[o_cnt + 2 * (size_t)S + s]; uq[s] = (int64_t)x[o_cnt + 3 * (size_t)S + s]; }
        head.clear();
        for (int q = 0; q < W && head.size() < 1000; ++q) {
            const size_t n = (size_t)x[o_head + (size_t)q * 1001];
            for (size_t i = 0; i < n && head.size() < 1000; ++i) head.push_back((uint32_t)x[o_head + (size_t)q * 1001 + 1 + i]);
        }
    } else {
        PTX_TRY(agree(local_rc));
    }
    local_rc = 0;

    std::vector<SpeciesProfileRow> sp_profile;   // species_taxid, predicted_abundance, predicted_coverage
    auto report_part = [&](int r) { return report + ".part" + std::to_string(r); };
    auto species_level = [&]() -> int {
    if (full_path) {
        // optional binning report: read_id, mapq, species, read_len; no header (profile.rs:3337-3351).  Sharded: every rank
        // writes the rows of its byte range to a part file, rank 0 joins them in rank (= file) order below.
        if (want_report && (rk == 0 || sharded)) {
            const std::string path = sharded ? report_part(rk) : report;
            std::ofstream f(path);
            if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", path.c_str());
            for (uint64_t r = 0; r < R; ++r) {
                f.write(mf.data + text_begin + hr.id_span[r].first, hr.id_span[r].second);
                f << '\t';
                if (hr.mapq[r] != 255) f << (int)hr.mapq[r];
                f << '\t' << (sp_idx[r] >= 0 ? ranges[sp_idx[r]].species : std::string("U")) << '\t' << hr.qlen[r] << '\n';
            }
            f.close();
            if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", path.c_str());
        }
        const std::string len_path = choose(opt(cfg->species_len_file), join(db_dir, "species_genomes_stats.txt"));
        if (len_path.empty()) return fail(ctx, PANTAX_HIP_E_IO, "Neither species length file '%s' nor '%s' is a valid file path", opt(cfg->species_len_file).c_str(), join(db_dir, "species_genomes_stats.txt").c_str());
        std::vector<std::pair<std::string, double>> lens;
        std::string err = read_species_len(len_path, lens);
        if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
        std::unordered_map<std::string, double> len_of(lens.begin(), lens.end());
        std::vector<double> avg(S, 0.0);
        for (uint32_t s = 0; s < S; ++s) { auto it = len_of.find(ranges[s].species); if (it != len_of.end()) avg[s] = it->second; }
        std::vector<uint8_t> keep(S);
        std::vector<double> absolute(S), abundance(S);
        species_profile_host(S, head.data(), head.size(), rc.data(), bs.data(), lm.data(), uq.data(), avg.data(), cfg->filtered, keep.data(), absolute.data(), abundance.data());
        for (uint32_t s = 0; s < S; ++s) if (keep[s]) sp_profile.push_back({ranges[s].species, abundance[s], absolute[s]});
        std::stable_sort(sp_profile.begin(), sp_profile.end(), [](const SpeciesProfileRow &a, const SpeciesProfileRow &b) { return a.abundance > b.abundance; });   // :344
        if (rk == 0) {
            std::ofstream f(join(out_dir, "species_abundance.txt"));
            if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", join(out_dir, "species_abundance.txt").c_str());
            f << "species_taxid\tpredicted_abundance\tpredicted_coverage\n";
            for (auto &r : sp_profile) f << r.species << '\t' << fmt_f64(r.abundance) << '\t' << fmt_f64(r.coverage) << '\n';
        }
    } else {
        // strain only (profile.rs:3365-3417): species column comes from the saved binning file (positional join); a rank
        // of the sharded ingest takes the rows of its own reads
        std::string rb = choose(opt(cfg->reads_binning_file), join(wd, "reads_classification.tsv"));   // profile.rs:179-182
        if (rb.empty()) return fail(ctx, PANTAX_HIP_E_IO, "reads binning file '%s' is not a valid file path", join(wd, "reads_classification.tsv").c_str());
        std::unordered_map<std::string, int32_t> idx_of;
        for (uint32_t s = 0; s < S; ++s) idx_of.emplace(ranges[s].species, (int32_t)s);
        std::ifstream f(rb);
        std::string line;
        uint64_t row = 0;
        while (std::getline(f, line)) {
            if (row >= R_all) return fail(ctx, PANTAX_HIP_E_IO, "%s has more rows than the GAF (%llu)", rb.c_str(), (unsigned long long)R_all);
            if (row >= read_base && row < read_base + R) {
                size_t t1 = line.find('\t'), t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1), t3 = t2 == std::string::npos ? t2 : line.find('\t', t2 + 1);
                if (t2 == std::string::npos) return fail(ctx, PANTAX_HIP_E_IO, "malformed row in %s", rb.c_str());
                std::string spn = line.substr(t2 + 1, t3 == std::string::npos ? std::string::npos : t3 - t2 - 1);
                auto it = idx_of.find(spn);
                sp_idx[row - read_base] = it == idx_of.end() ? -1 : it->second;
            }
            ++row;
        }
        if (row != R_all) return fail(ctx, PANTAX_HIP_E_IO, "%s has %llu rows but the GAF has %llu (the join is positional, profile.rs:3381-3384)", rb.c_str(), (unsigned long long)row, (unsigned long long)R_all);
        if (!is_file(species_file)) return fail(ctx, PANTAX_HIP_E_IO, "species abundance file '%s' is not a valid file path", species_file.c_str());
        std::ifstream sf(species_file);
        bool header = true;
        while (std::getline(sf, line)) {
            if (header) { header = false; continue; }
            const size_t t1 = line.find('\t');
            const size_t t2 = t1 == std::string::npos ? t1 : line.find('\t', t1 + 1);
            if (t1 == std::string::npos || t2 == std::string::npos) continue;
            const std::string c1 = line.substr(t1 + 1, t2 - t1 - 1), c2 = line.substr(t2 + 1);
            char *e1 = nullptr, *e2 = nullptr;
            const double v1 = std::strtod(c1.c_str(), &e1), v2 = std::strtod(c2.c_str(), &e2);
            if (e1 == c1.c_str() || e2 == c2.c_str()) return fail(ctx, PANTAX_HIP_E_IO, "malformed row in %s: '%s'", species_file.c_str(), line.c_str());
            sp_profile.push_back({line.substr(0, t1), v1, v2});
        }
    }
    return 0;
    };   // species_level
    local_rc = species_level();
    PTX_TRY(agree(local_rc));   // also the barrier behind the report parts
    if (want_report && sharded && rk == 0) {   // the parts in rank order = file order
        std::ofstream out(report, std::ios::binary);
        bool ok = (bool)out;
        for (int r = 0; r < W && ok; ++r) {
            std::ifstream in(report_part(r), std::ios::binary);
            ok = (bool)in;
            if (ok && in.peek() != std::ifstream::traits_type::eof()) out << in.rdbuf();
            in.close();
            std::remove(report_part(r).c_str());
        }
        out.close();
        if (!ok || !out) local_rc = fail(ctx, PANTAX_HIP_E_IO, "cannot join the parts of %s", report.c_str());
    }
    if (full_path && (!cfg->strain || strain_done)) return agree(local_rc);

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
    if (!any_after_ds) return agree(local_rc);   // reference: warn + exit(0) (profile.rs:595-598); the same decision on every rank
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
    // (the device tokenizer has already compared the id hashes: when no two reads share one -- short reads -- nothing can repeat, the flags the
    // tokenizer left on the device stand as they are and no per-read column visits the host)
    if (local_rc == 0 && !sharded && hr.ids_distinct != 1) local_rc = host_cols();
    PTX_TRY(agree(local_rc));
    std::vector<uint8_t> flags(hr.flags);
    bool flags_dirty = false;
    if (strain_only) { flags_dirty = true; for (uint64_t r = 0; r < R; ++r) if (sp_idx[r] < 0) flags[r] |= PANTAX_HIP_READ_NULLFIELD; }   // "U" in the saved report
    if (!sharded) {
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
            for (uint64_t r = 0; r < R; ++r) if (sp_idx[r] >= 0 && mixed.count(hr.id_hash[r])) { flags[r] |= PANTAX_HIP_READ_DUPDROP; flags_dirty = true; }
        }
    } else {
        // The same rule over N byte ranges: the alignments of one read id may sit in different ranks' slices, so every binned
        // read sends (id hash, species, complete?) to the rank its hash selects; that rank sees ALL records of the id, decides
        // "some id repeats" (the reference's `unique` flag, which covers incomplete rows too) and "this id spans species"
        // (over complete rows); only when some id does repeat, the ids to drop are made known to every rank.
        struct IdRec { uint64_t hash; int32_t sp; uint32_t complete; };
        auto dest_of = [&](uint64_t h) { return (int)(((h * 0x9E3779B97F4A7C15ull) >> 33) % (uint64_t)W); };
        std::vector<uint64_t> send_off(W + 1, 0), recv_off;
        std::vector<IdRec> sendv;
        if (local_rc == 0) {
            std::vector<uint64_t> cur(W, 0);
            for (uint64_t r = 0; r < R; ++r) if (sp_idx[r] >= 0) ++cur[dest_of(hr.id_hash[r])];
            for (int j = 0; j < W; ++j) { send_off[j + 1] = send_off[j] + cur[j] * sizeof(IdRec); cur[j] = send_off[j] / sizeof(IdRec); }
            sendv.resize(send_off[W] / sizeof(IdRec));
            for (uint64_t r = 0; r < R; ++r)
                if (sp_idx[r] >= 0) sendv[cur[dest_of(hr.id_hash[r])]++] = IdRec{hr.id_hash[r], sp_idx[r], flags[r] ? 0u : 1u};
        }
        PTX_TRY(exchange_sizes(send_off.data(), recv_off, local_rc));
        std::vector<uint8_t> recvb;
        PTX_TRY(a2a_host(sendv.data(), send_off.data(), recvb, recv_off.data()));
        std::vector<uint64_t> mixed;
        bool dup_any = false;
        auto decide = [&]() -> int {
            const uint64_t n = recv_off[W] / sizeof(IdRec);
            std::vector<uint64_t> kh(n), kv(n);
            const IdRec *in = reinterpret_cast<const IdRec *>(recvb.data());
            for (uint64_t i = 0; i < n; ++i) { kh[i] = in[i].hash; kv[i] = ((uint64_t)in[i].complete << 32) | (uint32_t)in[i].sp; }
            if (n > 65536) {   // by hash on the device (stable LSD radix sort, the payload word rides along)
                DevBuf<uint64_t> a0, a1, b0, b1;
                DevBuf<uint32_t> table, tmp;
                PTX_TRY(upload(ctx, a0, kh.data(), n)); PTX_TRY(upload(ctx, a1, kv.data(), n));
                PTX_HIP(ctx, b0.alloc(n)); PTX_HIP(ctx, b1.alloc(n)); PTX_HIP(ctx, table.alloc(sort_table_elems(n))); PTX_HIP(ctx, tmp.alloc(16));
                SortBufs A, B;
                A.nw = B.nw = 2; A.k[0] = a0.p; A.k[1] = a1.p; B.k[0] = b0.p; B.k[1] = b1.p;
                std::vector<SortPass> passes;
                add_passes(passes, 0, 0, 64);
                bool in_b = false;
                PTX_TRY(radix_sort(ctx, A, B, n, passes.data(), (int)passes.size(), table.p, tmp.p, &in_b, nullptr));
                PTX_TRY(download(ctx, kh.data(), in_b ? b0.p : a0.p, n)); PTX_TRY(download(ctx, kv.data(), in_b ? b1.p : a1.p, n));
                PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            } else {
                std::vector<uint32_t> ord(n);
                for (uint64_t i = 0; i < n; ++i) ord[i] = (uint32_t)i;
                std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return kh[a] < kh[b]; });
                std::vector<uint64_t> h2(n), v2(n);
                for (uint64_t i = 0; i < n; ++i) { h2[i] = kh[ord[i]]; v2[i] = kv[ord[i]]; }
                kh.swap(h2); kv.swap(v2);
            }
            for (uint64_t i = 0; i < n;) {
                uint64_t j = i;
                int64_t sp0 = -1;
                bool mix = false;
                for (; j < n && kh[j] == kh[i]; ++j) {
                    if (!(kv[j] >> 32)) continue;                       // incomplete rows take no part in the species set
                    const int64_t spj = (int64_t)(uint32_t)kv[j];
                    if (sp0 < 0) sp0 = spj; else if (spj != sp0) mix = true;
                }
                if (j - i > 1) dup_any = true;
                if (mix) mixed.push_back(kh[i]);
                i = j;
            }
            return 0;
        };
        local_rc = decide();
        std::vector<double> y(2 + (size_t)W, 0.0);
        y[0] = local_rc != 0 ? 1.0 : 0.0; y[1] = dup_any ? 1.0 : 0.0; y[2 + rk] = (double)mixed.size();
        PTX_TRY(allreduce(y.data(), y.size()));
        if (y[0] != 0.0) return local_rc ? local_rc : others_failed();
        uint64_t n_mixed_all = 0;
        for (int q = 0; q < W; ++q) n_mixed_all += (uint64_t)y[2 + q];
        if (y[1] != 0.0 && n_mixed_all) {   // process_with_duplicates: ids whose complete alignments span species are dropped everywhere
            std::vector<uint64_t> so2(W + 1, 0), ro2(W + 1, 0), mine((size_t)W * mixed.size());
            for (int j = 0; j < W; ++j) {
                so2[j + 1] = so2[j] + mixed.size() * 8; ro2[j + 1] = ro2[j] + (uint64_t)y[2 + j] * 8;
                std::copy(mixed.begin(), mixed.end(), mine.begin() + (size_t)j * mixed.size());
            }
            std::vector<uint8_t> allb;
            PTX_TRY(a2a_host(mine.data(), so2.data(), allb, ro2.data()));
            const uint64_t *am = reinterpret_cast<const uint64_t *>(allb.data());
            std::unordered_set<uint64_t> drop(am, am + n_mixed_all);
            for (uint64_t r = 0; r < R; ++r) if (sp_idx[r] >= 0 && drop.count(hr.id_hash[r])) flags[r] |= PANTAX_HIP_READ_DUPDROP;
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
    // ---- the packed records of this rank's slice travel to the rank that owns their species; what arrives becomes
    // this rank's resident reads (one-process read order restricted to its species).  Dropped rows, "U" reads and reads of
    // unselected species stay behind -- none of them reaches get_node_abundances in the reference either.
    if (sharded) {
        Route rt;
        std::vector<uint64_t> send_off(W + 1, 0);
        auto pack = [&]() -> int {
            std::vector<int32_t> owner_all(S, -1);
            for (uint32_t i = 0; i < Ss; ++i) owner_all[sel[i]] = owner[i];
            if (R) PTX_TRY(upload(ctx, reads.rd->d_flags, flags.data(), R));
            reads.rd->has_flags = R != 0;
            reads.rd->g_flags_valid = false;
            PTX_TRY(route_pack(ctx, bin_db.db, reads.rd, owner_all.data(), W, rt));
            for (int j = 0; j <= W; ++j) send_off[j] = rt.word_off[j] * 4;
            return 0;
        };
        if (local_rc == 0) local_rc = pack();
        // {failure flag, bytes, reads, steps} of every (source, owner) pair in one all-reduce
        std::vector<double> m(3 * (size_t)W * W + 1, 0.0);
        if (local_rc == 0)
            for (int j = 0; j < W; ++j) {
                m[(size_t)rk * W + j] = (double)(send_off[j + 1] - send_off[j]);
                m[(size_t)W * W + (size_t)rk * W + j] = (double)rt.n_reads[j];
                m[2 * (size_t)W * W + (size_t)rk * W + j] = (double)rt.n_steps[j];
            }
        m[3 * (size_t)W * W] = local_rc != 0 ? 1.0 : 0.0;
        PTX_TRY(allreduce(m.data(), m.size()));
        if (m[3 * (size_t)W * W] != 0.0) return local_rc ? local_rc : others_failed();
        std::vector<uint64_t> recv_off(W + 1, 0), nr_from(W), nt_from(W);
        for (int i = 0; i < W; ++i) {
            recv_off[i + 1] = recv_off[i] + (uint64_t)m[(size_t)i * W + rk];
            nr_from[i] = (uint64_t)m[(size_t)W * W + (size_t)i * W + rk];
            nt_from[i] = (uint64_t)m[2 * (size_t)W * W + (size_t)i * W + rk];
        }
        DevBuf<uint32_t> d_recv;
        PTX_HIP(ctx, d_recv.alloc(recv_off[W] / 4 + 1));
        PTX_TRY(a2a_dev(rt.d_send.p, send_off.data(), d_recv.p, recv_off.data()));
        auto unpack = [&]() -> int {
            std::unique_ptr<pantax_hip_reads> routed(new pantax_hip_reads());
            PTX_TRY(reads_from_routed(ctx, d_recv.p, W, nr_from.data(), nt_from.data(), true, routed.get()));
            pantax_hip_reads_free(ctx, reads.rd);
            reads.rd = routed.release();
            return 0;
        };
        local_rc = unpack();   // carried by the {failure flag, sums} all-reduce below
        lap("route reads to owners");
    }
    // everything a rank does on its own shard; a failure here must not leave the other ranks waiting in the exchange below
    auto shard = [&]() -> int {
    // Where every selected species' graph comes from (optimize_otu's file choice, profile.rs:2888-2932), decided species by species on
    // a few dozen threads that read HEADERS only:
    //   image_cache >= 1 and a device-ready image <db>/species_graph_info/<otu>.hipdb that is not older than its source (SURVEY 8f-2,
    //     db_image.cpp): the arrays stream from the image;
    //   zip "serialize" and <otu>.bin: the arrays stream from the bincode file itself, 64-bit values narrowed on their way into the
    //     pinned ring (scan_graph_bin finds them with a dozen small reads; round 4 parsed every file into host vectors on 8 threads);
    //   "lz" / "zstd" containers and GFA text: decoded / parsed into host memory first, then the same pipeline.
    // With image_cache == 2 the species that did not come from an image leave one behind after the run.
    auto source_of = [&](const std::string &otu) {
        const std::string bin = join(join(db_dir, "species_graph_info"), otu + ".bin");
        if (zip == "serialize" && is_file(bin)) return bin;
        if (zip == "lz" && is_file(bin + ".lz4")) return bin + ".lz4";
        if (zip == "zstd" && is_file(bin + ".zst")) return bin + ".zst";
        return join(join(db_dir, "species_gfa"), otu + ".gfa");
    };
    auto image_of = [&](const std::string &otu) { return join(join(db_dir, "species_graph_info"), otu + ".hipdb"); };
    struct Source { int kind = 0; /* 0 none, 1 image, 2 streamed .bin, 3 host graph */ SpeciesImage img; BinIndex bin; std::string bin_path; std::vector<uint64_t> path_off; };
    std::vector<Source> src(Ss);
    {
        std::vector<std::string> hard(Ss);   // errors that end the run; the first one in species order is reported
        const int n_thr = (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency() / (unsigned)std::max(1, W)));
        parallel_for(Ss, n_thr, [&](uint64_t i0, uint64_t i1) {
            for (uint64_t i = i0; i < i1; ++i) {
                if (owner[i] != rk) { loaded[i] = 0; continue; }                                 // another rank's species
                const std::string &otu = ranges[sel[i]].species;
                const int64_t nvert = ranges[sel[i]].end - ranges[sel[i]].start + 1;
                Source &sc = src[i];
                if (cfg->image_cache >= 1) {
                    const std::string img = image_of(otu);
                    if (is_file(img) && file_mtime(img) >= file_mtime(source_of(otu)) && sc.img.open(img).empty() && (int64_t)sc.img.V == nvert) { sc.kind = 1; continue; }
                }
                const std::string gfa = join(join(db_dir, "species_gfa"), otu + ".gfa");
                const std::string bin = join(join(db_dir, "species_graph_info"), otu + ".bin");
                const std::string lz = bin + ".lz4", zst = bin + ".zst";
                std::string e2;
                uint64_t n_nodes = 0;
                if (zip == "serialize" && is_file(bin)) {
                    e2 = scan_graph_bin(bin, sc.bin);
                    if (e2.empty() && sc.bin.names_ascending) {
                        sc.kind = 2; sc.bin_path = bin; n_nodes = sc.bin.V;
                        sc.path_off.assign(sc.bin.walk_len.size() + 1, 0);
                        for (size_t h = 0; h < sc.bin.walk_len.size(); ++h) sc.path_off[h + 1] = sc.path_off[h] + sc.bin.walk_len[h];
                    } else if (e2.empty()) { e2 = read_graph_bin(bin, graphs[i]); sc.kind = 3; n_nodes = graphs[i].node_len.size(); }   // keys out of order: the general parser sorts them
                } else if (zip == "lz" && is_file(lz)) { e2 = read_graph_zip(lz, 2, graphs[i]); sc.kind = 3; n_nodes = graphs[i].node_len.size(); }
                else if (zip == "zstd" && is_file(zst)) { e2 = read_graph_zip(zst, 3, graphs[i]); sc.kind = 3; n_nodes = graphs[i].node_len.size(); }
                else if (is_file(gfa)) { e2 = read_gfa(gfa, graphs[i]); sc.kind = 3; n_nodes = graphs[i].node_len.size(); }
                else { hard[i] = "gfa information file " + gfa + " does not exist. Please check database."; continue; }
                if (!e2.empty()) { loaded[i] = 0; sc.kind = 0; continue; }            // "GFA read error" => species skipped (.ok()?)
                if ((int64_t)n_nodes != nvert)
                    hard[i] = "species " + otu + ": graph has " + std::to_string(n_nodes) + " nodes but its range spans " + std::to_string((long long)nvert);
            }
        });
        for (uint32_t i = 0; i < Ss; ++i) if (!hard[i].empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", hard[i].c_str());
    }
    lap("graph headers");
    for (uint32_t i = 0; i < Ss; ++i) if (loaded[i] && owner[i] == rk) use.push_back(i);
    Su = (uint32_t)use.size();
    info.assign(Su, pantax_hip_solve_info{});
    hap_off.assign(Su + 1, 0);
    if (Su) {
        std::vector<int64_t> g_rs(Su), g_re(Su);
        std::vector<GraphPart> parts(Su);
        std::vector<std::string> files(Su);
        for (uint32_t k = 0; k < Su; ++k) {   // one part per species: where its two arrays lie
            const uint32_t i = use[k];
            g_rs[k] = ranges[sel[i]].start; g_re[k] = ranges[sel[i]].end;
            const Source &sc = src[i];
            GraphPart &pt = parts[k];
            const std::vector<std::string> *names = nullptr;
            if (sc.kind == 1) {
                files[k] = sc.img.path; names = &sc.img.hap_names;
                sc.img.fill_part(pt, (int32_t)k);
            } else if (sc.kind == 2) {
                files[k] = sc.bin_path; names = &sc.bin.hap_names;
                pt.n_nodes = sc.bin.V; pt.n_haps = sc.bin.hap_names.size(); pt.path_off = sc.path_off.data();
                pt.len_seg.file = (int32_t)k; pt.len_seg.file_off = sc.bin.off_node_len; pt.len_seg.out_bytes = 4 * sc.bin.V; pt.len_seg.narrow = true;
                for (size_t h = 0; h < sc.bin.walk_len.size(); ++h) {
                    UploadSeg w; w.file = (int32_t)k; w.file_off = sc.bin.walk_off[h]; w.out_bytes = 4 * sc.bin.walk_len[h]; w.narrow = true;
                    pt.walk_segs.push_back(w);
                }
            } else {
                const HostGraph &hg = graphs[i];
                names = &hg.hap_names;
                pt.n_nodes = hg.node_len.size(); pt.n_haps = hg.hap_names.size(); pt.path_off = hg.path_off.data();
                pt.len_seg.src = hg.node_len.data(); pt.len_seg.out_bytes = 4 * hg.node_len.size(); pt.len_seg.narrow = true;
                UploadSeg w; w.src = hg.path_nodes.data(); w.out_bytes = 4 * hg.path_nodes.size();
                pt.walk_segs.push_back(w);
            }
            hap_names.insert(hap_names.end(), names->begin(), names->end());
            hap_off[k + 1] = hap_names.size();
        }
        met.resize(hap_names.size());
        std::vector<double> cov(Su);
        for (uint32_t k = 0; k < Su; ++k) cov[k] = sel_cov[use[k]];
        // A resident db addresses its path steps with 32 bits.  A selection of more than that (BASELINE configs[4] on one GPU: 1.1e10) is cut into
        // contiguous groups of species under the limit, and the groups go through the device ONE AFTER THE OTHER -- db upload, binning of the same
        // resident reads (the reads of the other groups' species fall outside every range: "U"), index, coverage, strain step; species are
        // independent from a4 on (profile.rs:3297-3319), their rows meet in the table code below exactly as those of one db would.
        // (3e9, not 2^32: the visit table of the index holds a slot per interior path step PLUS pads -- a fifth more at fifty strains per species, where
        // every 64-slot group holds one 50-visit node -- and its slots are 32-bit too; beyond them the whole db falls back to the node-block kernel,
        // correct but slower.)  The groups are balanced: ceil(total / limit) of them, each filled up to total / groups.
        const uint64_t steps_max = ctx->cfg.db_path_steps_max ? ctx->cfg.db_path_steps_max : 3000000000ull;
        uint64_t steps_total = 0;
        for (uint32_t k = 0; k < Su; ++k) steps_total += parts[k].path_off[parts[k].n_haps] - parts[k].path_off[0];
        // Round 6: the groups are also what lets the graphs TRAVEL beside the work on them -- group g + 1 goes from its files to HBM on a loader thread and
        // a copy stream of its own (db_upload_arrays) while this thread builds group g's tables and runs its index, coverage and strain step.  A selection
        // of 2e8 path steps and more is therefore cut into four groups even when one db could hold it (option db_groups: 1 = one db, n = that many).
        uint64_t n_groups = std::max<uint64_t>(1, (steps_total + steps_max - 1) / steps_max);
        if (ctx->cfg.db_groups > 0) n_groups = std::max<uint64_t>(n_groups, (uint64_t)ctx->cfg.db_groups);
        else if (steps_total >= 200000000ull && Su >= 8) n_groups = std::max<uint64_t>(n_groups, 4);
        n_groups = std::min<uint64_t>(n_groups, Su);
        const uint64_t steps_target = std::min<uint64_t>(steps_max, (steps_total + n_groups - 1) / n_groups);
        struct Group { uint32_t k0, k1; std::vector<GraphPart> gparts; };
        std::vector<Group> groups;
        uint64_t cum = 0;                                  // path steps of the groups cut so far
        for (uint32_t k0 = 0; k0 < Su;) {
            uint32_t k1 = k0;
            uint64_t steps = 0, nodes = 0;
            // the group ends where the running total comes closest to its share of the whole ((g + 1) / n of the steps): even groups, and the last of the n
            // takes whatever is left -- no small one behind it (the hard limits still cut: 32-bit path steps and node indices of one db)
            const uint64_t boundary = groups.size() + 1 >= n_groups ? ~0ull : (uint64_t)((double)steps_total * (double)(groups.size() + 1) / (double)n_groups);
            (void)steps_target;
            while (k1 < Su) {
                const uint64_t ps = parts[k1].path_off[parts[k1].n_haps] - parts[k1].path_off[0];
                if (k1 > k0 && (steps + ps > steps_max || nodes + parts[k1].n_nodes > 0xF0000000ull || (boundary != ~0ull && cum + steps + ps / 2 > boundary))) break;
                steps += ps; nodes += parts[k1].n_nodes; ++k1;
            }
            cum += steps;
            // the file indices of a group's segments are relative to the group's file list
            Group g{k0, k1, std::vector<GraphPart>(parts.begin() + k0, parts.begin() + k1)};
            if (k0)
                for (GraphPart &pt : g.gparts) {
                    if (pt.len_seg.file >= 0) pt.len_seg.file -= (int32_t)k0;
                    for (UploadSeg &w : pt.walk_segs) if (w.file >= 0) w.file -= (int32_t)k0;
                    for (UploadSeg *w : {&pt.pk.first_seg, &pt.pk.off_seg, &pt.pk.payload_seg}) if (w->file >= 0) w->file -= (int32_t)k0;
                }
            groups.push_back(std::move(g));
            k0 = k1;
        }
        const bool piped = groups.size() > 1;
        if (piped && !ctx->stream_up) PTX_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_up, hipStreamNonBlocking));
        // the loader: ONE group in flight.  begin() on this thread (small uploads through the ctx's staging), the arrays on the loader thread.
        struct Loader {
            pantax_hip_ctx *ctx;
            std::thread th;
            DbHolder db;
            int rc = 0;
            std::string err;
            double ms = 0;
            explicit Loader(pantax_hip_ctx *c) : ctx(c), db{c} {}
            void join() { if (th.joinable()) th.join(); }
            ~Loader() { join(); }
        };
        auto start_load = [&](const Group &g, Loader &L) -> int {
            const uint32_t Sg = g.k1 - g.k0;
            PTX_TRY(db_upload_begin(ctx, Sg, g_rs.data() + g.k0, g_re.data() + g.k0, g.gparts.data(), &L.db.db));
            const GraphPart *gp = g.gparts.data();
            const std::string *gf = files.data() + g.k0;
            pantax_hip_db *dbp = L.db.db;
            if (!piped) {
                const auto t0 = std::chrono::steady_clock::now();
                L.rc = db_upload_arrays(ctx, dbp, gp, gf, nullptr);
                L.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return L.rc;
            }
            L.th = std::thread([this_ctx = ctx, dbp, gp, gf, &L] {
                const auto t0 = std::chrono::steady_clock::now();
                if (hipSetDevice(this_ctx->device) != hipSuccess) L.rc = fail(this_ctx, PANTAX_HIP_E_HIP, "hipSetDevice on the graph loader thread");
                else L.rc = db_upload_arrays(this_ctx, dbp, gp, gf, this_ctx->stream_up);
                if (L.rc) L.err = pantax_hip_last_error(this_ctx);     // this thread's message: handed to the thread that reports
                L.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            });
            return 0;
        };
        bool flags_set = false;
        std::unique_ptr<Loader> cur(new Loader(ctx)), next;
        PTX_TRY(start_load(groups[0], *cur));
        if (!sharded) { PTX_TRY(reads_group(ctx, reads.rd)); lap("locus-grouped copy of the reads"); }   // (sharded: reads_from_routed grouped what arrived)
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const uint32_t k0 = groups[gi].k0, k1 = groups[gi].k1, Sg = k1 - k0;
            cur->join();
            if (cur->rc) return piped ? fail(ctx, cur->rc, "%s", cur->err.c_str()) : cur->rc;
            if (ctx->cfg.trace) std::fprintf(stderr, "[db_upload]            %-28s %9.3f ms%s\n", "graph arrays -> HBM", cur->ms, piped ? " (loader thread, beside the group before)" : "");
            if (gi + 1 < groups.size()) { next.reset(new Loader(ctx)); PTX_TRY(start_load(groups[gi + 1], *next)); }
            DbHolder &sdb = cur->db;
            PTX_TRY(db_upload_finish(ctx, sdb.db));
            lap(Sg == Su ? "db upload" : "db upload (a group of the species)");
            // the same resident reads with the strain-level drop flags; species binned against the selected ranges
            // (reads of unselected species fall outside every range => "U" => skipped, as in the reference
            // where only selected species are looked up in the per-species read map, profile.rs:3301-3303)
            // (strain only: species from the saved report decide membership -- rows it calls "U" carry a drop flag, see a5 above)
            if (!sharded && flags_dirty && !flags_set) { PTX_TRY(pantax_hip_reads_set_flags(ctx, reads.rd, flags.data())); flags_set = true; }   // sharded: flagged rows were not routed; else the tokenizer's flags stand
            pantax_hip_reads *const sreads_rd = reads.rd;
            PTX_TRY(pantax_hip_bin_reads(ctx, sdb.db, sreads_rd, nullptr, nullptr, nullptr, nullptr, nullptr));
            lap("  flags + bin selected");
            uint64_t nU = 0, n_abort = 0;
            PTX_TRY(pantax_hip_trio_index(ctx, sdb.db, &nU));
            lap("  trio index");
            PTX_TRY(pantax_hip_node_coverage(ctx, sdb.db, sreads_rd, nullptr, nullptr, nullptr, nullptr, &n_abort));
            lap("  node coverage");
            // --sample_test: 500 rows whatever --sample says (profile.rs:1387-1393)
            pantax_hip_strain_config sc{cfg->unique_trio_nodes_fraction, cfg->unique_trio_nodes_mean_count_f, cfg->single_cov_ratio, cfg->min_depth, cfg->shift,
                                        cfg->sample_test ? 500 : cfg->sample_nodes, cfg->solver_semantics};
            PTX_TRY(pantax_hip_strain_profile(ctx, sdb.db, &sc, nullptr, cov.data() + k0, met.data() + hap_off[k0], info.data() + k0));
            lap("strain step");
            if (cfg->image_cache == 2) {   // leave images behind for the next run
                for (uint32_t k = k0; k < k1; ++k)
                    if (src[use[k]].kind != 1) {
                        const std::vector<std::string> names(hap_names.begin() + (ptrdiff_t)hap_off[k], hap_names.begin() + (ptrdiff_t)hap_off[k + 1]);
                        PTX_TRY(db_save_image(ctx, sdb.db, k - k0, names, image_of(ranges[sel[use[k]]].species)));
                    }
                lap("graph images written");
            }
            cur = std::move(next);
        }
    }
    return 0;
    };   // shard
    const int shard_rc = local_rc ? local_rc : shard();

    // ---- a15: abundance_est (profile.rs:3091-3289)
    std::vector<GenomeRow> genomes;
    local_rc = shard_rc;
    if (local_rc == 0) {
        const std::string err = read_genomes_info(join(db_dir, "genomes_info.txt"), genomes);   // the reference always reads <db>/genomes_info.txt (:3099)
        if (!err.empty()) local_rc = fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
    }
    std::unordered_multimap<std::string, size_t> by_hap;
    for (size_t i = 0; i < genomes.size(); ++i) by_hap.emplace(genomes[i].hap_id, i);
    std::vector<uint8_t> reported(Su, 0), pass(hap_names.size() ? hap_names.size() : 1, 0);
    double sum_all = 0.0, sum_pass = 0.0;
    if (local_rc == 0) {
        for (uint32_t k = 0; k < Su; ++k) {
            reported[k] = (info[k].status1 == 0 && info[k].status2 == 0) ? 1 : 0;
            // (a limit of the solver's tables -- none is tied to the number of candidate strains -- drops the species like a failed
            // solve in the reference, profile.rs:2999-3003: say so)
            if (info[k].status1 == PANTAX_HIP_E_LIMIT || info[k].status2 == PANTAX_HIP_E_LIMIT)
                std::fprintf(stderr, "[pantax_hip_profile] warning: species %s (%d candidate strains after the first filter) exceeds a table of this "
                                     "build's LP solver; it is left out of strain_abundance.txt\n",
                             ranges[sel[use[k]]].species.c_str(), info[k].n_candidates);
        }
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
    if (use_comm) {   // rows of the other ranks reach rank 0 through part files in the work directory (one node, one file system)
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

// nothing throws across the boundary: an allocation failure or a parser exception becomes a status + message
extern "C" int pantax_hip_profile(pantax_hip_ctx *ctx, const pantax_hip_profiling_config *cfg) {
    if (!ctx || !cfg) return PANTAX_HIP_E_INVALID;
    try {
        return profile_impl(ctx, cfg);
    } catch (const std::bad_alloc &) {
        return fail(ctx, PANTAX_HIP_E_LIMIT, "profile: out of host memory");
    } catch (const std::exception &e) {
        return fail(ctx, PANTAX_HIP_E_STATE, "profile: %s", e.what());
    } catch (...) {
        return fail(ctx, PANTAX_HIP_E_STATE, "profile: unknown exception");
    }
}
