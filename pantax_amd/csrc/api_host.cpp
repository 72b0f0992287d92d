// api_host.cpp -- scalar host finishing of a3 (species_profiling, profile.rs:299-349) and a15
// (abundance_est filters, profile.rs:3219-3245).  No per-read or per-node work happens here.
#include <algorithm>
#include <cmath>
#include <vector>
#include "common.hpp"

using namespace ptx;

extern "C" {

int pantax_hip_species_profile(pantax_hip_ctx *ctx, const pantax_hip_db *db, pantax_hip_reads *reads, const int64_t *read_count,
                               const int64_t *base_sum, const int64_t *less_multi, const int64_t *uniq_count, const double *avg_len,
                               int filtered, uint8_t *keep_out, double *absolute_out, double *abundance_out) {
    if (!ctx || !db || !reads || !read_count || !base_sum || !less_multi || !uniq_count || !avg_len || !keep_out || !absolute_out || !abundance_out)
        return PANTAX_HIP_E_INVALID;
    if (!reads->binned) return fail(ctx, PANTAX_HIP_E_STATE, "species_profile: call pantax_hip_bin_reads first");
    PTX_ENTER(ctx);
    // profile.rs:312-319: distinct read_len among the first 1000 rows of the frame without "U" reads
    std::vector<uint32_t> head;
    auto scan = [&](const int32_t *sp, const uint32_t *ql, uint64_t n) {
        for (uint64_t i = 0; i < n && head.size() < 1000; ++i) if (sp[i] >= 0) head.push_back(ql[i]);
    };
    uint64_t off = reads->h_pre_species.size();
    scan(reads->h_pre_species.data(), reads->h_pre_qlen.data(), off);   // head fetched together with the counters
    uint64_t CH = 16384;   // rarely needed: fewer than 1000 binned reads among the first 2048 rows
    std::vector<int32_t> sp;
    std::vector<uint32_t> ql;
    for (; off < reads->R && head.size() < 1000; off += CH, CH = std::min<uint64_t>(CH * 8, 1 << 20)) {
        uint64_t n = std::min<uint64_t>(CH, reads->R - off);
        sp.resize(n); ql.resize(n);
        PTX_TRY(species_ensure(ctx, reads));
        PTX_TRY(download(ctx, sp.data(), reads->d_species.p + off, n));
        PTX_TRY(download(ctx, ql.data(), reads->d_qlen.p + off, n));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        scan(sp.data(), ql.data(), n);
    }
    species_profile_host(db->S, head.data(), head.size(), read_count, base_sum, less_multi, uniq_count, avg_len, filtered, keep_out, absolute_out, abundance_out);
    return 0;
}

}  // extern "C"

// the scalar part of species_profiling (profile.rs:299-349) given the read lengths of the first (up to 1000) binned rows
void ptx::species_profile_host(uint32_t S, const uint32_t *head_qlen, size_t n_head, const int64_t *read_count, const int64_t *base_sum,
                               const int64_t *less_multi, const int64_t *uniq_count, const double *avg_len, int filtered, uint8_t *keep_out,
                               double *absolute_out, double *abundance_out) {
    int64_t first_len = n_head ? (int64_t)head_qlen[0] : -1;
    bool equal = n_head != 0;
    for (size_t i = 1; i < n_head && i < 1000; ++i) if ((int64_t)head_qlen[i] != first_len) equal = false;   // :312-319
    double total = 0.0;
    for (uint32_t s = 0; s < S; ++s) {
        keep_out[s] = 0; absolute_out[s] = 0.0; abundance_out[s] = 0.0;
        if (read_count[s] == 0) continue;
        if (filtered) {   // profile.rs:224-245 (the inner join drops species without any MAPQ 3..60 read)
            if (less_multi[s] == 0) continue;
            if (!(uniq_count[s] > 0 && (double)less_multi[s] > (double)read_count[s] / 10.0)) continue;
        }
        if (!(avg_len[s] > 0.0)) continue;
        int64_t base_count = equal ? read_count[s] * first_len : base_sum[s];   // :214/:246 vs :259/:266
        keep_out[s] = 1;
        absolute_out[s] = (double)base_count / avg_len[s];                       // :336
        total += absolute_out[s];
    }
    for (uint32_t s = 0; s < S; ++s) if (keep_out[s]) abundance_out[s] = absolute_out[s] / total;   // :341
}

extern "C" {

int pantax_hip_db_reset(pantax_hip_ctx *ctx, pantax_hip_db *db) {
    if (!ctx || !db) return PANTAX_HIP_E_INVALID;
    std::lock_guard<std::recursive_mutex> ptx_lock__(ctx->mu);
    db->trio_built = false;
    db->cov_done = false;
    db->U = 0;
    return 0;
}

int pantax_hip_abundance_filter(uint32_t n_species, const uint64_t *hap_off, const pantax_hip_hap_metrics *met, const uint8_t *species_reported,
                                double single_cov_diff, int64_t min_cov, uint8_t *pass_out, double *sum_all_out, double *sum_pass_out,
                                double *sp_all_out, double *sp_pass_out) {
    if (!hap_off || !met || !pass_out) return PANTAX_HIP_E_INVALID;
    double all = 0.0, pass = 0.0;
    for (uint32_t s = 0; s < n_species; ++s) {
        const uint64_t h0 = hap_off[s], h1 = hap_off[s + 1];
        const bool reported = !species_reported || species_reported[s];
        const uint64_t group_size = h1 - h0;   // hap_id count per species (profile.rs:3219-3223)
        double s_all = 0.0, s_pass = 0.0;
        for (uint64_t h = h0; h < h1; ++h) {
            pass_out[h] = 0;
            if (!reported || !(met[h].has & PANTAX_HIP_HAS_SECOND)) continue;   // null predicted_coverage fails every comparison
            const double cov = met[h].second_sol;
            s_all += cov;                                                        // :3198 (sum skips nulls)
            const bool diff_ok = (met[h].has & PANTAX_HIP_HAS_TOTAL_DIFF) && met[h].total_cov_diff <= single_cov_diff;
            if ((group_size > 1 || diff_ok) && cov >= (double)min_cov && cov != 0.0) { pass_out[h] = 1; s_pass += cov; }   // :3232-3241
        }
        all += s_all; pass += s_pass;
        if (sp_all_out) sp_all_out[s] = s_all;
        if (sp_pass_out) sp_pass_out[s] = s_pass;
    }
    if (sum_all_out) *sum_all_out = all;
    if (sum_pass_out) *sum_pass_out = pass;
    return 0;
}

}  // extern "C"
