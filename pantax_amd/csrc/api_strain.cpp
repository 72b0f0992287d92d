// api_strain.cpp -- a9..a14 host orchestration: first_filter_paths (profile.rs:1080-1227), the two
// LP solves on device, second_filter_paths (:1229-1285), abundace_constraint (:3028-3070), and the
// solver seam pantax_hip_pao_solve (X_opt signature, profile.rs:2690-2698).
// The scalar filter logic is host C++ (a handful of flops per haplotype); everything that touches
// per-node or per-trio data runs in the kernels of stage_lad.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include "lad.hpp"

using namespace ptx;

namespace {

inline double round2(double x) { return std::round(x * 100.0) / 100.0; }   // f64::round: half away from zero

struct SpeciesState {
    std::vector<uint32_t> cand;   // possible_paths_idx (hap index within species)
    std::vector<uint8_t> keep;    // second_possible_paths_idx membership per candidate
    bool same_path = false, second_opt = false, failed = false;
    int fail_code = 0;
};

// shared by strain_profile and pao_solve: runs prepare + solve #1 (+ optional second solve driven
// by `second` callback) on an already filled LadBatch/candidate list
int run_solve(Ctx *ctx, const Db *db, LadBatch *lb, const std::vector<int32_t> &list, std::vector<double> &x, std::vector<double> &obj,
              std::vector<int32_t> &status, std::vector<int32_t> &iters) {
    PTX_TRY(lad_solve_launch(ctx, db, lb, list));
    uint32_t S = db->S;
    x.resize((size_t)S * LAD_MAXP); obj.resize(S); status.resize(S); iters.resize(S);
    if (list.empty()) return 0;
    PTX_TRY(download(ctx, x.data(), lb->d_x.p, (size_t)S * LAD_MAXP));
    PTX_TRY(download(ctx, obj.data(), lb->d_obj.p, S));
    PTX_TRY(download(ctx, status.data(), lb->d_status.p, S));
    PTX_TRY(download(ctx, iters.data(), lb->d_iters.p, S));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // namespace

extern "C" {

int pantax_hip_strain_profile(pantax_hip_ctx *ctx, pantax_hip_db *db, const pantax_hip_strain_config *cfg, const uint8_t *species_active,
                              const double *species_coverage, pantax_hip_hap_metrics *met, pantax_hip_solve_info *info_out) {
    if (!ctx || !db || !cfg || !met) return PANTAX_HIP_E_INVALID;
    PTX_HIP(ctx, hipSetDevice(ctx->device));
    if (!db->cov_done) return fail(ctx, PANTAX_HIP_E_STATE, "strain_profile: call pantax_hip_node_coverage first");
    if (!db->trio_built) return fail(ctx, PANTAX_HIP_E_STATE, "strain_profile: call pantax_hip_trio_index first");
    if (cfg->sample_nodes != 0)
        return fail(ctx, PANTAX_HIP_E_LIMIT, "strain_profile: --sample %d requested; row sub-sampling (profile.rs:1394-1400, rand 0.9.2 ChaCha12) is not implemented, run with sample 0", cfg->sample_nodes);
    const uint32_t S = db->S;
    const uint64_t H = db->H;
    std::memset(met, 0, sizeof(pantax_hip_hap_metrics) * H);
    std::vector<pantax_hip_solve_info> info(S);
    std::memset(info.data(), 0, sizeof(pantax_hip_solve_info) * S);
    for (auto &i : info) i.obj1 = i.obj2 = NAN;

    // ---- device reductions: per-hap trio stats, node abundance + per-species stats
    DevBuf<uint32_t> &d_nnz = db->d_hap_nnz;
    DevBuf<double> &d_mean = db->d_hap_mean;
    LadBatch &lb = db->lad;
    PTX_TRY(hap_trio_stats_launch(ctx, db, d_nnz, d_mean));
    PTX_TRY(node_stats_launch(ctx, db, &lb, cfg->min_depth));
    std::vector<uint32_t> nnz(H ? H : 1), nvalid(S), nzcnt(S);
    std::vector<double> meanf(H ? H : 1), amax(S), nzsum(S);
    PTX_TRY(download(ctx, nnz.data(), d_nnz.p, H));
    PTX_TRY(download(ctx, meanf.data(), d_mean.p, H));
    PTX_TRY(download(ctx, amax.data(), lb.d_amax.p, S));
    PTX_TRY(download(ctx, nvalid.data(), lb.d_nvalid.p, S));
    PTX_TRY(download(ctx, nzsum.data(), lb.d_nzsum.p, S));
    PTX_TRY(download(ctx, nzcnt.data(), lb.d_nzcnt.p, S));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));

    // ---- first_filter_paths (profile.rs:1080-1227)
    std::vector<SpeciesState> st(S);
    lb.h_p.assign(S, 0);
    lb.h_cand.assign((size_t)S * LAD_MAXP, 0);
    for (uint32_t s = 0; s < S; ++s) {
        if (species_active && !species_active[s]) continue;
        const uint64_t h0 = db->h_hap_off[s], h1 = db->h_hap_off[s + 1];
        const uint32_t Hs = (uint32_t)(h1 - h0);
        if (Hs == 0) continue;
        const uint64_t Us = db->h_hap_trio_off[h1] - db->h_hap_trio_off[h0];
        SpeciesState &ss = st[s];
        if (Hs != 1 && Us != 0) {                                  // :1098
            for (uint32_t h = 0; h < Hs; ++h) {
                uint64_t nt = db->h_hap_trio_off[h0 + h + 1] - db->h_hap_trio_off[h0 + h];
                if (nt == 0) continue;                             // :1119
                double frac = (double)nnz[h0 + h] / (double)nt;    // :1135
                pantax_hip_hap_metrics &m = met[h0 + h];
                m.unique_trio_nodes_fraction = round2(frac); m.has |= PANTAX_HIP_HAS_FRACTION;   // :1136-1138
                double fm = meanf[h0 + h];
                if (cfg->shift) {                                  // :1140-1165
                    double sh;
                    if (fm >= 1.0) { sh = cfg->unique_trio_nodes_fraction + (0.8 - cfg->unique_trio_nodes_fraction) * fm / 100.0; if (sh > 0.8) sh = 0.8; }
                    else sh = cfg->unique_trio_nodes_fraction * fm;
                    if (frac < sh) continue;
                } else if (frac < cfg->unique_trio_nodes_fraction) continue;   // :1168
                m.frequencies_mean = fm; m.has |= PANTAX_HIP_HAS_FREQ_MEAN;
                ss.cand.push_back(h);
            }
        } else {
            bool all_same = true;
            if (Hs != 1) {                                         // :1187-1190
                const uint64_t q0 = db->h_path_off[h0], l0 = db->h_path_off[h0 + 1] - q0;
                for (uint32_t h = 1; h < Hs && all_same; ++h) {
                    const uint64_t q = db->h_path_off[h0 + h], l = db->h_path_off[h0 + h + 1] - q;
                    if (l != l0 || std::memcmp(&db->h_path_nodes[q], &db->h_path_nodes[q0], l0 * sizeof(uint32_t)) != 0) all_same = false;
                }
            }
            if (Hs == 1 || all_same) {                             // :1191-1205, :1211-1224
                ss.same_path = Hs != 1;
                double fm = nzcnt[s] ? nzsum[s] / (double)nzcnt[s] : 0.0;
                met[h0].frequencies_mean = round2(fm); met[h0].has |= PANTAX_HIP_HAS_FREQ_MEAN;
                ss.cand.push_back(0);
            } else {
                for (uint32_t h = 0; h < Hs; ++h) ss.cand.push_back(h);   // :1208
            }
        }
        info[s].n_candidates = (int32_t)ss.cand.size();
        if (ss.cand.size() > (size_t)LAD_MAXP) {
            ss.failed = true; ss.fail_code = PANTAX_HIP_E_LIMIT;   // this build: <= 64 candidate paths per species
            continue;
        }
        lb.h_p[s] = (int32_t)ss.cand.size();
        for (size_t k = 0; k < ss.cand.size(); ++k) lb.h_cand[(size_t)s * LAD_MAXP + k] = ss.cand[k];
    }

    // ---- a10 + row grouping on device, then solve #1
    PTX_TRY(lad_prepare(ctx, db, &lb));
    std::vector<unsigned long long> ratio((size_t)S * LAD_MAXP * 2);
    PTX_TRY(download(ctx, ratio.data(), lb.d_ratio.p, ratio.size()));
    std::vector<double> ub((size_t)S * LAD_MAXP, 0.0);
    std::vector<int32_t> list1;
    for (uint32_t s = 0; s < S; ++s) {
        if (lb.h_p[s] <= 0) continue;
        list1.push_back((int32_t)s);
        for (int k = 0; k < lb.h_p[s]; ++k) ub[(size_t)s * LAD_MAXP + k] = 1.05 * amax[s];   // profile.rs:1327
    }
    PTX_TRY(upload(ctx, lb.d_ub, ub.data(), ub.size()));
    std::vector<double> x1, obj1, x2, obj2;
    std::vector<int32_t> st1, it1, st2, it2;
    PTX_TRY(run_solve(ctx, db, &lb, list1, x1, obj1, st1, it1));

    // ---- path_cov_ratio, first_sol, second_filter_paths (profile.rs:1229-1285)
    std::vector<int32_t> list2;
    for (int32_t s : list1) {
        SpeciesState &ss = st[s];
        const uint64_t h0 = db->h_hap_off[s];
        const uint32_t Hs = (uint32_t)(db->h_hap_off[s + 1] - h0);
        const uint64_t Us = db->h_hap_trio_off[db->h_hap_off[s + 1]] - db->h_hap_trio_off[h0];
        info[s].status1 = st1[s]; info[s].iters1 = it1[s]; info[s].obj1 = obj1[s];
        info[s].n_rows = nvalid[s];
        info[s].n_patterns = lb.h_sp_pat_off[s + 1] - lb.h_sp_pat_off[s];
        if (st1[s] != 0) { ss.failed = true; ss.fail_code = PANTAX_HIP_E_SOLVER; continue; }   // profile.rs:2999-3003
        const int p = lb.h_p[s];
        ss.keep.assign(p, 0);
        for (int k = 0; k < p; ++k) {
            pantax_hip_hap_metrics &m = met[h0 + ss.cand[k]];
            // f32 ratio of exact integer sums (profile.rs:1344-1361 accumulates in f32; identical while sums < 2^24)
            float cov = (float)ratio[((size_t)s * LAD_MAXP + k) * 2], len = (float)ratio[((size_t)s * LAD_MAXP + k) * 2 + 1];
            m.path_cov_ratio = (double)(cov / len); m.has |= PANTAX_HIP_HAS_RATIO;
            m.first_sol = x1[(size_t)s * LAD_MAXP + k]; m.has |= PANTAX_HIP_HAS_FIRST;
        }
        if (Hs != 1 && Us > 0) {
            ss.second_opt = true;
            for (int k = 0; k < p; ++k) {
                pantax_hip_hap_metrics &m = met[h0 + ss.cand[k]];
                double fm = (m.has & PANTAX_HIP_HAS_FREQ_MEAN) ? m.frequencies_mean : 0.0;
                if (fm == 0.0) continue;                            // :1238
                double sol = m.first_sol;
                double fr = round2(std::fabs(sol - fm) / (sol + fm));
                m.divergence = fr; m.has |= PANTAX_HIP_HAS_DIVERGENCE;
                if (fr > cfg->unique_trio_nodes_mean_count_f) {
                    if (fr <= 0.6) {
                        double sc = m.unique_trio_nodes_fraction * m.path_cov_ratio;
                        if (sc < cfg->single_cov_ratio || sol == 0.0) continue;
                        m.is_rescue = 1; m.has |= PANTAX_HIP_HAS_RESCUE; ss.keep[k] = 1;
                    }
                } else if (sol != 0.0) ss.keep[k] = 1;
            }
            list2.push_back(s);
        } else if ((Hs != 1 && Us == 0 && ss.same_path) || Hs == 1) {
            pantax_hip_hap_metrics &m = met[h0];
            double fm = m.frequencies_mean;
            if (fm > 0.0) {
                double sol = m.first_sol;
                m.divergence = round2(std::fabs(sol - fm) / (sol + fm)); m.has |= PANTAX_HIP_HAS_DIVERGENCE;
                m.second_sol = sol; m.has |= PANTAX_HIP_HAS_SECOND;
            }
        } else {
            for (int k = 0; k < p; ++k) { pantax_hip_hap_metrics &m = met[h0 + ss.cand[k]]; m.second_sol = m.first_sol; m.has |= PANTAX_HIP_HAS_SECOND; }
        }
    }
    // ---- solve #2 with dropped candidates pinned to 0 (profile.rs:1484-1508, Gurobi semantics).
    // When the second filter drops nothing the second LP is the first LP again (m.reset() + no new
    // constraint), so its optimum is the one already computed: reuse it instead of re-solving.
    if (!list2.empty()) {
        std::vector<int32_t> resolve;
        for (int32_t s : list2) {
            bool any_dropped = false;
            for (int k = 0; k < lb.h_p[s]; ++k) if (!st[s].keep[k]) { ub[(size_t)s * LAD_MAXP + k] = 0.0; any_dropped = true; }
            if (any_dropped) resolve.push_back(s);
        }
        if (!resolve.empty()) {
            PTX_TRY(upload(ctx, lb.d_ub, ub.data(), ub.size()));
            PTX_TRY(run_solve(ctx, db, &lb, resolve, x2, obj2, st2, it2));
        }
        std::vector<uint8_t> resolved(S, 0);
        for (int32_t s : resolve) resolved[s] = 1;
        for (int32_t s : list2) {
            SpeciesState &ss = st[s];
            const bool rs = resolved[s] != 0;
            info[s].status2 = rs ? st2[s] : st1[s]; info[s].iters2 = rs ? it2[s] : 0; info[s].obj2 = rs ? obj2[s] : obj1[s];
            if (info[s].status2 != 0) { ss.failed = true; ss.fail_code = PANTAX_HIP_E_SOLVER; continue; }
            const uint64_t h0 = db->h_hap_off[s];
            for (int k = 0; k < lb.h_p[s]; ++k)
                if (ss.keep[k]) {
                    pantax_hip_hap_metrics &m = met[h0 + ss.cand[k]];
                    m.second_sol = rs ? x2[(size_t)s * LAD_MAXP + k] : x1[(size_t)s * LAD_MAXP + k];
                    m.has |= PANTAX_HIP_HAS_SECOND;
                }
        }
    }
    // ---- failed species are dropped whole (reference returns None); abundace_constraint for the rest
    for (uint32_t s = 0; s < S; ++s) {
        if (species_active && !species_active[s]) continue;
        const uint64_t h0 = db->h_hap_off[s], h1 = db->h_hap_off[s + 1];
        if (h1 == h0) continue;
        if (st[s].failed) {
            for (uint64_t h = h0; h < h1; ++h) std::memset(&met[h], 0, sizeof(met[h]));
            info[s].status1 = info[s].status1 ? info[s].status1 : st[s].fail_code;
            continue;
        }
        if (!species_coverage) continue;
        const double sc = species_coverage[s];                     // profile.rs:3044-3047
        double sum = 0.0, mx = -INFINITY;
        for (uint64_t h = h0; h < h1; ++h) {
            pantax_hip_hap_metrics &m = met[h];
            if ((m.has & PANTAX_HIP_HAS_RESCUE) && m.is_rescue && (m.has & PANTAX_HIP_HAS_FIRST) && (m.has & PANTAX_HIP_HAS_SECOND))
                m.second_sol = std::min(m.first_sol, m.second_sol);   // :3031-3035
            double v = (m.has & PANTAX_HIP_HAS_SECOND) ? m.second_sol : 0.0;
            sum += v; mx = std::max(mx, v);
        }
        double diff = std::fabs(sum - sc) / ((sum + sc) / 2.0);     // :3050
        for (uint64_t h = h0; h < h1; ++h) { met[h].total_cov_diff = diff; met[h].has |= PANTAX_HIP_HAS_TOTAL_DIFF; }
        if (mx > 1.05 * sc) {                                       // :3055-3066
            double f = sc / sum;
            for (uint64_t h = h0; h < h1; ++h) {
                pantax_hip_hap_metrics &m = met[h];
                if (!((m.has & PANTAX_HIP_HAS_RESCUE) && m.is_rescue) && (m.has & PANTAX_HIP_HAS_SECOND)) m.second_sol *= f;
            }
        }
    }
    if (info_out) std::memcpy(info_out, info.data(), sizeof(pantax_hip_solve_info) * S);
    return 0;
}

int pantax_hip_pao_solve(pantax_hip_ctx *ctx, uint32_t n_nodes, const int64_t *node_len, const double *node_abundance,
                         const uint64_t *node_base_cov, uint32_t n_paths, const uint64_t *path_off, const uint32_t *path_nodes,
                         uint32_t n_cand, const uint32_t *cand_path_idx, const uint8_t *fixed_zero, double *x_out,
                         float *path_cov_ratio_out, double *obj_out, int32_t *status_out) {
    if (!ctx || !node_len || !node_abundance || !path_off || !path_nodes || !cand_path_idx || !x_out) return PANTAX_HIP_E_INVALID;
    if (n_cand == 0) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve: no candidate paths (the reference skips the solver, profile.rs:2968)");
    if (n_cand > (uint32_t)LAD_MAXP) return fail(ctx, PANTAX_HIP_E_LIMIT, "pao_solve: %u candidate paths; this build handles <= %d", n_cand, LAD_MAXP);
    PTX_HIP(ctx, hipSetDevice(ctx->device));
    // a one-species resident DB around the caller's graph
    int64_t rs = 1, re = n_nodes;
    uint64_t node_off[2] = {0, n_nodes}, hap_off[2] = {0, n_paths};
    pantax_hip_graphs g{1, &rs, &re, node_off, node_len, hap_off, path_off, path_nodes};
    pantax_hip_db *db = nullptr;
    PTX_TRY(pantax_hip_db_upload(ctx, &g, &db));
    std::unique_ptr<pantax_hip_db, void (*)(pantax_hip_db *)> guard(db, [](pantax_hip_db *d) { delete d; });
    LadBatch lb;
    lb.S = 1;
    std::vector<uint32_t> cov32(n_nodes, 0);
    if (node_base_cov) for (uint32_t v = 0; v < n_nodes; ++v) cov32[v] = (uint32_t)node_base_cov[v];
    PTX_TRY(upload(ctx, db->d_cov, cov32.data(), n_nodes));
    PTX_TRY(upload(ctx, lb.d_ab, node_abundance, n_nodes));
    double amax = -INFINITY; uint32_t nvalid = 0;
    for (uint32_t v = 0; v < n_nodes; ++v) { amax = std::max(amax, node_abundance[v]); if (node_abundance[v] > 0.0) ++nvalid; }
    PTX_TRY(upload(ctx, lb.d_amax, &amax, 1));
    PTX_TRY(upload(ctx, lb.d_nvalid, &nvalid, 1));
    lb.h_p.assign(1, (int32_t)n_cand);
    lb.h_cand.assign(LAD_MAXP, 0);
    for (uint32_t k = 0; k < n_cand; ++k) {
        if (cand_path_idx[k] >= n_paths) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve: candidate %u names path %u of %u", k, cand_path_idx[k], n_paths);
        lb.h_cand[k] = cand_path_idx[k];
    }
    PTX_TRY(lad_prepare(ctx, db, &lb));
    std::vector<double> ub(LAD_MAXP, 0.0);
    for (uint32_t k = 0; k < n_cand; ++k) ub[k] = (fixed_zero && fixed_zero[k]) ? 0.0 : 1.05 * amax;
    PTX_TRY(upload(ctx, lb.d_ub, ub.data(), ub.size()));
    std::vector<double> x, obj;
    std::vector<int32_t> st, it;
    PTX_TRY(run_solve(ctx, db, &lb, std::vector<int32_t>{0}, x, obj, st, it));
    for (uint32_t k = 0; k < n_cand; ++k) x_out[k] = x[k];
    if (path_cov_ratio_out) {
        std::vector<unsigned long long> ratio(LAD_MAXP * 2);
        PTX_TRY(download(ctx, ratio.data(), lb.d_ratio.p, ratio.size()));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t k = 0; k < n_cand; ++k) path_cov_ratio_out[k] = (float)ratio[2 * k] / (float)ratio[2 * k + 1];
    }
    if (obj_out) *obj_out = obj[0];
    if (status_out) *status_out = st[0];
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st[0] != 0) return fail(ctx, PANTAX_HIP_E_SOLVER, "pao_solve: LAD solver stopped with status %d after %d pivots", st[0], it[0]);
    return 0;
}

}  // extern "C"
