// api_strain.cpp -- a9..a14: first_filter_paths (profile.rs:1080-1227), the two LP solves,
// second_filter_paths (:1229-1285), abundace_constraint (:3028-3070), and the solver seam
// pantax_hip_pao_solve (X_opt signature, profile.rs:2690-2698).
//
// The whole strain step is enqueued on the ctx stream without waiting for the host: the two filter
// decisions are taken by small device kernels (stage_lad.hip), row / pattern counts stay on the
// device, and ONE download + synchronisation at the end brings back the raw per-haplotype and
// per-species results.  The host then only redoes the reporting arithmetic (rounded fractions,
// divergence, abundance constraint) on those values, using the decisions the device took.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include "lad.hpp"
#include "primitives.hpp"

using namespace ptx;

namespace {

inline double round2(double x) { return std::round(x * 100.0) / 100.0; }   // f64::round: half away from zero

// Every small per-species / per-haplotype result of the step lives in ONE device arena (db->d_arena):
// it is zeroed with one memset before the step and fetched with one copy after it.  The DevBufs that the
// kernels write through are non-owning views into it.
struct StrainRaw {   // host pointers into db->h_arena after fetch
    const uint32_t *nnz, *nvalid, *nzcnt, *sp_pat_off, *counts;
    const double *meanf, *amax, *nzsum, *x1, *x2, *obj1, *obj2;
    const int32_t *hap_bit, *sp_p, *st1, *st2, *it1, *it2;
    const uint8_t *fixed2, *need2;
    const unsigned long long *ratio;
};

struct ArenaLayout {
    size_t amax, nzsum, obj1, obj2, x1, x2, ratio, meanf, nvalid, nzcnt, sp_p, sp_pat_off, st1, st2, it1, it2, counts, nnz, hap_bit, fixed2, need2, total;
    ArenaLayout(uint32_t S, uint64_t H) {
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 15) & ~size_t(15); return o; };
        const size_t Hn = H ? H : 1;
        amax = take(8 * S); nzsum = take(8 * S); obj1 = take(8 * S); obj2 = take(8 * S);
        x1 = take(8 * Hn); x2 = take(8 * Hn); ratio = take(8 * Hn * 2);   // per column; column k of species s lives at hap_off[s] + k
        meanf = take(8 * Hn);
        nvalid = take(4 * S); nzcnt = take(4 * S); sp_p = take(4 * S); sp_pat_off = take(4 * ((size_t)S + 1));
        st1 = take(4 * S); st2 = take(4 * S); it1 = take(4 * S); it2 = take(4 * S); counts = take(16);
        nnz = take(4 * Hn); hap_bit = take(4 * Hn);
        fixed2 = take(Hn); need2 = take(S);
        total = off;
    }
};

int bind_arena(Ctx *ctx, Db *db, LadBatch &lb, const ArenaLayout &L) {
    const uint32_t S = db->S;
    const size_t Hn = db->H ? db->H : 1;
    PTX_HIP(ctx, db->d_arena.alloc(L.total));
    uint8_t *b = db->d_arena.p;
    if (!lb.prezeroed) PTX_HIP(ctx, hipMemsetAsync(b, 0, L.total, ctx->stream));   // unsolved species read back as x = 0, status 0, 0 pivots
    lb.d_amax.view(b + L.amax, S); lb.d_nzsum.view(b + L.nzsum, S); lb.d_obj.view(b + L.obj1, S); lb.d_obj2.view(b + L.obj2, S);
    lb.d_x.view(b + L.x1, Hn); lb.d_x2.view(b + L.x2, Hn);
    lb.d_ratio.view(b + L.ratio, Hn * 2);
    db->d_hap_mean.view(b + L.meanf, Hn);
    lb.d_nvalid.view(b + L.nvalid, S); lb.d_nzcnt.view(b + L.nzcnt, S); lb.d_p.view(b + L.sp_p, S); lb.d_sp_pat_off.view(b + L.sp_pat_off, (size_t)S + 1);
    lb.d_status.view(b + L.st1, S); lb.d_status2.view(b + L.st2, S); lb.d_iters.view(b + L.it1, S); lb.d_iters2.view(b + L.it2, S);
    lb.d_counts.view(b + L.counts, 4);
    db->d_hap_nnz.view(b + L.nnz, Hn); lb.d_hap_bit.view(b + L.hap_bit, Hn);
    lb.d_fixed2.view(b + L.fixed2, Hn); lb.d_need2.view(b + L.need2, S);
    return 0;
}

int fetch_arena_enqueue(Ctx *ctx, Db *db, const ArenaLayout &L, int slot) {
    for (int k = 0; k < 2; ++k) {   // both slots at once: the first step that runs AHEAD of another one must not pay for page-locking memory
        PTX_HIP(ctx, db->h_arena[k].reserve(L.total));
        if (!db->ev_step[k]) PTX_HIP(ctx, hipEventCreateWithFlags(&db->ev_step[k], hipEventDisableTiming));
    }
    PTX_HIP(ctx, hipMemcpyAsync(db->h_arena[slot].p, db->d_arena.p, L.total, hipMemcpyDeviceToHost, ctx->stream));   // the one host round trip of the step
    PTX_HIP(ctx, hipEventRecord(db->ev_step[slot], ctx->stream));   // a later step may already be enqueued behind it: the wait is for THIS step
    return 0;
}
int fetch_arena_wait(Ctx *ctx, Db *db, LadBatch &lb, const ArenaLayout &L, StrainRaw &r, int slot) {
    PTX_HIP(ctx, hipEventSynchronize(db->ev_step[slot]));
    const uint8_t *b = db->h_arena[slot].p;
    r.amax = (const double *)(b + L.amax); r.nzsum = (const double *)(b + L.nzsum); r.obj1 = (const double *)(b + L.obj1); r.obj2 = (const double *)(b + L.obj2);
    r.x1 = (const double *)(b + L.x1); r.x2 = (const double *)(b + L.x2); r.ratio = (const unsigned long long *)(b + L.ratio);
    r.meanf = (const double *)(b + L.meanf);
    r.nvalid = (const uint32_t *)(b + L.nvalid); r.nzcnt = (const uint32_t *)(b + L.nzcnt); r.sp_p = (const int32_t *)(b + L.sp_p);
    r.sp_pat_off = (const uint32_t *)(b + L.sp_pat_off);
    r.st1 = (const int32_t *)(b + L.st1); r.st2 = (const int32_t *)(b + L.st2); r.it1 = (const int32_t *)(b + L.it1); r.it2 = (const int32_t *)(b + L.it2);
    r.counts = (const uint32_t *)(b + L.counts);
    r.nnz = (const uint32_t *)(b + L.nnz); r.hap_bit = (const int32_t *)(b + L.hap_bit);
    r.fixed2 = b + L.fixed2; r.need2 = b + L.need2;
    if (r.counts[3]) return fail(ctx, PANTAX_HIP_E_STATE, "strain step: %u problems in the kernels that built the unique-trio index this step read (visit groups out of order, "
                                                           "or group offsets that are not this visit table's)", r.counts[3]);
    if (r.counts[2]) return fail(ctx, PANTAX_HIP_E_LIMIT, "strain step: more than %u membership patterns (internal: the pattern tables hold one entry per node)", lb.k_cap);
    return 0;
}

}  // namespace

namespace ptx {

// The two zero-fills of the strain step (result arena, membership masks) issued ahead of time: in the resident step the
// main stream idles before the coverage kernel while the trio index is built on the side stream, so they cost nothing
// there instead of ~10 us between coverage and the first strain kernel.  strain_enqueue consumes the flag.
int strain_prezero(Ctx *ctx, Db *db) {
    LadBatch &lb = db->lad;
    const ArenaLayout L(db->S, db->H);
    lb.prezeroed = false;
    PTX_TRY(bind_arena(ctx, db, lb, L));
    PTX_HIP(ctx, lb.d_mask.alloc(db->V));
    if (!use_node_haps(ctx, db)) PTX_TRY(zero_fill(ctx, lb.d_mask.p, db->V * sizeof(uint64_t)));   // (the by-node mask kernel writes every word)
    lb.prezeroed = true;
    return 0;
}

// Enqueues the whole strain step and the download of its result arena; nothing waits for the host.
// d_active: device [S] or null.
int strain_enqueue(Ctx *ctx, Db *db, const pantax_hip_strain_config *cfg, const uint8_t *d_active, int slot) {
    if (!db->cov_done) return fail(ctx, PANTAX_HIP_E_STATE, "strain_profile: call pantax_hip_node_coverage first");
    if (!db->trio_built) return fail(ctx, PANTAX_HIP_E_STATE, "strain_profile: call pantax_hip_trio_index first");
    if (cfg->sample_nodes < 0) return fail(ctx, PANTAX_HIP_E_INVALID, "strain_profile: sample_nodes %d", cfg->sample_nodes);
    if (cfg->solver_semantics != PANTAX_HIP_SEMANTICS_GUROBI && cfg->solver_semantics != PANTAX_HIP_SEMANTICS_HIGHS)
        return fail(ctx, PANTAX_HIP_E_INVALID, "strain_profile: solver_semantics %d", cfg->solver_semantics);
    const uint32_t S = db->S;
    LadBatch &lb = db->lad;
    const ArenaLayout L(S, db->H);
    const auto t_begin = std::chrono::steady_clock::now();
    double marks[10] = {0}; int n_marks = 0;
    auto mark = [&]() { if (n_marks < 10) marks[n_marks++] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    struct SlowReport { double *m; int *n; bool on; ~SlowReport() {
        if (*n && m[*n - 1] > 2.0 && on) {
            std::fprintf(stderr, "[strain_enqueue] slow call, ms at marks (bind, hap stats, node stats, sample, first filter, lad prepare, lad pair, fetch):");
            for (int i = 0; i < *n; ++i) std::fprintf(stderr, " %.2f", m[i]);
            std::fprintf(stderr, "\n");
        } } } slow_report{marks, &n_marks, ctx->cfg.trace};
    PTX_TRY(bind_arena(ctx, db, lb, L));
    // the error word of the index build this step reads (a REBUILD does not wait for the host: its kernels' complaints -- a visit group out of order,
    // group offsets that are not this table's -- come back with the step's results); taken here, before the next step's rebuild may clear it
    if (db->trio_scratch.d_tot.p)
        PTX_HIP(ctx, hipMemcpyAsync(lb.d_counts.p + 3, db->trio_scratch.d_tot.p + 2, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    mark();
    // the resident step (the coverage pass left its counts to node_stats_launch): this call's two statistics passes are the only readers of the coverage
    // arena, once -- they zero it for the next step's coverage pass.  A stage caller may read bases / trio_bases again, so nothing is cleaned for it.
    // cov_clean_async: the fill goes onto the side stream instead, in front of the LPs (below) -- not while every kernel is being clocked (a bracket on the side
    // stream measures the overlap, and the per-kernel table wants the fill's own time).  -1: from 1 GiB of arena on (cfg4: 4.8 GB, 17.73 -> 17.42 ms a step; the reference-DB shape 27.7 -> 27.3; at 0.4 GB the
    // two events cost more than the 0.07 ms fill)
    const bool clocked = ctx->timing && ctx->timing_filter.empty();     // (named launches only: the fill overlaps, unbracketed)
    const bool clean_async = db->cov_count_pending && !ctx->cfg.cov_self_clean && !clocked &&
                             (ctx->cfg.cov_clean_async > 0 || (ctx->cfg.cov_clean_async < 0 && db->cov_arena_total >= ((size_t)1 << 30)));
    // the species flags the coverage pass of THIS step masked its reads with (the resident step: cov_count_pending): the two statistics passes skip what it skipped
    const uint8_t *skip_absent = db->cov_count_pending ? d_active : nullptr;
    db->cov_self_clean = db->cov_count_pending && ctx->cfg.cov_self_clean && db->U != 0;
    PTX_TRY(hap_trio_stats_launch(ctx, db, db->d_hap_nnz, db->d_hap_mean, skip_absent));                 // a9 statistics
    mark();
    PTX_TRY(node_stats_launch(ctx, db, &lb, cfg->min_depth, skip_absent));                               // abundances + per-species stats
    if (db->cov_self_clean) { db->cov_arena_clean = true; db->cov_done = false; db->cov_self_clean = false; }   // (cov_done: the arena no longer holds a coverage result)
    mark();
    PTX_TRY(row_sample_apply(ctx, db, &lb, cfg->sample_nodes));
    mark();                             // a11 (no-op unless a species is larger than --sample)
    const FilterCfg fc{cfg->unique_trio_nodes_fraction, cfg->unique_trio_nodes_mean_count_f, cfg->single_cov_ratio, cfg->shift};
    PTX_TRY(first_filter_launch(ctx, db, &lb, d_active, fc));                               // a9 decision -> LP columns
    // nothing below reads the unique-trio tables: the NEXT step may rebuild them from here on, beside this step's row
    // sort and LPs (api_step.cpp)
    if (!db->ev_trio_free) PTX_HIP(ctx, hipEventCreateWithFlags(&db->ev_trio_free, hipEventDisableTiming));
    // ... by default from behind the ROW COMPACTION (lad_prepare records the event there): the compaction is a chained scan whose tiles
    // spin on their predecessors, and beside the rebuild's kernels its 3 ms stretched to 9-13 (round 4); PANTAX_TRIO_FREE=filter: from here
    const bool free_at_filter = ctx->cfg.trio_free_at_filter;
    db->trio_free_pending = !free_at_filter;
    if (free_at_filter) { PTX_HIP(ctx, hipEventRecord(db->ev_trio_free, ctx->stream)); db->trio_free_valid = true; }
    mark();
    int pmax_bound = 1;                                                                     // mask bits of a row key: min(#haps, 64) (wide species: a 64-bit hash)
    for (uint32_t s = 0; s < S; ++s) pmax_bound = std::max<int>(pmax_bound, (int)std::min<uint64_t>(db->h_hap_off[s + 1] - db->h_hap_off[s], LAD_MAXP));
    {
        const int rc_prep = lad_prepare(ctx, db, &lb, true, pmax_bound);                   // a10 + row grouping
        if (rc_prep != 0) {   // failed before it recorded the event: no stale event may order the next step's rebuild (it falls back to ev_seq)
            db->trio_free_pending = false; db->trio_free_valid = false;
            return rc_prep;
        }
    }
    if (db->trio_free_pending) { PTX_HIP(ctx, hipEventRecord(db->ev_trio_free, ctx->stream)); db->trio_free_valid = true; db->trio_free_pending = false; }
    mark();
    if (clean_async) PTX_TRY(coverage_arena_clean_async(ctx, db));   // beside the LPs: one workgroup per species, most of the memory system idle
    PTX_TRY(lad_pair_launch(ctx, db, &lb, pmax_bound, fc));                                 // LP 1 -> a13 decision -> LP 2, objectives
    mark();
    PTX_TRY(fetch_arena_enqueue(ctx, db, L, slot));
    mark();
    lb.prezeroed = false;
    return 0;
}

// Waits for the step (the one host round trip) and does the reporting arithmetic.  species_active /
// species_coverage: host [S] (may be null), read only after the wait -- a resident step fills them from the
// device's own species decisions through `after_wait`.
int strain_finish(Ctx *ctx, Db *db, const pantax_hip_strain_config *cfg, const uint8_t *species_active, const double *species_coverage,
                  pantax_hip_hap_metrics *met, pantax_hip_solve_info *info_out, void (*after_wait)(void *), void *after_wait_arg, int slot) {
    const uint32_t S = db->S;
    const uint64_t H = db->H;
    std::memset(met, 0, sizeof(pantax_hip_hap_metrics) * H);
    std::vector<pantax_hip_solve_info> info(S);
    std::memset(info.data(), 0, sizeof(pantax_hip_solve_info) * S);
    for (auto &i : info) i.obj1 = i.obj2 = NAN;
    LadBatch &lb = db->lad;
    const ArenaLayout L(S, H);
    StrainRaw r;
    PTX_TRY(fetch_arena_wait(ctx, db, lb, L, r, slot));                                         // the one host round trip
    if (after_wait) after_wait(after_wait_arg);
    lb.n_rows = r.counts[0]; lb.K = r.counts[1];
    lb.h_sp_pat_off.assign(r.sp_pat_off, r.sp_pat_off + S + 1);

    // ---- reporting (host): metrics of every haplotype from the raw values and the device's decisions
    for (uint32_t s = 0; s < S; ++s) {
        if (species_active && !species_active[s]) continue;
        const uint64_t h0 = db->h_hap_off[s], h1 = db->h_hap_off[s + 1];
        const uint32_t Hs = (uint32_t)(h1 - h0);
        if (Hs == 0) continue;
        const uint64_t Us = db->h_hap_trio_off[h1] - db->h_hap_trio_off[h0];
        const bool trio_mode = Hs != 1 && Us != 0;                                          // profile.rs:1098
        const bool single_mode = !trio_mode && (Hs == 1 || db->h_all_same[s]);              // :1191-1205, :1211-1224
        const int p = r.sp_p[s];
        info[s].n_candidates = p < 0 ? -p : p;
        bool failed = false; int fail_code = 0;
        if (p < 0) { failed = true; fail_code = PANTAX_HIP_E_LIMIT; }                        // (not produced any more: the first filter keeps any number of columns)
        std::vector<uint64_t> cand(p > 0 ? p : 0);                                          // column k -> global hap
        for (uint64_t h = h0; h < h1; ++h) if (r.hap_bit[h] >= 0 && r.hap_bit[h] < p) cand[r.hap_bit[h]] = h;
        // first-filter metrics (set for every haplotype that was looked at, candidate or not)
        if (trio_mode) {
            for (uint64_t h = h0; h < h1; ++h) {
                const uint64_t nt = db->h_hap_trio_off[h + 1] - db->h_hap_trio_off[h];
                if (nt == 0) continue;                                                      // :1119
                met[h].unique_trio_nodes_fraction = round2((double)r.nnz[h] / (double)nt); met[h].has |= PANTAX_HIP_HAS_FRACTION;   // :1136-1138
                if (r.hap_bit[h] >= 0) { met[h].frequencies_mean = r.meanf[h]; met[h].has |= PANTAX_HIP_HAS_FREQ_MEAN; }            // :1165 / :1180
            }
        } else if (single_mode) {
            const double fm = r.nzcnt[s] ? r.nzsum[s] / (double)r.nzcnt[s] : 0.0;
            met[h0].frequencies_mean = round2(fm); met[h0].has |= PANTAX_HIP_HAS_FREQ_MEAN;  // :1203-1204, :1222-1223
        }
        if (p > 0) {
            info[s].status1 = r.st1[s]; info[s].iters1 = r.it1[s]; info[s].obj1 = r.obj1[s];
            info[s].n_rows = r.nvalid[s];
            info[s].n_patterns = r.sp_pat_off[s + 1] - r.sp_pat_off[s];
            if (r.st1[s] != 0) { failed = true; fail_code = PANTAX_HIP_E_SOLVER; }           // profile.rs:2999-3003
        }
        if (p > 0 && !failed) {
            for (int k = 0; k < p; ++k) {
                pantax_hip_hap_metrics &m = met[cand[k]];
                // f32 ratio of exact integer sums (profile.rs:1344-1361 accumulates in f32; identical while sums < 2^24)
                const float cov = (float)r.ratio[(h0 + k) * 2], len = (float)r.ratio[(h0 + k) * 2 + 1];
                m.path_cov_ratio = (double)(cov / len); m.has |= PANTAX_HIP_HAS_RATIO;
                m.first_sol = r.x1[h0 + k]; m.has |= PANTAX_HIP_HAS_FIRST;
            }
            if (trio_mode) {                                                                // second_filter_paths, :1234-1268
                const bool rs = r.need2[s] != 0;                                            // LP 2 actually differed from LP 1
                // highs_opt keeps the first K columns of the second solution, K = number of survivors, and zips them with the candidates
                // (profile.rs:2865-2879): a survivor at position >= K gets no second_sol.  Gurobi & co: every survivor its own x (:1500-1508)
                int n_keep = 0;
                for (int k = 0; k < p; ++k) n_keep += r.fixed2[h0 + k] ? 0 : 1;
                const int k_lim = cfg->solver_semantics == PANTAX_HIP_SEMANTICS_HIGHS ? n_keep : p;
                info[s].status2 = rs ? r.st2[s] : r.st1[s]; info[s].iters2 = rs ? r.it2[s] : 0; info[s].obj2 = rs ? r.obj2[s] : r.obj1[s];
                if (info[s].status2 != 0) { failed = true; fail_code = PANTAX_HIP_E_SOLVER; }
                for (int k = 0; k < p && !failed; ++k) {
                    pantax_hip_hap_metrics &m = met[cand[k]];
                    const double fm = (m.has & PANTAX_HIP_HAS_FREQ_MEAN) ? m.frequencies_mean : 0.0;
                    const bool keep = !r.fixed2[h0 + k];
                    if (fm != 0.0) {                                                        // :1238
                        const double sol = m.first_sol;
                        const double f = round2(std::fabs(sol - fm) / (sol + fm));
                        m.divergence = f; m.has |= PANTAX_HIP_HAS_DIVERGENCE;
                        if (keep && f > cfg->unique_trio_nodes_mean_count_f) { m.is_rescue = 1; m.has |= PANTAX_HIP_HAS_RESCUE; }   // :1251-1259
                    }
                    if (keep && k < k_lim) { m.second_sol = rs ? r.x2[h0 + k] : r.x1[h0 + k]; m.has |= PANTAX_HIP_HAS_SECOND; }   // :1500-1508 / :2871-2879
                }
            } else if (single_mode) {                                                       // :1269-1278
                pantax_hip_hap_metrics &m = met[h0];
                const double fm = m.frequencies_mean;
                if (fm > 0.0) {
                    const double sol = m.first_sol;
                    m.divergence = round2(std::fabs(sol - fm) / (sol + fm)); m.has |= PANTAX_HIP_HAS_DIVERGENCE;
                    m.second_sol = sol; m.has |= PANTAX_HIP_HAS_SECOND;
                }
            } else {                                                                        // :1279-1283
                for (int k = 0; k < p; ++k) { pantax_hip_hap_metrics &m = met[cand[k]]; m.second_sol = m.first_sol; m.has |= PANTAX_HIP_HAS_SECOND; }
            }
        }
        // ---- failed species are dropped whole (reference returns None); abundace_constraint for the rest
        if (failed) {
            for (uint64_t h = h0; h < h1; ++h) std::memset(&met[h], 0, sizeof(met[h]));
            info[s].status1 = info[s].status1 ? info[s].status1 : fail_code;
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

}  // namespace ptx

extern "C" {

int pantax_hip_strain_profile(pantax_hip_ctx *ctx, pantax_hip_db *db, const pantax_hip_strain_config *cfg, const uint8_t *species_active,
                              const double *species_coverage, pantax_hip_hap_metrics *met, pantax_hip_solve_info *info_out) {
    if (!ctx || !db || !cfg || !met) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    const uint8_t *d_active = nullptr;
    if (species_active) { PTX_TRY(upload_small(ctx, db->d_active, species_active, db->S)); d_active = db->d_active.p; }
    if (db->step_inflight) return fail(ctx, PANTAX_HIP_E_STATE, "strain_profile: %d enqueued step(s) of this db have not been collected", db->step_inflight);
    PTX_TRY(strain_enqueue(ctx, db, cfg, d_active, 0));
    return strain_finish(ctx, db, cfg, species_active, species_coverage, met, info_out, nullptr, nullptr, 0);
}

int pantax_hip_pao_solve_batch(pantax_hip_ctx *ctx, const pantax_hip_species_batch *in, const pantax_hip_solution_batch *out) {
    if (!ctx || !in || !out || !in->node_off || !in->node_len || !in->node_abundance || !in->hap_off || !in->path_off || !in->path_nodes ||
        !in->cand_off || !in->cand_path_idx || !out->x || !out->status)
        return PANTAX_HIP_E_INVALID;
    const uint32_t S = in->n_species;
    if (S == 0) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve_batch: n_species == 0");
    PTX_ENTER(ctx);
    const uint64_t V = in->node_off[S], H = in->hap_off[S];
    // a resident DB around the caller's graphs: species s takes the node ids node_off[s]+1 .. node_off[s+1]
    std::vector<int64_t> rs(S), re(S);
    for (uint32_t s = 0; s < S; ++s) {
        if (in->node_off[s + 1] <= in->node_off[s]) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve_batch: species %u has no nodes", s);
        rs[s] = (int64_t)in->node_off[s] + 1; re[s] = (int64_t)in->node_off[s + 1];
    }
    pantax_hip_graphs g{S, rs.data(), re.data(), in->node_off, in->node_len, in->hap_off, in->path_off, in->path_nodes};
    pantax_hip_db *db = nullptr;
    PTX_TRY(pantax_hip_db_upload(ctx, &g, &db));
    std::unique_ptr<pantax_hip_db, void (*)(pantax_hip_db *)> guard(db, [](pantax_hip_db *d) { delete d; });
    LadBatch &lb = db->lad;
    lb.S = S;
    const ArenaLayout L(S, H);
    PTX_TRY(bind_arena(ctx, db, lb, L));
    std::vector<uint32_t> cov32(V ? V : 1, 0);
    if (in->node_base_cov) for (uint64_t v = 0; v < V; ++v) cov32[v] = (uint32_t)in->node_base_cov[v];
    PTX_TRY(upload(ctx, db->d_cov, cov32.data(), V));
    db->cov_count_pending = false;
    PTX_TRY(upload(ctx, lb.d_ab, in->node_abundance, V));
    std::vector<double> amax(S);
    std::vector<uint32_t> nvalid(S);
    lb.h_p.assign(S, 0);
    lb.h_cand.assign(H ? H : 1, 0);
    std::vector<uint8_t> fixed(H ? H : 1, 0);
    int pmax = 1;
    for (uint32_t s = 0; s < S; ++s) {
        double mx = -INFINITY; uint32_t nv = 0;
        for (uint64_t v = in->node_off[s]; v < in->node_off[s + 1]; ++v) { mx = std::max(mx, in->node_abundance[v]); if (in->node_abundance[v] > 0.0) ++nv; }
        amax[s] = mx; nvalid[s] = nv;
        const uint64_t c0 = in->cand_off[s], c1 = in->cand_off[s + 1], nh = in->hap_off[s + 1] - in->hap_off[s];
        out->status[s] = 0;
        if (c1 - c0 > nh) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve_batch: species %u has %llu candidates for %llu paths", s, (unsigned long long)(c1 - c0), (unsigned long long)nh);
        for (uint64_t k = c0; k < c1; ++k) {
            if (in->cand_path_idx[k] >= nh) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve_batch: species %u candidate %llu names path %u of %llu", s, (unsigned long long)(k - c0), in->cand_path_idx[k], (unsigned long long)nh);
            lb.h_cand[in->hap_off[s] + (k - c0)] = in->cand_path_idx[k];
            fixed[in->hap_off[s] + (k - c0)] = (in->fixed_zero && in->fixed_zero[k]) ? 1 : 0;
        }
        lb.h_p[s] = (int32_t)(c1 - c0);
        pmax = std::max(pmax, (int)std::min<uint64_t>(c1 - c0, LAD_MAXP));
    }
    PTX_TRY(upload(ctx, lb.d_amax, amax.data(), S));
    PTX_TRY(upload(ctx, lb.d_nvalid, nvalid.data(), S));
    PTX_TRY(lad_prepare(ctx, db, &lb, false, pmax));
    PTX_TRY(upload(ctx, lb.d_fixed2, fixed.data(), fixed.size()));
    PTX_TRY(lad_solve_launch(ctx, db, &lb, pmax, nullptr, lb.d_fixed2.p, lb.d_x.p, lb.d_obj.p, lb.d_status.p, lb.d_iters.p));
    std::vector<double> x(H ? H : 1), obj(S);
    std::vector<unsigned long long> ratio((H ? H : 1) * 2);
    std::vector<int32_t> st(S), it(S);
    std::vector<uint32_t> counts(4);
    PTX_TRY(download(ctx, x.data(), lb.d_x.p, x.size()));
    PTX_TRY(download(ctx, obj.data(), lb.d_obj.p, S));
    PTX_TRY(download(ctx, st.data(), lb.d_status.p, S));
    PTX_TRY(download(ctx, it.data(), lb.d_iters.p, S));
    PTX_TRY(download(ctx, ratio.data(), lb.d_ratio.p, ratio.size()));
    PTX_TRY(download(ctx, counts.data(), lb.d_counts.p, 4));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (counts[2]) return fail(ctx, PANTAX_HIP_E_LIMIT, "pao_solve_batch: more membership patterns than this build sizes for");
    for (uint32_t s = 0; s < S; ++s) {
        const uint64_t c0 = in->cand_off[s];
        const int p = lb.h_p[s];
        for (int k = 0; k < p; ++k) {
            out->x[c0 + k] = x[in->hap_off[s] + k];
            if (out->path_cov_ratio) out->path_cov_ratio[c0 + k] = (float)ratio[(in->hap_off[s] + k) * 2] / (float)ratio[(in->hap_off[s] + k) * 2 + 1];
        }
        if (out->obj) out->obj[s] = (p > 0 && nvalid[s]) ? obj[s] : 0.0;
        if (out->iters) out->iters[s] = p > 0 ? it[s] : 0;
        if (p > 0 && st[s] != 0) out->status[s] = PANTAX_HIP_E_SOLVER;
    }
    return 0;
}

// one species = a batch of one (same kernels, same answers)
int pantax_hip_pao_solve(pantax_hip_ctx *ctx, uint32_t n_nodes, const int64_t *node_len, const double *node_abundance,
                         const uint64_t *node_base_cov, uint32_t n_paths, const uint64_t *path_off, const uint32_t *path_nodes,
                         uint32_t n_cand, const uint32_t *cand_path_idx, const uint8_t *fixed_zero, double *x_out,
                         float *path_cov_ratio_out, double *obj_out, int32_t *status_out) {
    if (!ctx || !node_len || !node_abundance || !path_off || !path_nodes || !cand_path_idx || !x_out) return PANTAX_HIP_E_INVALID;
    if (n_cand == 0) return fail(ctx, PANTAX_HIP_E_INVALID, "pao_solve: no candidate paths (the reference skips the solver, profile.rs:2968)");
    PTX_ENTER(ctx);
    const uint64_t node_off[2] = {0, n_nodes}, hap_off[2] = {0, n_paths}, cand_off[2] = {0, n_cand};
    const pantax_hip_species_batch in{1, node_off, node_len, node_abundance, node_base_cov, hap_off, path_off, path_nodes, cand_off, cand_path_idx, fixed_zero};
    int32_t st = 0, it = 0;
    const pantax_hip_solution_batch out{x_out, path_cov_ratio_out, obj_out, &st, &it};
    PTX_TRY(pantax_hip_pao_solve_batch(ctx, &in, &out));
    if (status_out) *status_out = st == PANTAX_HIP_E_SOLVER ? 1 : st;
    if (st != 0) return fail(ctx, PANTAX_HIP_E_SOLVER, "pao_solve: LAD solver stopped after %d pivots", it);
    return 0;
}

}  // extern "C"
