// api_step.cpp -- pantax_hip_profile_step: one pass of the hot path over RESIDENT inputs, the in-memory
// core of profile::profile (profile.rs:3325-3364) between "GAF parsed" and "tables written":
// rcls_profile -> species_profiling -> (trio_nodes_info) -> get_node_abundances -> strain_profiling ->
// abundance_est filters.  Everything is enqueued on the ctx stream; the host waits ONCE, at the end.
// The species decision (a3) is taken by a device kernel, so binning, coverage and the LP solves follow each
// other without a round trip.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lad.hpp"

using namespace ptx;

namespace {
struct SpOut { Db *db; uint8_t *keep; double *absolute; int slot; };
void copy_species_out(void *arg) {   // runs right after the step's single wait
    SpOut *o = static_cast<SpOut *>(arg);
    const uint32_t S = o->db->S;
    std::memcpy(o->absolute, o->db->h_sp_out[o->slot].p, sizeof(double) * S);
    std::memcpy(o->keep, o->db->h_sp_out[o->slot].p + sizeof(double) * S, S);
}

// Everything of one step enqueued on the device, nothing waited for.  The device runs steps strictly one after the other:
// the side stream (unique-trio build) first waits for all the main-stream work enqueued so far, i.e. for the previous step.
int step_enqueue(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads, const double *avg_len, const pantax_hip_step_config *cfg) {
    if (db->d_node_rec.p == nullptr) return fail(ctx, PANTAX_HIP_E_STATE, "profile_step: the db was uploaded without graphs (ranges only)");
    if (db->step_inflight >= 2) return fail(ctx, PANTAX_HIP_E_STATE, "profile_step_enqueue: two steps of this db are already in flight; collect one first");
    const uint32_t S = db->S;
    const int slot = db->step_enq;
    // debug aid (PANTAX_HIP_TRACE): host time of the sections of an enqueue that took more than 2 ms
    const auto t_begin = std::chrono::steady_clock::now();
    double marks[8] = {0}; int n_marks = 0;
    auto mark = [&]() { if (n_marks < 8) marks[n_marks++] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    struct SlowReport { double *m; int *n; bool on; ~SlowReport() {
        if (*n && m[*n - 1] > 2.0 && on) {
            std::fprintf(stderr, "[step_enqueue] slow call, ms at marks (trio enqueued, bin+species, prezero+cov prepare, coverage, strain):");
            for (int i = 0; i < *n; ++i) std::fprintf(stderr, " %.2f", m[i]);
            std::fprintf(stderr, "\n");
        } } } slow_report{marks, &n_marks, ctx->cfg.trace};
    // a7 first, on the side stream: the unique-trio index depends on the graphs only (the reference rebuilds it every
    // run, profile.rs:2936), so it is built while the main stream bins the reads and takes the species decision
    bool forked = false;
    if (cfg->rebuild_trio && !(db->trio_prefetched && db->trio_built)) { db->trio_built = false; db->cov_done = false; db->U = 0; }
    db->trio_prefetched = false;   // (a prefetched index serves ONE step: the run it was started for)
    if (!db->trio_built) {
        // the previous step still reads the index this build replaces -- up to its first filter (strain_enqueue records the event
        // behind it); what follows there (masks, row sort, LPs, objective) runs beside the rebuild.  Without such an event
        // (stage calls in between) the side stream waits for everything enqueued so far.
        if (db->trio_free_valid && !ctx->cfg.trio_after_step) PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, db->ev_trio_free, 0));
        else {
            PTX_HIP(ctx, hipEventRecord(ctx->ev_seq, ctx->stream));
            PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_seq, 0));
        }
        hipStream_t main_stream = ctx->stream;
        ctx->stream_main = main_stream; ctx->stream = ctx->stream2;
        const int rc = trio_index_build(ctx, db, false);   // without the row-order export copies: no stage of the step reads them
        const hipError_t e = hipEventRecord(ctx->ev_fork, ctx->stream2);
        ctx->stream = main_stream; ctx->stream_main = nullptr;
        if (rc != 0) return rc;
        PTX_HIP(ctx, e);
        forked = true;
    }
    // a failure between the fork and the join must not leave the index half built behind a `trio_built` flag: the next
    // call would read it with no ordering against the side stream
    struct ForkGuard {
        Ctx *c; Db *d; bool armed;
        ~ForkGuard() { if (armed) { (void)hipStreamSynchronize(c->stream2); d->trio_built = false; d->cov_done = false; } }
    } fork_guard{ctx, db, forked};
    mark();
    // a2 + a3 counters
    PTX_HIP(ctx, db->d_counters.alloc(bin_counter_words(S)));
    PTX_TRY(bin_reads_launch(ctx, db, reads, db->d_counters.p));
    // a3 decision on the device: keep -> d_active, predicted_coverage -> d_sp_abs
    PTX_TRY(upload_small(ctx, db->d_avg_len, avg_len, S));
    PTX_HIP(ctx, db->d_sp_out.alloc(sizeof(double) * S + S));             // [S f64 predicted_coverage][S u8 keep]: one copy back
    db->d_sp_abs.view(db->d_sp_out.p, S);
    db->d_active.view(db->d_sp_out.p + sizeof(double) * S, S);
    PTX_TRY(species_profile_launch(ctx, db, reads, db->d_counters.p, db->d_avg_len.p, cfg->filtered, db->d_active.p, db->d_sp_abs.p));
    for (int k = 0; k < 2; ++k) PTX_HIP(ctx, db->h_sp_out[k].reserve(sizeof(double) * S + S));
    PTX_HIP(ctx, hipMemcpyAsync(db->h_sp_out[slot].p, db->d_sp_out.p, sizeof(double) * S + S, hipMemcpyDeviceToHost, ctx->stream));
    mark();
    PTX_TRY(strain_prezero(ctx, db));   // zero-fills of the strain step, while this stream would wait for the trio index anyway
    struct PreZeroGuard { LadBatch &lb; ~PreZeroGuard() { lb.prezeroed = false; } } prezero_guard{db->lad};   // never outlives this call
    PTX_TRY(coverage_prepare(ctx, db, reads, true));   // arena zero-fill + walk sums of long reads: need the binning, not the trio index
    struct CovPrepGuard { Db *d; ~CovPrepGuard() { d->cov_prepared = false; } } covprep_guard{db};
    mark();
    // a8 needs both
    if (forked) PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
    fork_guard.armed = false;   // joined: everything later on the main stream is ordered behind the index
    PTX_TRY(coverage_launch(ctx, db, reads, db->d_active.p, true, !ctx->cfg.cov_count));   // (PANTAX_COV_COUNT=1: popcount_kernel as in the stage call)
    mark();
    // a9 .. a14
    pantax_hip_strain_config sc{cfg->unique_trio_nodes_fraction, cfg->unique_trio_nodes_mean_count_f, cfg->single_cov_ratio, cfg->min_depth,
                                cfg->shift, cfg->sample_nodes, cfg->solver_semantics};
    PTX_TRY(strain_enqueue(ctx, db, &sc, db->d_active.p, slot));
    mark();
    db->step_cfg[slot] = *cfg;
    db->step_enq ^= 1;
    ++db->step_inflight;
    return 0;
}

// The oldest enqueued step: its one host wait, the reporting arithmetic and the a15 filters (host scalars).
int step_collect(pantax_hip_ctx *ctx, pantax_hip_db *db, uint8_t *keep_out, double *absolute_out, pantax_hip_hap_metrics *met,
                 pantax_hip_solve_info *info_out, uint8_t *pass_out, double *species_sum_all_out, double *species_sum_pass_out) {
    if (db->step_inflight <= 0) return fail(ctx, PANTAX_HIP_E_STATE, "profile_step_collect: no enqueued step of this db is waiting");
    const uint32_t S = db->S;
    const int slot = db->step_col;
    const pantax_hip_step_config cfg = db->step_cfg[slot];
    db->step_col ^= 1;
    --db->step_inflight;   // also when the step turns out to have failed: its slot is free again
    pantax_hip_strain_config sc{cfg.unique_trio_nodes_fraction, cfg.unique_trio_nodes_mean_count_f, cfg.single_cov_ratio, cfg.min_depth, cfg.shift,
                                cfg.sample_nodes, cfg.solver_semantics};
    SpOut so{db, keep_out, absolute_out, slot};
    std::vector<pantax_hip_solve_info> info(S);
    PTX_TRY(strain_finish(ctx, db, &sc, keep_out, absolute_out, met, info.data(), copy_species_out, &so, slot));
    if (info_out) std::memcpy(info_out, info.data(), sizeof(pantax_hip_solve_info) * S);
    std::vector<uint8_t> reported(S);
    for (uint32_t s = 0; s < S; ++s) reported[s] = (keep_out[s] && info[s].status1 == 0 && info[s].status2 == 0) ? 1 : 0;
    return pantax_hip_abundance_filter(S, db->h_hap_off.data(), met, reported.data(), cfg.single_cov_diff, cfg.min_cov, pass_out, nullptr, nullptr,
                                       species_sum_all_out, species_sum_pass_out);
}
}  // namespace

// The per-run index build of the COMING step, started now: on the side stream behind the last reader of the index it replaces,
// joined into the main stream at once -- whatever the caller enqueues next (the load of that run's reads, the step) is ordered behind
// it, while copies on other streams and the host-side work of a load run beside it.
extern "C" int pantax_hip_trio_index_prefetch(pantax_hip_ctx *ctx, pantax_hip_db *db) {
    if (!ctx || !db) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    if (db->d_node_rec.p == nullptr) return fail(ctx, PANTAX_HIP_E_STATE, "trio_index_prefetch: the db was uploaded without graphs (ranges only)");
    db->trio_built = false; db->cov_done = false; db->U = 0; db->trio_prefetched = false;
    if (db->trio_free_valid && !ctx->cfg.trio_after_step) PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, db->ev_trio_free, 0));
    else {
        PTX_HIP(ctx, hipEventRecord(ctx->ev_seq, ctx->stream));
        PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_seq, 0));
    }
    hipStream_t main_stream = ctx->stream;
    ctx->stream_main = main_stream; ctx->stream = ctx->stream2;
    const int rc = trio_index_build(ctx, db, false);
    const hipError_t e = hipEventRecord(ctx->ev_fork, ctx->stream2);
    ctx->stream = main_stream; ctx->stream_main = nullptr;
    if (rc != 0 || e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream2); db->trio_built = false; if (rc != 0) return rc; PTX_HIP(ctx, e); }
    PTX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
    db->trio_prefetched = true;
    return 0;
}

extern "C" int pantax_hip_profile_step_enqueue(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads, const double *avg_len,
                                               const pantax_hip_step_config *cfg) {
    if (!ctx || !db || !reads || !avg_len || !cfg) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    return step_enqueue(ctx, db, reads, avg_len, cfg);
}

extern "C" int pantax_hip_profile_step_collect(pantax_hip_ctx *ctx, pantax_hip_db *db, uint8_t *keep_out, double *absolute_out,
                                               pantax_hip_hap_metrics *met, pantax_hip_solve_info *info_out, uint8_t *pass_out,
                                               double *species_sum_all_out, double *species_sum_pass_out) {
    if (!ctx || !db || !keep_out || !absolute_out || !met || !pass_out) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    return step_collect(ctx, db, keep_out, absolute_out, met, info_out, pass_out, species_sum_all_out, species_sum_pass_out);
}

extern "C" int pantax_hip_profile_step(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads, const double *avg_len,
                                       const pantax_hip_step_config *cfg, uint8_t *keep_out, double *absolute_out,
                                       pantax_hip_hap_metrics *met, pantax_hip_solve_info *info_out, uint8_t *pass_out,
                                       double *species_sum_all_out, double *species_sum_pass_out) {
    if (!ctx || !db || !reads || !avg_len || !cfg || !keep_out || !absolute_out || !met || !pass_out) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    if (db->step_inflight) return fail(ctx, PANTAX_HIP_E_STATE, "profile_step: %d enqueued step(s) of this db have not been collected", db->step_inflight);
    PTX_TRY(step_enqueue(ctx, db, reads, avg_len, cfg));
    return step_collect(ctx, db, keep_out, absolute_out, met, info_out, pass_out, species_sum_all_out, species_sum_pass_out);
}
