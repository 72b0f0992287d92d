// lad.hpp -- device-side pieces of the strain step (a9, a10, a12): per-hap trio statistics,
// candidate-path masks + path_cov_ratio, LP row grouping and the batched exact LAD solver.
#pragma once
#include "common.hpp"

namespace ptx {

// a9: per-hap unique-trio statistics (zscore_filter profile.rs:1028-1051; :1114-1147)
int hap_trio_stats_launch(Ctx *ctx, const Db *db, DevBuf<uint32_t> &d_ntrio_nz /*[H]*/, DevBuf<double> &d_mean /*[H]*/,
                          const uint8_t *d_active = nullptr /* device [S] or null: species the coverage pass skipped are not read */);
// node abundance + per-species stats
int node_stats_launch(Ctx *ctx, const Db *db, LadBatch *lb, int64_t min_depth, const uint8_t *d_active = nullptr);
// a11: species with more valid rows than sample_nodes keep the rows rand 0.9.2's choose_multiple(seed 42) would keep
int row_sample_apply(Ctx *ctx, const Db *db, LadBatch *lb, int64_t sample_nodes);
// a10: masks, ratios; then LP rows sorted and grouped into patterns
int lad_prepare(Ctx *ctx, const Db *db, LadBatch *lb, bool cand_on_device, int pmax_bound);
// a12: one workgroup per species (those with d_p[s] > 0 and need[s], when given); variables with fixed[s*64+k]
// are pinned to 0.  Writes x, obj, status, iters for the solved species.  Nothing is read back.
int lad_solve_launch(Ctx *ctx, const Db *db, LadBatch *lb, int pmax_bound, const uint8_t *d_need, const uint8_t *d_fixed, double *d_x,
                     double *d_obj, int32_t *d_status, int32_t *d_iters);
// the strain step's LP1 -> second filter -> LP2 (+ both objectives) as two launches
struct FilterCfg;
int lad_pair_launch(Ctx *ctx, const Db *db, LadBatch *lb, int pmax_bound, const FilterCfg &fc);
// a9 / a13 decisions on the device (the host redoes only the reporting arithmetic at the end of the step)
struct FilterCfg { double fr, fc, sr; int shift; };
int first_filter_launch(Ctx *ctx, const Db *db, LadBatch *lb, const uint8_t *d_active, const FilterCfg &fc);

// the strain step in two halves (api_strain.cpp): everything enqueued / the one wait + host reporting
int strain_prezero(Ctx *ctx, Db *db);   // optional, ahead of strain_enqueue: its two zero-fills, issued where the stream has slack
int strain_enqueue(Ctx *ctx, Db *db, const pantax_hip_strain_config *cfg, const uint8_t *d_active, int slot);   // slot: which pinned result buffer / event (0 or 1)
int strain_finish(Ctx *ctx, Db *db, const pantax_hip_strain_config *cfg, const uint8_t *species_active, const double *species_coverage,
                  pantax_hip_hap_metrics *met, pantax_hip_solve_info *info_out, void (*after_wait)(void *), void *after_wait_arg, int slot);
int trio_index_build(Ctx *ctx, Db *db, bool with_keys);

}  // namespace ptx
