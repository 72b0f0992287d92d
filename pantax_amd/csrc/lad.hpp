// lad.hpp -- device-side pieces of the strain step (a9, a10, a12): per-hap trio statistics,
// candidate-path masks + path_cov_ratio, LP row grouping and the batched exact LAD solver.
#pragma once
#include "common.hpp"

namespace ptx {

// a9: per-hap unique-trio statistics (zscore_filter profile.rs:1028-1051; :1114-1147)
int hap_trio_stats_launch(Ctx *ctx, const Db *db, DevBuf<uint32_t> &d_ntrio_nz /*[H]*/, DevBuf<double> &d_mean /*[H]*/);
// node abundance + per-species stats
int node_stats_launch(Ctx *ctx, const Db *db, LadBatch *lb, int64_t min_depth);
// a10: masks, ratios; then LP rows sorted and grouped into patterns
int lad_prepare(Ctx *ctx, const Db *db, LadBatch *lb);
// a12: solve every species in solve_list with bounds d_ub; writes d_x, d_obj, d_status, d_iters
int lad_solve_launch(Ctx *ctx, const Db *db, LadBatch *lb, const std::vector<int32_t> &solve_list);

}  // namespace ptx
