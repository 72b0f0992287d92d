// lad.hpp -- device-side pieces of the strain step (a9, a10, a12): per-hap trio statistics,
// candidate-path masks + path_cov_ratio, LP row grouping and the batched exact LAD solver.
#pragma once
#include "common.hpp"

namespace ptx {

constexpr int LAD_MAXP = 64;  // candidate paths per species (one u64 membership mask per node)

// One batch = every species that has at least one candidate path, solved concurrently
// (one workgroup per species; the LPs are block-diagonal: profile.rs:3297-3319 runs them as
// independent rayon tasks).
struct LadBatch {
    uint32_t S = 0;
    // per species (host mirrors + device)
    std::vector<int32_t> h_p;           // [S] number of candidates (0 = not solved)
    std::vector<uint32_t> h_cand;       // [S*LAD_MAXP] candidate -> hap index within species
    DevBuf<int32_t> d_p;
    DevBuf<int32_t> d_hap_bit;          // [H] bit index of hap in its species' candidate list, -1 if none
    DevBuf<uint64_t> d_mask;            // [V] candidate membership mask per node (the 0/1 coeff matrix, row-wise)
    DevBuf<double> d_ab;                // [V] node_abundance = bases / len  (profile.rs:980-990)
    DevBuf<unsigned long long> d_ratio; // [S*LAD_MAXP*2] sum cov, sum len per candidate (exact integers)
    // per species node stats
    DevBuf<double> d_amax;              // [S] max node abundance (profile.rs:1316-1319)
    DevBuf<uint32_t> d_nvalid;          // [S] #nodes with abundance > 0 (= n_eval, profile.rs:1380-1385, :1447)
    DevBuf<double> d_nzsum;             // [S] sum of min_depth-filtered non-zero abundances (profile.rs:1193-1201)
    DevBuf<uint32_t> d_nzcnt;           // [S]
    // sorted LP rows (a_v > 0 and mask != 0), grouped into patterns
    uint64_t n_rows = 0;
    uint32_t K = 0;
    DevBuf<double> d_row_a;             // [n_rows] abundances sorted by (species, mask, a)
    DevBuf<uint64_t> d_pat_mask;        // [K]
    DevBuf<uint32_t> d_pat_start;       // [K+1]
    DevBuf<uint32_t> d_pat_species;     // [K]
    DevBuf<uint32_t> d_sp_pat_off;      // [S+1]
    std::vector<uint32_t> h_sp_pat_off;
    // solver scratch per pattern
    DevBuf<double> d_pat_eps, d_sc_s, d_sc_rho;
    DevBuf<uint32_t> d_sc_lo, d_sc_up, d_ls_lo, d_ls_hi, d_ls_mid;
    // solver in/out per species
    DevBuf<double> d_ub, d_x, d_obj;    // [S*LAD_MAXP], [S*LAD_MAXP], [S]
    DevBuf<int32_t> d_status, d_iters;  // [S]
    DevBuf<int32_t> d_solve_list;       // [n_solve]
};

// a9: per-hap unique-trio statistics (zscore_filter profile.rs:1028-1051; :1114-1147)
int hap_trio_stats_launch(Ctx *ctx, const Db *db, DevBuf<uint32_t> &d_ntrio_nz /*[H]*/, DevBuf<double> &d_mean /*[H]*/);
// node abundance + per-species stats
int node_stats_launch(Ctx *ctx, const Db *db, LadBatch *lb, int64_t min_depth);
// a10: masks, ratios; then LP rows sorted and grouped into patterns
int lad_prepare(Ctx *ctx, const Db *db, LadBatch *lb);
// a12: solve every species in solve_list with bounds d_ub; writes d_x, d_obj, d_status, d_iters
int lad_solve_launch(Ctx *ctx, const Db *db, LadBatch *lb, const std::vector<int32_t> &solve_list);

}  // namespace ptx
