/*
 * pantax_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Single-threaded plain-C restatement of the PanTax profiling hot path
 * (reference: /root/reference/pantax/src/profile.rs, rcls.rs).
 * It exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * have something to check the HIP path against. Nothing under pantax_amd/ may
 * include, link or call it.
 *
 * PARITY STATUS: "parity unpinned" by the reference's own tests (it has none
 * for this path).  The integer paths are pinned by the
 * hand-computed micro-graph vectors in tests/golden/micro_*.json; the LP
 * (third-party arithmetic: Gurobi 11 / HiGHS 1.12 behind profile.rs:1312-1460,
 * 2754-2822) is pinned against SciPy-HiGHS 1.8.0 solutions committed under
 * tests/golden/lp_*.npz (generator: oracle/gen_golden.py).
 *
 * Every function cites the reference lines it restates.
 */
#ifndef PANTAX_ORACLE_H
#define PANTAX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* types.rs:51-55 `Graph`: nodes_len + paths (BTreeMap order = byte-wise hap name
 * order; the caller passes paths already in that order). Node ids are local,
 * 0-based (global id - range_start). */
typedef struct {
    uint32_t n_nodes;
    const int64_t *node_len;    /* [n_nodes] */
    uint32_t n_paths;
    const uint64_t *path_off;   /* [n_paths+1] */
    const uint32_t *path_nodes; /* [path_off[n_paths]] */
} orc_graph;

/* Unique-trio table (profile.rs:658-740). The reference's row order is the
 * iteration order of an FxHashSet (arbitrary); this restatement fixes the order
 * to (hap index, window position) which is what the HIP path emits too. */
typedef struct {
    uint64_t n_unique;
    uint32_t *abc;       /* [3*n_unique] canonical (w0<=w2) keys */
    uint32_t *hap;       /* [n_unique] the single hap that owns the trio */
    int64_t *len;        /* [n_unique] len(a)+len(b)+len(c) */
    uint64_t *hap_off;   /* [n_paths+1] rows of hap h are [hap_off[h],hap_off[h+1]) */
    /* lookup acceleration (not reference state): keys sorted, with row index */
    uint32_t *sorted_abc;
    uint64_t *sorted_row;
} orc_trio_table;

int orc_trio_index(const orc_graph *g, orc_trio_table *out);
void orc_trio_free(orc_trio_table *t);

/* rcls.rs:237-258 + 306-323: per-read min/max node id, first range in FILE ORDER
 * with start<=min && max<=end, else -1 ("U"). Empty path => (-1,-1) => "U". */
int orc_bin_reads(uint64_t n_reads, const uint64_t *step_off, const uint32_t *node_id,
                  uint32_t n_ranges, const int64_t *range_start, const int64_t *range_end,
                  int32_t *species_idx_out);

/* profile.rs:208-297: per-species counters over reads with species != "U".
 * out arrays are [n_ranges]. */
int orc_species_counts(uint64_t n_reads, const int32_t *species_idx, const int64_t *read_len,
                       const int64_t *mapq, uint32_t n_ranges,
                       int64_t *read_count, int64_t *base_sum, int64_t *less_multi, int64_t *uniq_count);

/* profile.rs:299-349 finishing: equal-length test on the first 1000 non-U rows,
 * MAPQ filter (`filtered`), absolute = base_count / avg_len, abundance = absolute / sum.
 * keep_out[s]=1 when species s survives; rows are NOT sorted here (callers sort
 * descending by abundance as profile.rs:344 does). avg_len[s] <= 0 means "missing". */
int orc_species_profile(uint64_t n_reads, const int32_t *species_idx, const int64_t *read_len,
                        uint32_t n_ranges, const int64_t *read_count, const int64_t *base_sum,
                        const int64_t *less_multi, const int64_t *uniq_count, const double *avg_len,
                        int filtered, uint8_t *keep_out, double *absolute_out, double *abundance_out);

/* profile.rs:743-1026 get_node_abundances, integer part. `range_start` is the
 * species' first global node id (1-based; optimize_otu does start-1 then
 * id-1-start, profile.rs:2886, :790). pstart/pend are GAF col 8/9.
 * Reads for which the reference would abort (assert at :854, or an index panic
 * at :849) are skipped whole and counted in *n_abort. */
int orc_node_coverage(const orc_graph *g, const orc_trio_table *trio, int64_t range_start,
                      uint64_t n_reads, const uint64_t *step_off, const uint32_t *node_id,
                      const int64_t *pstart, const int64_t *pend,
                      int64_t *bases_per_node /*[V]*/, uint64_t *node_base_cov /*[V]*/,
                      int64_t *trio_bases /*[U]*/, uint64_t *n_abort);

/* profile.rs:1028-1051 + 1114-1147: per-hap unique-trio statistics.
 * n_trio[h], n_nonzero[h], mean_filtered[h] (mean of |z|<3 non-zero abundances,
 * 0.0 when the z-score filter returns empty). */
int orc_hap_trio_stats(const orc_trio_table *t, uint32_t n_paths, const int64_t *trio_bases,
                       uint64_t *n_trio, uint64_t *n_nonzero, double *mean_filtered);

/* profile.rs:1333-1361 (== 2705-2729): 0/1 membership masks of the candidate
 * paths per node, and path_cov_ratio accumulated in f32 in ascending node order.
 * cand[k] = index into graph paths; row v of the matrix = ORC_NW(n_cand) words at mask + v * ORC_NW(n_cand), bit k & 63
 * of word k >> 6 set iff node v is on path cand[k] (one word per node up to 64 candidates; no cap on n_cand, like the
 * reference's dense matrix). */
#define ORC_NW(n_cand) ((uint32_t)(((n_cand) + 63u) / 64u ? ((n_cand) + 63u) / 64u : 1u))
int orc_path_masks(const orc_graph *g, uint32_t n_cand, const uint32_t *cand,
                   const uint64_t *node_base_cov, uint64_t *mask_out /*[V * ORC_NW(n_cand)]*/, float *ratio_out /*[n_cand]*/);

/* The PAO LP (profile.rs:1312-1460 Gurobi form; 2754-2822 HiGHS form):
 *   min (1/n) sum_{v: a_v>0} | sum_{k in mask_v} x_k - a_v |,  0 <= x_k <= ub_k
 * (binary indicators are inert).  rows = ALL nodes; the
 * function selects a_v > 0 itself (profile.rs:1380-1385).  ub_k = 1.05*max(a) or
 * 0.0 for variables fixed in the second solve (profile.rs:1484-1488).
 * Exact active-set (Bloomfield-Steiger / Barrodale-Roberts style) LAD descent in
 * f64. status: 0 optimal, 1 iteration limit. */
int orc_lad_solve(uint64_t n_nodes, const uint64_t *mask, const double *abund, uint32_t n_cand,
                  const double *ub, double *x_out, double *obj_out, int32_t *iters_out, int32_t *status_out);

/* objective of a given x (same definition), for checking foreign solutions */
double orc_lad_objective(uint64_t n_nodes, const uint64_t *mask, const double *abund, uint32_t n_cand,
                         const double *x);

/* ---- per-species strain metrics: optimize_otu (profile.rs:2884-3026) ---- */
#define ORC_HAS_FRACTION 1u
#define ORC_HAS_FREQ_MEAN 2u
#define ORC_HAS_RATIO 4u
#define ORC_HAS_FIRST 8u
#define ORC_HAS_DIVERGENCE 16u
#define ORC_HAS_SECOND 32u
#define ORC_HAS_RESCUE 64u
#define ORC_HAS_TOTAL_DIFF 128u

typedef struct {
    uint32_t has; /* ORC_HAS_* bits = Option::is_some() */
    double unique_trio_nodes_fraction, frequencies_mean, path_cov_ratio, first_sol, divergence, second_sol,
        total_cov_diff;
    int32_t is_rescue;
} orc_hap_metrics;

typedef struct {
    double unique_trio_nodes_fraction;     /* --fr, main.rs:108-114 */
    double unique_trio_nodes_mean_count_f; /* --fc 0.46 */
    double single_cov_ratio;               /* --sr 0.85 */
    int64_t min_depth;                     /* --min_depth 0 */
    int32_t shift;                         /* main.rs:119-124 */
    int32_t sample_nodes;                  /* --sample (cli.rs:227 default 500000; --sample_test = 500); 0 = off */
    int32_t solver_semantics;              /* 0: second solve's x handed out as Gurobi / cplex / cbc / glpk do (profile.rs:1500-1508); 1: as highs_opt does,
                                            * first cut to K = #survivors columns, then zipped with the candidates (profile.rs:2865-2879) */
} orc_strain_config;

/* a11: sample_sorted (profile.rs:1287-1295) = StdRng::seed_from_u64(seed) + slice::choose_multiple + sort.
 * rand 0.9.2 / rand_chacha 0.9.0 are NOT under /root/reference (Cargo.lock pins them); this is a restatement of
 * their published algorithm and is PARITY UNPINNED.  Writes the `amount` chosen positions of 0..length, ascending. */
int orc_sample_sorted_positions(uint32_t length, uint32_t amount, uint64_t seed, uint32_t *out);
/* one 64-byte ChaCha block (rounds = 20 / 12 / 8), 64-bit counter, zero stream id: for the keystream known answers */
void orc_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]);

/* first_filter_paths .. second solve for one species, given integer histogram
 * outputs. metrics_out is [n_paths]. n_candidates_out = possible_paths_idx.len().
 * Returns 0, or <0 if the LP failed (reference: species dropped, profile.rs:2999-3003). */
int orc_optimize_species(const orc_graph *g, const orc_trio_table *trio, const int64_t *bases_per_node,
                         const uint64_t *node_base_cov, const int64_t *trio_bases,
                         const orc_strain_config *cfg, orc_hap_metrics *metrics_out,
                         uint32_t *n_candidates_out, double *obj1_out, double *obj2_out);

/* profile.rs:3028-3070 abundace_constraint */
int orc_abundance_constraint(double species_coverage, uint32_t n_paths, orc_hap_metrics *metrics);

#ifdef __cplusplus
}
#endif

/* ---- filter_max_alignment_mt (gaf_filter.rs:44-97) ----
 * text = the whole GAF; lines end at '\n' (a '\r' before it is dropped, BufRead::lines).  keep_out[i] = 1 when raw
 * line i is written.  parse_line (gaf_filter.rs:21-42): trim, split on tabs, >= 16 fields, i32 fields 9/11/3/2, f64 after
 * the last ':' of field 15.  best per read id = max (matches, identity); written: mapq > 20, span > 1000, == best,
 * one line per id -- the FIRST such line in file order (the reference's pick among ties and its output order depend
 * on rayon's scheduling).  Returns the number of lines, or -1. */
int64_t orc_gaf_filter(const char *text, uint64_t size, uint8_t *keep_out, uint64_t *n_records_out);

#endif
