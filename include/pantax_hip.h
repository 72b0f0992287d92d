/*
 * pantax_hip.h -- C ABI of the MI355X-native PanTax profiling path.
 *
 * This is the drop-in boundary: a Rust `extern "C"` block
 * (or cgo / ctypes) binds to exactly these symbols.  Plain pointers and sizes
 * only; every array is caller-owned and only read unless marked [out].
 * Conventions: 0 = success, negative = pantax_hip_status; nothing aborts or
 * throws across the boundary; one ctx per process per GPU; calls on one ctx are
 * serialised on its HIP stream.
 *
 * Reference seams this replaces (paths relative to pantax/src):
 *   - pipeline seam:  profile::profile(ProfilingConfig)            profile.rs:3325 (called main.rs:51-54)
 *   - solver seam:    match args.solver { "gurobi" | "highs" ... } profile.rs:2969-3009
 *                     fn X_opt(&mut GurobiOptVar, nvert, paths, node_abundance_vec,
 *                              node_base_cov, node_len, args)        profile.rs:2690-2698
 *   - histogram seam: get_node_abundances(...)                      profile.rs:743-750
 *   - binning seam:   rcls::rcls_profile / process_single_read_simple  rcls.rs:452-458, 237-258
 * INTEGRATION.md shows the Rust-side binding for each.
 */
#ifndef PANTAX_HIP_H
#define PANTAX_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PANTAX_HIP_OK = 0,
    PANTAX_HIP_E_INVALID = -1,       /* bad argument */
    PANTAX_HIP_E_HIP = -2,           /* HIP runtime error (message in last_error) */
    PANTAX_HIP_E_NO_DEVICE = -3,     /* no usable gfx950 device: the product never falls back to CPU */
    PANTAX_HIP_E_LIMIT = -4,         /* size limit of this build (32-bit node / step / row positions; > 30000 haplotypes in one species) */
    PANTAX_HIP_E_SOLVER = -5,        /* LP did not reach optimality (reference: Err(e) => species dropped, profile.rs:2999-3003) */
    PANTAX_HIP_E_IO = -6,            /* file missing / malformed (pipeline seam) */
    PANTAX_HIP_E_STATE = -7          /* stage called before its prerequisite */
} pantax_hip_status;

typedef struct pantax_hip_ctx pantax_hip_ctx;     /* owns the HIP stream, scratch, timers */
typedef struct pantax_hip_db pantax_hip_db;       /* device-resident graphs of the species this rank owns */
typedef struct pantax_hip_reads pantax_hip_reads; /* device-resident packed alignment records */

/* Threading: every entry point may be called from any host thread; calls that share a ctx are serialised inside
 * (the reference calls its solver from rayon workers, profile.rs:3297-3304).  Error text is per calling thread. */
/* ONE ctx drives ONE GPU, and a process holds one ctx per GPU it uses: the design is one process per GPU (a Rust host starts N
 * processes, or N threads with one ctx each).  device_ids / n_devices keep SURVEY 8b's signature; n_devices must be 1, anything
 * else is PANTAX_HIP_E_INVALID.  The environment is read HERE and nowhere else: every PANTAX_<OPTION> variable sets the option of
 * that name (lower case) once; afterwards options change only through pantax_hip_set_option -- no entry point calls getenv on its
 * way, so a host thread that changes the environment beside a running call cannot race with the library. */
int pantax_hip_init(pantax_hip_ctx **out, const int *device_ids, int n_devices);
void pantax_hip_destroy(pantax_hip_ctx *ctx);
/* options of a ctx: "hip_trace" (phase times on stderr), "stage_threads" (host threads that fill the pinned upload ring),
 * "gaf_piece_bytes", and the switches that force one of the in-tree HIP paths for tests and measurements (pantax_amd/csrc/common.hpp
 * CtxConfig lists them).  value NULL = the default.  PANTAX_HIP_E_INVALID for an unknown name or an unparsable value. */
int pantax_hip_set_option(pantax_hip_ctx *ctx, const char *name, const char *value);
const char *pantax_hip_last_error(const pantax_hip_ctx *ctx); /* ctx may be NULL: init errors */
const char *pantax_hip_version(void);

/* ---- DB side: types.rs:51-55 `Graph` of every species, concatenated ---------------------- */
typedef struct {
    uint32_t n_species;
    const int64_t *range_start; /* [S] first global node id, 1-based (species_range.txt col 2) */
    const int64_t *range_end;   /* [S] last global node id (col 3) */
    const uint64_t *node_off;   /* [S+1] species s owns node_len[node_off[s] .. node_off[s+1]) */
    const int64_t *node_len;    /* [V] Graph::nodes_len */
    const uint64_t *hap_off;    /* [S+1] species s owns haplotypes [hap_off[s] .. hap_off[s+1]) in BTreeMap (byte) order */
    const uint64_t *path_off;   /* [H+1] walk of hap h = path_nodes[path_off[h] .. path_off[h+1]) */
    const uint32_t *path_nodes; /* [P] species-local 0-based node ids (Graph::paths values) */
} pantax_hip_graphs;

int pantax_hip_db_upload(pantax_hip_ctx *ctx, const pantax_hip_graphs *g, pantax_hip_db **out);
/* the same db from one `Graph` per species as the host holds them after load_from_zip_graph (zip.rs:236-283) -- nothing is
 * concatenated on the host: the arrays travel species by species through one pinned chunk pipeline, lengths and walks are checked
 * on the device (length > 0, profile.rs:494; every walk inside its graph, :849) */
typedef struct {
    uint64_t n_nodes, n_haps;
    const int64_t *node_len;    /* [n_nodes] Graph::nodes_len */
    const uint64_t *path_off;   /* [n_haps+1] local CSR of the walks, haplotypes in BTreeMap (byte) order */
    const uint32_t *path_nodes; /* [path_off[n_haps] - path_off[0]] species-local 0-based node ids */
} pantax_hip_graph_part;
int pantax_hip_db_upload_parts(pantax_hip_ctx *ctx, uint32_t n_species, const int64_t *range_start, const int64_t *range_end,
                               const pantax_hip_graph_part *parts, pantax_hip_db **out);
void pantax_hip_db_free(pantax_hip_ctx *ctx, pantax_hip_db *db);

/* ---- read side: packed form of the GAF columns rcls.rs:127-137 selects ------------------- */
#define PANTAX_HIP_READ_NULLFIELD 1u /* a selected column was `*`: dropped at strain level, profile.rs:380-399 */
#define PANTAX_HIP_READ_DUPDROP 2u   /* duplicate read id spanning >1 species, profile.rs:406-437 */
typedef struct {
    uint64_t n_reads;
    uint64_t n_steps;
    const uint32_t *step_off; /* [R+1] */
    const uint32_t *node_id;  /* [T] node ids exactly as written in GAF col 6 */
    const uint32_t *pstart;   /* [R] GAF col 8 (read_start) */
    const uint32_t *pend;     /* [R] GAF col 9 (read_end) */
    const uint32_t *qlen;     /* [R] GAF col 2 (read_len) */
    const uint8_t *mapq;      /* [R] GAF col 12; 255 = null */
    const uint8_t *flags;     /* [R] PANTAX_HIP_READ_* or NULL */
} pantax_hip_packed_reads;

int pantax_hip_reads_upload(pantax_hip_ctx *ctx, const pantax_hip_packed_reads *r, pantax_hip_reads **out);
void pantax_hip_reads_free(pantax_hip_ctx *ctx, pantax_hip_reads *reads);

/* ---- a2 + a3: read -> species binning (rcls.rs:237-258) and the species counters
 * (profile.rs:208-297).  species_idx_out[r] = index into the db's species or -1 ("U").
 * Counter arrays are [n_species]; any out pointer may be NULL. */
int pantax_hip_bin_reads(pantax_hip_ctx *ctx, const pantax_hip_db *db, pantax_hip_reads *reads,
                         int32_t *species_idx_out, int64_t *read_count_out, int64_t *base_sum_out,
                         int64_t *less_multi_out, int64_t *uniq_count_out);

/* a3 finishing (species_profiling, profile.rs:299-349): equal-length test on the first 1000 reads
 * with species != "U" (:312-319), the MAPQ filter when `filtered` (:239-245), absolute =
 * base_count / avg_len (:336), abundance = absolute / sum (:341).  avg_len[s] <= 0 = species missing
 * from species_genomes_stats.txt.  Outputs [n_species]; rows are in db order (callers sort by
 * abundance, :344).  Requires bin_reads on `reads`. */
int pantax_hip_species_profile(pantax_hip_ctx *ctx, const pantax_hip_db *db, pantax_hip_reads *reads,
                               const int64_t *read_count, const int64_t *base_sum, const int64_t *less_multi,
                               const int64_t *uniq_count, const double *avg_len, int filtered,
                               uint8_t *keep_out, double *absolute_out, double *abundance_out);

/* drop everything derived from the graphs (trio index, coverage state) so the next calls rebuild it:
 * the reference recomputes trio_nodes_info inside every optimize_otu call (profile.rs:2936). */
int pantax_hip_db_reset(pantax_hip_ctx *ctx, pantax_hip_db *db);

/* ---- a7: unique-trio index (profile.rs:658-740), built on device once per db.
 * Rows are ordered (species, hap, window position). */
int pantax_hip_trio_index(pantax_hip_ctx *ctx, pantax_hip_db *db, uint64_t *n_unique_total_out);
/* copy the table out: abc [3*U] canonical species-local keys, hap [U] hap index within
 * its species, len [U], hap_trio_off [H+1] (global hap numbering). NULLs are skipped. */
int pantax_hip_trio_get(pantax_hip_ctx *ctx, const pantax_hip_db *db, uint32_t *abc_out, uint32_t *hap_out,
                        int64_t *len_out, uint64_t *hap_trio_off_out);

/* ---- a8: get_node_abundances integer part (profile.rs:743-1026).  Requires bin_reads
 * (and trio_index if trio_bases_out != NULL).  species_active: [S] 0/1 or NULL = all
 * (the species load_species_range keeps, profile.rs:553-656).  Outputs: [V], [V], [U].
 * n_abort_out counts reads on which the reference would abort (assert profile.rs:854 /
 * index panic :849); they contribute nothing. */
int pantax_hip_node_coverage(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads,
                             const uint8_t *species_active, int64_t *bases_per_node_out,
                             uint64_t *node_base_cov_out, int64_t *trio_bases_out, uint64_t *n_abort_out);

/* ---- a9..a14: strain level for every active species (optimize_otu profile.rs:2884-3026,
 * abundace_constraint :3028-3070).  Requires node_coverage to have run on (db, reads). */
#define PANTAX_HIP_HAS_FRACTION 1u
#define PANTAX_HIP_HAS_FREQ_MEAN 2u
#define PANTAX_HIP_HAS_RATIO 4u
#define PANTAX_HIP_HAS_FIRST 8u
#define PANTAX_HIP_HAS_DIVERGENCE 16u
#define PANTAX_HIP_HAS_SECOND 32u
#define PANTAX_HIP_HAS_RESCUE 64u
#define PANTAX_HIP_HAS_TOTAL_DIFF 128u
typedef struct { /* HapMetrics, profile.rs:1065-1078; `has` bit = Option::is_some() */
    uint32_t has;
    int32_t is_rescue;
    double unique_trio_nodes_fraction, frequencies_mean, path_cov_ratio, first_sol, divergence, second_sol,
        total_cov_diff;
} pantax_hip_hap_metrics;

typedef struct { /* the ProfilingConfig fields optimize_otu reads (types.rs:57-91, defaults main.rs:102-171) */
    double unique_trio_nodes_fraction;     /* --fr: 0.3 short / 0.5 long */
    double unique_trio_nodes_mean_count_f; /* --fc 0.46 */
    double single_cov_ratio;               /* --sr 0.85 */
    int64_t min_depth;                     /* --min_depth 0 */
    int32_t shift;                         /* --shift */
    int32_t sample_nodes;                  /* --sample (cli.rs:227 default 500000; pass 500 for --sample_test): a species with more valid
                                            * LP rows keeps the rows of sample_sorted (profile.rs:1287-1295); 0 = never sample */
    int32_t solver_semantics;              /* which backend's handling of the SECOND solve's solution is reproduced (the LP and its optimum are the same for all):
                                            * PANTAX_HIP_SEMANTICS_GUROBI (0, default; also cplex / cbc / glpk): every candidate that survives
                                            * second_filter_paths takes its own x of the second solve (profile.rs:1500-1508);
                                            * PANTAX_HIP_SEMANTICS_HIGHS (1): highs_opt first cuts the solution to its first K columns, K = number of
                                            * survivors, and zips THAT with the candidates (profile.rs:2865-2879) -- a survivor at candidate position
                                            * >= K is left without a second_sol (it then counts as 0, :3036, and is dropped from the table, :3237).
                                            * What a maintainer without a Gurobi licence can diff against is `--solver highs`: this switch makes that
                                            * diff come out empty. */
} pantax_hip_strain_config;
#define PANTAX_HIP_SEMANTICS_GUROBI 0
#define PANTAX_HIP_SEMANTICS_HIGHS 1

typedef struct { /* per species solver report */
    int32_t n_candidates, status1, status2, iters1, iters2;
    uint32_t n_rows, n_patterns;
    double obj1, obj2;
} pantax_hip_solve_info;

int pantax_hip_strain_profile(pantax_hip_ctx *ctx, pantax_hip_db *db, const pantax_hip_strain_config *cfg,
                              const uint8_t *species_active /*[S] or NULL*/,
                              const double *species_coverage /*[S] predicted_coverage, profile.rs:3044-3047*/,
                              pantax_hip_hap_metrics *metrics_out /*[H]*/, pantax_hip_solve_info *info_out /*[S] or NULL*/);

/* a15 core (abundance_est, profile.rs:3219-3245), host only: pass_out[h] = (group_size > 1 ||
 * total_cov_diff <= single_cov_diff) && predicted_coverage >= min_cov && predicted_coverage != 0;
 * sum_all_out = sum of every non-null predicted_coverage (the :3198 normaliser), sum_pass_out = the
 * same over passing rows (:3243).  These two numbers are the only cross-GPU reduction of the path. */
int pantax_hip_abundance_filter(uint32_t n_species, const uint64_t *hap_off, const pantax_hip_hap_metrics *metrics,
                                const uint8_t *species_reported /*[S] 0 = species dropped (solver error / inactive)*/,
                                double single_cov_diff, int64_t min_cov, uint8_t *pass_out,
                                double *sum_all_out, double *sum_pass_out,
                                double *species_sum_all_out /*[S] or NULL*/, double *species_sum_pass_out /*[S] or NULL*/);

/* ---- SURVEY 8e, reads over N GPUs: bin where tokenised, route to the owner of the species ------------------------
 * The reference groups the reads by species in one process (group_reads_by_species, profile.rs:439-463) and hands each
 * species' records to its rayon task.  With one process per GPU every rank holds a 1/N slice of the reads (its byte range
 * of the GAF): it bins the slice against ALL species ranges (pantax_hip_bin_reads), packs one message per owner rank
 * (route_pack: stable partition on the device, order of the slice kept; "U" reads, reads of species nobody owns and reads
 * with a drop flag are left behind), the host moves the messages (RCCL all-to-all(v) over xGMI on the device buffers, MPI,
 * ...), and the owner builds resident reads from what it received (reads_from_routed: messages in source-rank order, so the
 * result is the one-process read order restricted to the owner's species).
 * Message layout, 32-bit words: n_steps[n] pstart[n] pend[n] qlen[n] mapq[n] node_id[n_steps_total]. */
typedef struct pantax_hip_route pantax_hip_route;
int pantax_hip_reads_route_pack(pantax_hip_ctx *ctx, const pantax_hip_db *db /* the db `reads` were binned against */,
                                const pantax_hip_reads *reads, const int32_t *owner_of_species /*[S] rank, or < 0: nobody*/,
                                int world_size /* <= 64 */, pantax_hip_route **out, uint64_t *n_reads_to /*[W] out*/,
                                uint64_t *n_steps_to /*[W] out*/);
/* the W messages, back to back in rank order: buf_out[word_off_out[d] .. word_off_out[d+1]) goes to rank d
 * (word_off_out [W+1], may be NULL: 5 * n_reads_to + n_steps_to each).  on_device = 1: device pointer (valid until
 * route_free), 0: pinned host copy. */
int pantax_hip_route_buffer(pantax_hip_ctx *ctx, pantax_hip_route *route, int on_device, const uint32_t **buf_out,
                            uint64_t *word_off_out);
void pantax_hip_route_free(pantax_hip_ctx *ctx, pantax_hip_route *route);
/* recv: the messages of ranks 0..W-1 for this rank, back to back (device pointer when on_device, else host memory);
 * n_reads_from / n_steps_from [W] as announced by the senders.  The reads come out ready for pantax_hip_bin_reads. */
int pantax_hip_reads_from_routed(pantax_hip_ctx *ctx, const uint32_t *recv, int on_device, int world_size,
                                 const uint64_t *n_reads_from, const uint64_t *n_steps_from, pantax_hip_reads **out);

/* ---- resident step: the in-memory core of profile::profile (profile.rs:3325-3364) between "GAF parsed"
 * and "tables written", over a db and reads that are already in HBM.  One call = rcls_profile ->
 * species_profiling -> trio_nodes_info (when rebuild_trio) -> get_node_abundances -> strain_profiling ->
 * the abundance_est filters; the stages above remain available one by one and give identical results.
 * The host waits once, at the end.  Outputs are the LOCAL quantities of this rank's species: the global
 * normalisers (profile.rs:341, :3198, :3243) are sums of absolute_out / species_sum_*_out over ranks. */
typedef struct {
    double unique_trio_nodes_fraction, unique_trio_nodes_mean_count_f, single_cov_ratio, single_cov_diff; /* --fr --fc --sr --sd */
    int64_t min_cov, min_depth;
    int32_t shift, filtered, sample_nodes /* as in pantax_hip_strain_config */, rebuild_trio /* 1 = like the reference, every run */;
    int32_t solver_semantics;              /* as in pantax_hip_strain_config */
} pantax_hip_step_config;

int pantax_hip_profile_step(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads,
                            const double *avg_len /*[S] species_genomes_stats.txt*/, const pantax_hip_step_config *cfg,
                            uint8_t *keep_out /*[S] species kept by species_profiling*/, double *absolute_out /*[S] predicted_coverage*/,
                            pantax_hip_hap_metrics *metrics_out /*[H]*/, pantax_hip_solve_info *info_out /*[S] or NULL*/,
                            uint8_t *pass_out /*[H]*/, double *species_sum_all_out /*[S] or NULL*/,
                            double *species_sum_pass_out /*[S] or NULL*/);

/* The same step in two halves, for a caller that streams samples: `enqueue` puts the whole step on the device and returns
 * without waiting; `collect` takes the OLDEST enqueued step of the db: its one host wait, the reporting arithmetic and the
 * a15 filters (outputs as above).  Up to two steps of a db may be in flight, so step i+1 can be enqueued before step i is
 * collected and the device never waits for the host between two steps.  On the device the main-stream work of the steps runs
 * one step after the other; only the unique-trio rebuild of step i+1 (side stream, rebuild_trio != 0) may start earlier:
 * behind the first filter of step i -- the last reader of the index -- beside step i's masks, row sort and LPs, which read
 * copies.  Every step computes what the one-call form computes; `reads` / `avg_len` of an enqueued step must stay valid until its collect (avg_len is copied at enqueue). */
int pantax_hip_profile_step_enqueue(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads, const double *avg_len,
                                    const pantax_hip_step_config *cfg);
int pantax_hip_profile_step_collect(pantax_hip_ctx *ctx, pantax_hip_db *db, uint8_t *keep_out, double *absolute_out,
                                    pantax_hip_hap_metrics *metrics_out, pantax_hip_solve_info *info_out, uint8_t *pass_out,
                                    double *species_sum_all_out, double *species_sum_pass_out);

/* A resident db that serves one sample after the other: the unique-trio index of the COMING run (trio_nodes_info, profile.rs:2936 --
 * it depends on the graphs only) is started now, e.g. right before that run's GAF is loaded, so that it is built beside the PCIe
 * transfer instead of in front of the coverage pass.  The next profile_step / _enqueue of the db with rebuild_trio != 0 uses it instead
 * of building again (one build per run, as in the reference); nothing is waited for here. */
int pantax_hip_trio_index_prefetch(pantax_hip_ctx *ctx, pantax_hip_db *db);

/* ---- the device sort of the LP row grouping as a host-buffer utility: rows (k0[i], k1[i], k2[i]) sorted
 * ascending as tuples, in place.  algo: 0 = what the strain step would pick for n rows, 1 = LSD radix sort,
 * 2 = sample sort (n <= 600000), 4 = the batched sort of the many-species step: entries arrive grouped by ascending k0
 * (one segment per k0 value, of any size), (k1, k2) is sorted inside every segment, straight from the node arrays: an entry
 * whose k1 is 0 or whose k2 is not the bit pattern of a positive double is no row -- the rows come back
 * sorted in the first entries, the rest of the three arrays is zeroed; 5 = 4 with k1 < 256 and the segment number packed into
 * the mask word on the device (the step's two-word records).  (3, round 3's segmented sort, is gone: PANTAX_HIP_E_INVALID.) */
int pantax_hip_sort_rows(pantax_hip_ctx *ctx, uint64_t n, uint64_t *k0, uint64_t *k1, uint64_t *k2, int algo);

/* filter_max_alignment_mt (gaf_filter.rs:44-97, called by alignment.rs:171 on long-read GAFs): per read id
 * keep the line with the largest (matches, identity) if it also has mapq > 20 and span > 1000; one line per id.
 * Parsed and grouped on the device; the kept lines are written in FILE ORDER (the reference's order and its choice
 * among equal-best lines are rayon scheduling accidents; here: the first such line).  out_path NULL =
 * "<stem>_filtered.gaf" beside the input.  Counters may be NULL. */
int pantax_hip_gaf_filter(pantax_hip_ctx *ctx, const char *gaf_path, const char *out_path, uint64_t *n_lines,
                          uint64_t *n_records, uint64_t *n_written);

/* device-ready images of the species of a resident db, one file per species (graph in the kernels' layouts +
 * its unique-trio index): written from a db (the index is built first if needed; hap_names[H] in db order) and read
 * back into a db whose trio index is already in place -- a6 becomes "map + copy", a7 a no-op. */
int pantax_hip_db_save_images(pantax_hip_ctx *ctx, pantax_hip_db *db, const char *const *paths /*[S]*/,
                              const char *const *hap_names /*[H]*/);
int pantax_hip_db_load_images(pantax_hip_ctx *ctx, uint32_t n_species, const char *const *paths, const int64_t *range_start,
                              const int64_t *range_end, pantax_hip_db **out);

/* a11 (sample_sorted, profile.rs:1287-1295): which of n_valid rows `StdRng::seed_from_u64(seed)` +
 * `choose_multiple(sample_nodes)` keeps, as a bitmap over their ranks (bits_out: (n_valid+31)/32 words).  Host only.
 * rand 0.9.2 / rand_chacha 0.9.0 (Cargo.lock) are restated, not linked: parity with the crates is unpinned. */
int pantax_hip_sample_ranks(uint64_t n_valid, uint64_t sample_nodes, uint64_t seed, uint32_t *bits_out);
/* one ChaCha block (64-bit counter, zero stream id, `rounds` = 8/12/20) of the generator above, for known-answer tests */
int pantax_hip_chacha_block(const uint32_t *key8, uint64_t counter, int rounds, uint32_t *out16);

/* ---- solver seam: one species, host buffers in, same meaning as X_opt's arguments
 * (profile.rs:2690-2698).  cand_path_idx = possible_paths_idx; fixed_zero[k]=1 pins x_k = 0
 * (second solve, profile.rs:1484-1488).  x_out [n_cand]; path_cov_ratio_out [n_cand] or NULL. */
int pantax_hip_pao_solve(pantax_hip_ctx *ctx, uint32_t n_nodes, const int64_t *node_len,
                         const double *node_abundance, const uint64_t *node_base_cov, uint32_t n_paths,
                         const uint64_t *path_off, const uint32_t *path_nodes, uint32_t n_cand,
                         const uint32_t *cand_path_idx, const uint8_t *fixed_zero, double *x_out,
                         float *path_cov_ratio_out, double *obj_out, int32_t *status_out);

/* The same seam for MANY species in one call (SURVEY.md 8b "pao_solve_batch": what the GPU wants -- one workgroup per
 * species, all LPs side by side).  A Rust caller that keeps its own first filter (profile.rs:2969-3009 inside the rayon loop
 * of :3297-3319) collects the arguments of its X_opt calls as arrays of offsets and gets every species' answer back.
 * Species s owns nodes [node_off[s], node_off[s+1]), haplotypes [hap_off[s], hap_off[s+1]) and candidates
 * [cand_off[s], cand_off[s+1]); path_off indexes path_nodes over all haplotypes of the batch; cand_path_idx is the
 * haplotype's index WITHIN its species.  Any number of candidates, as in the reference (dense nvert x npaths matrix,
 * profile.rs:1333-1342): 1..64 take the one-word path, 65..256 four mask words, more than 256 haplotypes a solver whose mask words,
 * basis inverse and column state are sized at run time.  status[s]: 0 solved (or nothing to solve: no candidates),
 * PANTAX_HIP_E_SOLVER -- per species, like the reference drops only the species whose solver failed
 * (profile.rs:2999-3003); the call itself fails only on invalid arguments or a HIP error. */
typedef struct {
    uint32_t n_species;
    const uint64_t *node_off;       /* [S+1] */
    const int64_t *node_len;        /* [V] */
    const double *node_abundance;   /* [V] */
    const uint64_t *node_base_cov;  /* [V] or NULL (only path_cov_ratio needs it) */
    const uint64_t *hap_off;        /* [S+1] */
    const uint64_t *path_off;       /* [H+1] */
    const uint32_t *path_nodes;     /* [P] species-local 0-based node ids */
    const uint64_t *cand_off;       /* [S+1] */
    const uint32_t *cand_path_idx;  /* [C] possible_paths_idx of every species, concatenated */
    const uint8_t *fixed_zero;      /* [C] or NULL: 1 pins x_k = 0 (second solve, profile.rs:1484-1488) */
} pantax_hip_species_batch;
typedef struct {
    double *x;              /* [C] out */
    float *path_cov_ratio;  /* [C] out or NULL */
    double *obj;            /* [S] out or NULL */
    int32_t *status;        /* [S] out */
    int32_t *iters;         /* [S] out or NULL: pivots of the active-set solver */
} pantax_hip_solution_batch;
int pantax_hip_pao_solve_batch(pantax_hip_ctx *ctx, const pantax_hip_species_batch *in, const pantax_hip_solution_batch *out);

/* ---- pipeline seam: files in, files out (profile.rs:3325) -------------------------------- */
typedef struct { /* ProfilingConfig (types.rs:57-91) as plain C; NULL path = reference default under db/wd */
    const char *db, *wd, *output_dir;
    const char *genomes_metadata, *range_file, *input_aln_file, *species_len_file, *out_binning_file, *reads_binning_file;
    double min_species_abundance, unique_trio_nodes_fraction, unique_trio_nodes_mean_count_f, single_cov_ratio,
        single_cov_diff;
    int64_t min_cov, min_depth;
    int32_t species, strain, shift, filtered, full, force, mode, sample_nodes;
    const char *designated_species; /* --ds or NULL */
    const char *zip;                /* "serialize" (.bin) | "lz" (.bin.lz4) | "zstd" (.bin.zst) | NULL (= GFA); "h5" is refused */
    /* one process per GPU: with world_size > 1 this process takes its share of the selected species (longest-processing-
     * time packing on reads + graph size, the same table on every rank) and rank 0 writes the tables.
     * With `alltoallv` set the INPUT is sharded too: rank r tokenises and bins only its line-aligned 1/N byte range
     * of the GAF, the species counters are summed over the ranks, the duplicate-id rule (profile.rs:361-437) is decided on
     * (id hash, species) records exchanged by hash, and the packed records of every read travel to the rank that owns its
     * species (pantax_hip_reads_route_pack / _from_routed) -- no rank ever reads the whole file.  Without it every rank
     * tokenises the whole GAF (simple, but N x the ingest).
     * Collectives per run: allreduce_sum a handful of times (run mode; {failure flag, counters}; {failure flag, the two
     * normalisers}; barriers around the part files) -- every rank-local failure is carried in such a flag, so the ranks
     * always leave together -- and, when sharded, alltoallv two to three times (id records; packed reads; the ids to drop,
     * only if some id does span species). */
    int32_t rank, world_size;
    /* device-ready graph images <db>/species_graph_info/<otu>.hipdb (graphs + unique-trio index):
     * 0 = ignore them, 1 = use them when every selected species has a fresh one, 2 = as 1, and write them after a run
     * that had to parse the graphs */
    int32_t image_cache;
    /* world_size > 1: in-place sum over all ranks of buf[0..n) (e.g. ncclAllReduce + stream sync, MPI_Allreduce); 0 = ok */
    int (*allreduce_sum)(void *user, double *buf, uint64_t n);
    void *comm_user;
    /* world_size > 1, optional (NULL = unsharded input): bytes send[send_off[j] .. send_off[j+1]) of rank i arrive as
     * recv[recv_off[i] .. recv_off[i+1]) on rank j; both offset arrays have world_size + 1 entries and every rank passes
     * recv offsets that match what the others send (the library exchanges the sizes through allreduce_sum first).
     * E.g. ncclGroupStart + ncclSend/ncclRecv per peer + ncclGroupEnd + stream sync, or MPI_Alltoallv.  0 = ok.
     * comm_device_buffers != 0: send / recv are DEVICE pointers of this ctx's GPU (RCCL moves HBM to HBM over xGMI);
     * 0: host pointers (the library stages through pinned memory). */
    int (*alltoallv)(void *user, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off);
    int32_t comm_device_buffers;
    int32_t sample_test;            /* --sample_test (cli.rs:230-232): sample_sorted keeps 500 rows whatever --sample says (profile.rs:1387-1393, :2738-2744) */
    int32_t solver_semantics;       /* --solver: PANTAX_HIP_SEMANTICS_HIGHS for "highs", PANTAX_HIP_SEMANTICS_GUROBI for every other backend (pantax_hip_strain_config) */
    double minimization_min_cov;    /* types.rs:72 (main.rs:150 sets 0; no CLI flag).  It only shifts the indicator rows z_i >= (x_i - this) / (2 max), and the
                                     * indicators are bound by nothing but sum z <= npaths (profile.rs:1374-1378): INERT at any value.  Mirrored for
                                     * completeness of the struct; negative or non-finite values are refused. */
} pantax_hip_profiling_config;

/* A selection whose graphs hold more path steps than one resident db addresses (2^32: BASELINE configs[4] on one GPU) goes through the device in
 * groups of species, one after the other, inside the call -- no limit on the size of the DB other than the device memory a single group needs. */
int pantax_hip_profile(pantax_hip_ctx *ctx, const pantax_hip_profiling_config *cfg);

/* ---- a1 / a6 host readers (no GPU needed): the file contracts of the pipeline seam, exposed so a
 * caller that keeps its own orchestration can still reuse the tokenizer and graph loaders ---------- */
typedef struct pantax_hip_gaf pantax_hip_gaf;     /* owns the packed arrays of one tokenised GAF */
/* load_gaf_file_lazy (rcls.rs:119-146): columns 1,2,6,7,8,9,12; '@' comment lines and empty lines skipped; "*" AND the
 * empty field = null (the reader's null_values / missing_is_null; => PANTAX_HIP_READ_NULLFIELD for cols 6-9, mapq 255,
 * qlen 0); a non-integer in an integer column reads as null; integers above 2^32 - 1 are clamped to it (32-bit packed
 * columns).  The rules are restated, independently of this library, in oracle/gaf_reader.py, which the tests compare
 * every tokenizer with.  err_out (may be NULL) receives a static/thread-local message on failure. */
int pantax_hip_gaf_load(const char *path, int n_threads, pantax_hip_gaf **out, const char **err_out);
/* the same tokenisation on the device (text uploaded once, five launches): identical arrays.  LIMIT of the
 * device readers (this entry, pantax_hip_reads_load_gaf, pantax_hip_gaf_filter): the text travels in pieces cut at line
 * ends -- a sixth of the text, 64 MiB .. 1 GiB, less only through PANTAX_GAF_PIECE_BYTES -- and a piece holds whole lines:
 * ONE LINE longer than 3.5 GiB (or than a lowered piece size) is refused with PANTAX_HIP_E_LIMIT.  A HiFi / ONT line with
 * a long cs tag is kilobytes to megabytes: far inside the default. */
int pantax_hip_gaf_load_device(pantax_hip_ctx *ctx, const char *path, pantax_hip_gaf **out);
int pantax_hip_gaf_view(const pantax_hip_gaf *gaf, pantax_hip_packed_reads *view_out);
/* file -> packed reads RESIDENT in HBM, tokenised on the device, ready for pantax_hip_bin_reads; the walks never
 * visit the host.  The text travels in pieces on an upload stream (pread into a pinned ring on a few host threads) while the
 * piece before is tokenised.  gaf_out (optional) receives the host-side columns (read_len, mapq, flags; its view has
 * node_id / step_off / pstart / pend = NULL); with gaf_out == NULL those columns are not brought back at all. */
int pantax_hip_reads_load_gaf(pantax_hip_ctx *ctx, const char *path, pantax_hip_reads **reads_out, pantax_hip_gaf **gaf_out);
/* replace the per-read drop flags of resident reads (a5: null fields, duplicate ids); NULL clears them.  The reads
 * must be binned again afterwards. */
int pantax_hip_reads_set_flags(pantax_hip_ctx *ctx, pantax_hip_reads *reads, const uint8_t *flags);
void pantax_hip_gaf_free(pantax_hip_gaf *gaf);

typedef struct pantax_hip_graph pantax_hip_graph; /* one species graph in `Graph` shape (types.rs:51-55) */
/* format 0 = GFA S/W/P lines (read_gfa, profile.rs:466-545), 1 = bincode-1 .bin (zip.rs:236-247),
 * 2 = .bin.lz4 (LZ4 frame), 3 = .bin.zst (zip.rs:250-265; liblz4.so.1 / libzstd.so.1 are bound at run time) */
int pantax_hip_graph_load(const char *path, int format, pantax_hip_graph **out, const char **err_out);
/* sizes: n_nodes, n_haps, n_steps; arrays are valid until graph_free; hap_names_out[i] NUL-terminated */
int pantax_hip_graph_view(const pantax_hip_graph *g, uint64_t *n_nodes, uint64_t *n_haps, const int64_t **node_len,
                          const uint64_t **path_off, const uint32_t **path_nodes, const char *const **hap_names);
void pantax_hip_graph_free(pantax_hip_graph *g);

/* the float text of the two tables (polars CsvWriter behind rcls.rs:409-420: shortest round-trip digits, "16.0" for integral
 * values; exemplar rows README.md:343, :354).  Host only.  Returns the length, or < 0. */
int pantax_hip_format_f64(double v, char *buf, size_t cap);

/* ---- measurement: HIP-event timings of kernels launched on the ctx stream ---------------- */
int pantax_hip_timing_enable(pantax_hip_ctx *ctx, int on);
int pantax_hip_timing_reset(pantax_hip_ctx *ctx);
/* name != NULL/"": bracket only launches of that kernel, or of several ("a|b") (the event pairs of ~100 launches per step cost
 * ~0.15 ms; a throughput measurement that also wants one kernel's duration times just that one) */
int pantax_hip_timing_filter(pantax_hip_ctx *ctx, const char *name);
/* fills up to cap entries; returns the number of distinct kernel names (or <0) */
int pantax_hip_timing_get(pantax_hip_ctx *ctx, int cap, const char **names_out, uint64_t *launches_out,
                          double *total_ms_out);
int pantax_hip_sync(pantax_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
