/*
 * oracle_parallel.c -- TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
 *
 * A pthread driver around the single-threaded restatement in pantax_oracle.c, shaped like the reference's own
 * parallelism: reads are classified in parallel slices (rcls.rs:452-458 runs process_single_read_simple under rayon) and
 * the species are handed to worker threads one at a time, heaviest first (the rayon par_iter over species of
 * profile.rs:3297-3319; each worker does what optimize_otu does for its species: group the species' reads
 * (profile.rs:439-463), trio_nodes_info :658, get_node_abundances :743, the filters and both PAO solves :1028-1511,
 * abundace_constraint :3028).  No algorithm lives here: every number comes from the functions of pantax_oracle.c.
 * Nothing under pantax_amd/ may include, link or call it.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pantax_oracle.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------ binning in slices */
typedef struct {
    uint64_t r0, r1;
    const uint64_t *step_off;
    const uint32_t *node_id;
    uint32_t n_ranges;
    const int64_t *rs, *re;
    int32_t *out;
} bin_job;

static void *bin_worker(void *arg) {
    bin_job *j = (bin_job *)arg;
    /* orc_bin_reads indexes node_id through absolute step offsets: hand it the slice's offsets and the whole id array */
    orc_bin_reads(j->r1 - j->r0, j->step_off + j->r0, j->node_id, j->n_ranges, j->rs, j->re, j->out + j->r0);
    return NULL;
}

int orc_par_bin_reads(int n_threads, uint64_t n_reads, const uint64_t *step_off, const uint32_t *node_id, uint32_t n_ranges,
                      const int64_t *range_start, const int64_t *range_end, int32_t *species_idx_out) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    bin_job *jobs = (bin_job *)malloc(sizeof(bin_job) * (size_t)n_threads);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    for (int t = 0; t < n_threads; ++t) {
        jobs[t] = (bin_job){ n_reads * (uint64_t)t / (uint64_t)n_threads, n_reads * (uint64_t)(t + 1) / (uint64_t)n_threads, step_off, node_id, n_ranges,
                             range_start, range_end, species_idx_out };
        pthread_create(&th[t], NULL, bin_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

/* reads grouped by species (profile.rs:439-463 group_reads_by_species): stable counting sort of the read indices.
 * first[s] .. first[s] + cnt[s] index `order`; reads with species -1 ("U") are left out. */
int orc_group_reads(uint64_t n_reads, const int32_t *species_idx, uint32_t n_species, uint64_t *first /*[S+1]*/, uint64_t *order /*[n_reads]*/) {
    uint64_t *cur = (uint64_t *)calloc((size_t)n_species + 1, sizeof(uint64_t));
    if (!cur) return -1;
    for (uint64_t r = 0; r < n_reads; ++r) if (species_idx[r] >= 0) cur[species_idx[r] + 1]++;
    first[0] = 0;
    for (uint32_t s = 0; s < n_species; ++s) first[s + 1] = first[s] + cur[s + 1];
    for (uint32_t s = 0; s < n_species; ++s) cur[s] = first[s];
    for (uint64_t r = 0; r < n_reads; ++r) if (species_idx[r] >= 0) order[cur[species_idx[r]]++] = r;
    free(cur);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ species on worker threads */
typedef struct {
    uint32_t n_species;
    const orc_graph *graphs;          /* [S] */
    const int64_t *range_start;       /* [S] first global node id */
    const uint64_t *step_off;         /* reads (whole set or a prefix) */
    const uint32_t *node_id;
    const int64_t *pstart, *pend;
    const uint64_t *first, *order;    /* orc_group_reads */
    const uint8_t *keep;              /* [S] species that pass the species-level filter */
    const double *absolute;           /* [S] species coverage (abundace_constraint) */
    orc_strain_config cfg;
    uint32_t n_todo;
    const uint32_t *todo;             /* species in the order they are handed out */
    const uint64_t *hap_off;          /* [S+1] rows of metrics_out */
    /* outputs, [S] each */
    orc_hap_metrics *metrics_out;     /* [hap_off[S]] or NULL */
    int32_t *rc_out;
    uint32_t *n_cand_out;
    uint64_t *n_rows_out;             /* nodes with bases > 0: the LP's rows before the validity test */
    double *obj1_out, *obj2_out, *t_trio, *t_cov, *t_lp;
    /* shared cursor */
    volatile uint32_t next;
    pthread_mutex_t mu;
} par_ctx;

static void *species_worker(void *arg) {
    par_ctx *c = (par_ctx *)arg;
    for (;;) {
        pthread_mutex_lock(&c->mu);
        const uint32_t k = c->next < c->n_todo ? c->next++ : UINT32_MAX;
        pthread_mutex_unlock(&c->mu);
        if (k == UINT32_MAX) break;
        const uint32_t s = c->todo[k];
        const orc_graph *g = &c->graphs[s];
        double t0 = now_s();
        orc_trio_table T;
        memset(&T, 0, sizeof T);
        if (orc_trio_index(g, &T) != 0) { c->rc_out[s] = -100; continue; }
        double t1 = now_s();
        /* this species' reads, in file order (the per-species frame of profile.rs:439-463) */
        const uint64_t lo = c->first[s], n = c->first[s + 1] - lo;
        uint64_t steps = 0;
        for (uint64_t i = 0; i < n; ++i) { const uint64_t r = c->order[lo + i]; steps += c->step_off[r + 1] - c->step_off[r]; }
        uint64_t *so = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
        uint32_t *nid = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(steps ? steps : 1));
        int64_t *ps = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        int64_t *pe = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        int64_t *bases = (int64_t *)calloc(g->n_nodes ? g->n_nodes : 1, sizeof(int64_t));
        uint64_t *cov = (uint64_t *)calloc(g->n_nodes ? g->n_nodes : 1, sizeof(uint64_t));
        int64_t *tb = (int64_t *)calloc(T.n_unique ? T.n_unique : 1, sizeof(int64_t));
        orc_hap_metrics *met = (orc_hap_metrics *)calloc(g->n_paths ? g->n_paths : 1, sizeof(orc_hap_metrics));
        if (!so || !nid || !ps || !pe || !bases || !cov || !tb || !met) { c->rc_out[s] = -101; }
        else {
            uint64_t o = 0;
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t r = c->order[lo + i], b = c->step_off[r], e = c->step_off[r + 1];
                so[i] = o;
                memcpy(nid + o, c->node_id + b, sizeof(uint32_t) * (size_t)(e - b));
                o += e - b;
                ps[i] = c->pstart[r]; pe[i] = c->pend[r];
            }
            so[n] = o;
            uint64_t n_abort = 0;
            orc_node_coverage(g, &T, c->range_start[s], n, so, nid, ps, pe, bases, cov, tb, &n_abort);
            double t2 = now_s();
            uint64_t rows = 0;
            for (uint32_t v = 0; v < g->n_nodes; ++v) rows += bases[v] > 0;
            c->n_rows_out[s] = rows;
            int rc = 0;
            if (c->keep[s]) {
                rc = orc_optimize_species(g, &T, bases, cov, tb, &c->cfg, met, &c->n_cand_out[s], &c->obj1_out[s], &c->obj2_out[s]);
                orc_abundance_constraint(c->absolute[s], g->n_paths, met);
                if (c->metrics_out) memcpy(c->metrics_out + c->hap_off[s], met, sizeof(orc_hap_metrics) * g->n_paths);
            }
            double t3 = now_s();
            c->rc_out[s] = rc;
            c->t_trio[s] = t1 - t0; c->t_cov[s] = t2 - t1; c->t_lp[s] = t3 - t2;
        }
        free(so); free(nid); free(ps); free(pe); free(bases); free(cov); free(tb); free(met);
        orc_trio_free(&T);
    }
    return NULL;
}

int orc_par_profile_species(int n_threads, uint32_t n_species, const orc_graph *graphs, const int64_t *range_start, const uint64_t *step_off,
                            const uint32_t *node_id, const int64_t *pstart, const int64_t *pend, const uint64_t *first, const uint64_t *order,
                            const uint8_t *keep, const double *absolute, const orc_strain_config *cfg, uint32_t n_todo, const uint32_t *todo,
                            const uint64_t *hap_off, orc_hap_metrics *metrics_out, int32_t *rc_out, uint32_t *n_cand_out, uint64_t *n_rows_out,
                            double *obj1_out, double *obj2_out, double *t_trio, double *t_cov, double *t_lp) {
    if (n_threads < 1) n_threads = 1;
    par_ctx c;
    memset(&c, 0, sizeof c);
    c.n_species = n_species; c.graphs = graphs; c.range_start = range_start; c.step_off = step_off; c.node_id = node_id;
    c.pstart = pstart; c.pend = pend; c.first = first; c.order = order; c.keep = keep; c.absolute = absolute; c.cfg = *cfg;
    c.n_todo = n_todo; c.todo = todo; c.hap_off = hap_off; c.metrics_out = metrics_out; c.rc_out = rc_out; c.n_cand_out = n_cand_out;
    c.n_rows_out = n_rows_out; c.obj1_out = obj1_out; c.obj2_out = obj2_out; c.t_trio = t_trio; c.t_cov = t_cov; c.t_lp = t_lp;
    c.next = 0;
    pthread_mutex_init(&c.mu, NULL);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return -1;
    int started = 0;
    for (int t = 0; t < n_threads; ++t) if (pthread_create(&th[t], NULL, species_worker, &c) == 0) { if (t != started) th[started] = th[t]; ++started; }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(th);
    pthread_mutex_destroy(&c.mu);
    return started ? 0 : -2;
}
