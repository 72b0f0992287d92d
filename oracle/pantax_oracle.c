/*
 * pantax_oracle.c -- TEST INFRASTRUCTURE ONLY (see pantax_oracle.h).
 * Plain C restatement of the PanTax profiling hot path; every function cites the
 * reference lines it follows (paths relative to /root/reference/pantax/src).
 * "parity unpinned" by the reference's own tests; pinned by tests/golden/.
 */
#define _GNU_SOURCE /* qsort_r */
#include "pantax_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* a7: trio_nodes_info, profile.rs:658-740                             */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t a, b, c;
    uint64_t q; /* window start position in path_nodes */
} trio_rec;

static int cmp_trio(const void *x, const void *y) {
    const trio_rec *p = (const trio_rec *)x, *r = (const trio_rec *)y;
    if (p->a != r->a) return p->a < r->a ? -1 : 1;
    if (p->b != r->b) return p->b < r->b ? -1 : 1;
    if (p->c != r->c) return p->c < r->c ? -1 : 1;
    return p->q < r->q ? -1 : (p->q > r->q);
}

int orc_trio_index(const orc_graph *g, orc_trio_table *out) {
    memset(out, 0, sizeof(*out));
    uint64_t P = g->path_off[g->n_paths];
    uint64_t nwin = 0;
    for (uint32_t h = 0; h < g->n_paths; ++h) {
        uint64_t len = g->path_off[h + 1] - g->path_off[h];
        if (len >= 3) nwin += len - 2;
    }
    trio_rec *rec = (trio_rec *)malloc((nwin ? nwin : 1) * sizeof(trio_rec));
    uint8_t *uniq = (uint8_t *)calloc(P ? P : 1, 1);
    uint64_t n = 0;
    /* profile.rs:666-682: path.windows(3), swap ends when w[0] > w[2] */
    for (uint32_t h = 0; h < g->n_paths; ++h) {
        uint64_t b = g->path_off[h], e = g->path_off[h + 1];
        for (uint64_t q = b; q + 2 < e; ++q) {
            uint32_t w0 = g->path_nodes[q], w1 = g->path_nodes[q + 1], w2 = g->path_nodes[q + 2];
            if (w0 > w2) { uint32_t t = w0; w0 = w2; w2 = t; }
            rec[n].a = w0; rec[n].b = w1; rec[n].c = w2; rec[n].q = q;
            ++n;
        }
    }
    qsort(rec, n, sizeof(trio_rec), cmp_trio);
    /* profile.rs:688-709: count_per_trio counts every (hap, position) occurrence;
     * unique <=> count == 1 */
    uint64_t U = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && rec[j].a == rec[i].a && rec[j].b == rec[i].b && rec[j].c == rec[i].c) ++j;
        if (j - i == 1) { uniq[rec[i].q] = 1; ++U; }
        i = j;
    }
    out->n_unique = U;
    out->abc = (uint32_t *)malloc((U ? U : 1) * 3 * sizeof(uint32_t));
    out->hap = (uint32_t *)malloc((U ? U : 1) * sizeof(uint32_t));
    out->len = (int64_t *)malloc((U ? U : 1) * sizeof(int64_t));
    out->hap_off = (uint64_t *)malloc((g->n_paths + 1) * sizeof(uint64_t));
    out->sorted_abc = (uint32_t *)malloc((U ? U : 1) * 3 * sizeof(uint32_t));
    out->sorted_row = (uint64_t *)malloc((U ? U : 1) * sizeof(uint64_t));
    uint64_t u = 0;
    uint64_t *row_of_q = (uint64_t *)malloc((P ? P : 1) * sizeof(uint64_t));
    for (uint32_t h = 0; h < g->n_paths; ++h) {
        out->hap_off[h] = u;
        uint64_t b = g->path_off[h], e = g->path_off[h + 1];
        for (uint64_t q = b; q + 2 < e; ++q) {
            if (!uniq[q]) continue;
            uint32_t w0 = g->path_nodes[q], w1 = g->path_nodes[q + 1], w2 = g->path_nodes[q + 2];
            if (w0 > w2) { uint32_t t = w0; w0 = w2; w2 = t; }
            out->abc[3 * u] = w0; out->abc[3 * u + 1] = w1; out->abc[3 * u + 2] = w2;
            out->hap[u] = h;
            /* profile.rs:712 */
            out->len[u] = g->node_len[w0] + g->node_len[w1] + g->node_len[w2];
            row_of_q[q] = u;
            ++u;
        }
    }
    out->hap_off[g->n_paths] = u;
    uint64_t s = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (!uniq[rec[i].q]) continue;
        out->sorted_abc[3 * s] = rec[i].a; out->sorted_abc[3 * s + 1] = rec[i].b; out->sorted_abc[3 * s + 2] = rec[i].c;
        out->sorted_row[s] = row_of_q[rec[i].q];
        ++s;
    }
    free(row_of_q); free(rec); free(uniq);
    return 0;
}

void orc_trio_free(orc_trio_table *t) {
    free(t->abc); free(t->hap); free(t->len); free(t->hap_off); free(t->sorted_abc); free(t->sorted_row);
    memset(t, 0, sizeof(*t));
}

static int64_t trio_lookup(const orc_trio_table *t, uint32_t a, uint32_t b, uint32_t c) {
    uint64_t lo = 0, hi = t->n_unique;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        const uint32_t *k = t->sorted_abc + 3 * mid;
        int lt = (k[0] != a) ? (k[0] < a) : (k[1] != b) ? (k[1] < b) : (k[2] < c);
        if (lt) lo = mid + 1; else hi = mid;
    }
    if (lo < t->n_unique) {
        const uint32_t *k = t->sorted_abc + 3 * lo;
        if (k[0] == a && k[1] == b && k[2] == c) return (int64_t)t->sorted_row[lo];
    }
    return -1;
}

/* ------------------------------------------------------------------ */
/* a2: rcls.rs:237-258, 306-323                                        */
/* ------------------------------------------------------------------ */
int orc_bin_reads(uint64_t n_reads, const uint64_t *step_off, const uint32_t *node_id,
                  uint32_t n_ranges, const int64_t *range_start, const int64_t *range_end,
                  int32_t *species_idx_out) {
    for (uint64_t r = 0; r < n_reads; ++r) {
        int64_t mn = -1, mx = -1; /* rcls.rs:248 `[] => (-1,-1)` */
        for (uint64_t i = step_off[r]; i < step_off[r + 1]; ++i) {
            int64_t v = node_id[i];
            if (i == step_off[r]) { mn = mx = v; }
            else { if (v < mn) mn = v; if (v > mx) mx = v; }
        }
        int32_t sp = -1;
        for (uint32_t s = 0; s < n_ranges; ++s) /* rcls.rs:253-257 find() = first in file order */
            if (mn >= range_start[s] && mx <= range_end[s]) { sp = (int32_t)s; break; }
        species_idx_out[r] = sp;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* a3: profile.rs:208-297 counters                                     */
/* ------------------------------------------------------------------ */
int orc_species_counts(uint64_t n_reads, const int32_t *species_idx, const int64_t *read_len,
                       const int64_t *mapq, uint32_t n_ranges,
                       int64_t *read_count, int64_t *base_sum, int64_t *less_multi, int64_t *uniq_count) {
    for (uint32_t s = 0; s < n_ranges; ++s) read_count[s] = base_sum[s] = less_multi[s] = uniq_count[s] = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        int32_t s = species_idx[r];
        if (s < 0) continue; /* profile.rs:3352-3356 filter species != "U" */
        read_count[s] += 1;
        base_sum[s] += read_len[r];
        if (mapq[r] >= 3 && mapq[r] <= 60) { /* profile.rs:224-232 */
            less_multi[s] += 1;
            if (mapq[r] == 60) uniq_count[s] += 1;
        }
    }
    return 0;
}

int orc_species_profile(uint64_t n_reads, const int32_t *species_idx, const int64_t *read_len,
                        uint32_t n_ranges, const int64_t *read_count, const int64_t *base_sum,
                        const int64_t *less_multi, const int64_t *uniq_count, const double *avg_len,
                        int filtered, uint8_t *keep_out, double *absolute_out, double *abundance_out) {
    /* profile.rs:312-319: unique read_len among the first 1000 rows of the non-U frame */
    int64_t first_len = -1; int equal = 1; uint64_t seen = 0;
    for (uint64_t r = 0; r < n_reads && seen < 1000; ++r) {
        if (species_idx[r] < 0) continue;
        if (seen == 0) first_len = read_len[r];
        else if (read_len[r] != first_len) equal = 0;
        ++seen;
    }
    if (seen == 0) equal = 0;
    double total = 0.0;
    for (uint32_t s = 0; s < n_ranges; ++s) {
        keep_out[s] = 0; absolute_out[s] = 0.0; abundance_out[s] = 0.0;
        if (read_count[s] == 0) continue;
        if (filtered) {
            /* profile.rs:239-245: inner join drops species with no mapq in [3,60] */
            if (less_multi[s] == 0) continue;
            if (!(uniq_count[s] > 0 && (double)less_multi[s] > (double)read_count[s] / 10.0)) continue;
        }
        if (!(avg_len[s] > 0)) continue; /* left join would give null */
        /* profile.rs:214/246 (count*len) vs :259/266 (sum len) */
        int64_t base_count = equal ? read_count[s] * first_len : base_sum[s];
        keep_out[s] = 1;
        absolute_out[s] = (double)base_count / avg_len[s]; /* profile.rs:336 */
    }
    /* profile.rs:341: sum in frame order; we use range-file order (group_by order is
     * unspecified in the reference) */
    for (uint32_t s = 0; s < n_ranges; ++s) if (keep_out[s]) total += absolute_out[s];
    for (uint32_t s = 0; s < n_ranges; ++s) if (keep_out[s]) abundance_out[s] = absolute_out[s] / total;
    return 0;
}

/* ------------------------------------------------------------------ */
/* a8: get_node_abundances, profile.rs:743-1026 (integer part)         */
/* ------------------------------------------------------------------ */
int orc_node_coverage(const orc_graph *g, const orc_trio_table *trio, int64_t range_start,
                      uint64_t n_reads, const uint64_t *step_off, const uint32_t *node_id,
                      const int64_t *pstart, const int64_t *pend,
                      int64_t *bases_per_node, uint64_t *node_base_cov, int64_t *trio_bases,
                      uint64_t *n_abort) {
    uint32_t V = g->n_nodes;
    uint64_t *bit_off = (uint64_t *)malloc((V + 1) * sizeof(uint64_t));
    bit_off[0] = 0;
    for (uint32_t v = 0; v < V; ++v) bit_off[v + 1] = bit_off[v] + (uint64_t)g->node_len[v];
    uint8_t *covered = (uint8_t *)calloc(bit_off[V] ? bit_off[V] : 1, 1); /* profile.rs:776-781 */
    memset(bases_per_node, 0, V * sizeof(int64_t));
    if (trio) memset(trio_bases, 0, trio->n_unique * sizeof(int64_t));
    uint64_t aborts = 0;
    uint64_t cap = 16;
    int64_t *rlen = (int64_t *)malloc(cap * sizeof(int64_t)); /* read_nodes_len by position of first occurrence */
    uint32_t *loc = (uint32_t *)malloc(cap * sizeof(uint32_t));

    for (uint64_t r = 0; r < n_reads; ++r) {
        uint64_t b = step_off[r], e = step_off[r + 1], k = e - b;
        if (k == 0) continue; /* profile.rs:794-796 */
        if (k > cap) { cap = k * 2; rlen = (int64_t *)realloc(rlen, cap * sizeof(int64_t)); loc = (uint32_t *)realloc(loc, cap * sizeof(uint32_t)); }
        int bad = 0;
        for (uint64_t i = 0; i < k; ++i) {
            /* profile.rs:790: id - 1 - start with start = range_start - 1 (profile.rs:2886) */
            int64_t l = (int64_t)node_id[b + i] - range_start;
            if (l < 0 || l >= (int64_t)V) { bad = 1; break; } /* would panic at profile.rs:849 */
            loc[i] = (uint32_t)l;
        }
        if (bad) { ++aborts; continue; }
        int64_t rs = pstart[r], re = pend[r];
        int64_t target_len = re - rs; /* profile.rs:800 */
        if (k == 1) { /* profile.rs:811 (start_node == end_node && len == 1) */
            uint32_t node = loc[0];
            if (target_len < 0) continue; /* profile.rs:821-827 */
            bases_per_node[node] += target_len; /* profile.rs:828-829 */
            rlen[0] = target_len;
            if (rs < re && re <= g->node_len[node]) /* profile.rs:832 */
                for (int64_t j = rs; j < re; ++j) covered[bit_off[node] + (uint64_t)j] = 1;
            continue; /* < 3 nodes: no trios */
        }
        if (rs > g->node_len[loc[0]]) { ++aborts; continue; } /* assert profile.rs:854 */
        int64_t seen = 0;
        for (uint64_t i = 0; i < k; ++i) {
            uint32_t node = loc[i];
            int64_t nl = g->node_len[node];
            int64_t aln, sidx;
            if (i == 0) { aln = nl - rs; sidx = rs; }             /* profile.rs:853-856 */
            else if (i == k - 1) {                                  /* profile.rs:857-859 */
                if (target_len < seen) target_len = seen;
                aln = target_len - seen; sidx = 0;
            } else { aln = nl; sidx = 0; }                          /* profile.rs:860-862 */
            int64_t hi = sidx + aln; if (hi > nl) hi = nl;         /* profile.rs:871 */
            for (int64_t j = sidx; j < hi; ++j) covered[bit_off[node] + (uint64_t)j] = 1;
            seen += aln;                                            /* profile.rs:878 */
            /* profile.rs:879-882: first occurrence in this read only */
            int first = 1; uint64_t fpos = i;
            for (uint64_t j = 0; j < i; ++j) if (loc[j] == node) { first = 0; fpos = j; break; }
            if (first) { rlen[i] = aln; bases_per_node[node] += aln; }
            else rlen[i] = rlen[fpos]; /* read_nodes_len.get(node) = first occurrence's length */
        }
        if (k < 3 || !trio || trio->n_unique == 0) continue; /* profile.rs:886-888 */
        for (uint64_t i = 0; i + 2 < k; ++i) { /* profile.rs:890-907 */
            uint32_t a = loc[i], bb = loc[i + 1], c = loc[i + 2];
            int64_t len_sum = rlen[i] + rlen[i + 1] + rlen[i + 2];
            /* lookup (a,b,c) or (c,b,a): table keys are canonical, so canonicalise */
            uint32_t ka = a, kc = c;
            if (ka > kc) { uint32_t t = ka; ka = kc; kc = t; }
            int64_t row = trio_lookup(trio, ka, bb, kc);
            if (row >= 0) trio_bases[row] += len_sum;
        }
    }
    /* profile.rs:844/874 + 1018-1023: node_base_cov = number of covered bases */
    for (uint32_t v = 0; v < V; ++v) {
        uint64_t c = 0;
        for (uint64_t j = bit_off[v]; j < bit_off[v + 1]; ++j) c += covered[j];
        node_base_cov[v] = c;
    }
    if (n_abort) *n_abort = aborts;
    free(rlen); free(loc); free(covered); free(bit_off);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a9: zscore_filter profile.rs:1028-1051 and the per-hap part of       */
/* first_filter_paths profile.rs:1114-1147                              */
/* ------------------------------------------------------------------ */
static double zscore_mean(const double *x, uint64_t n) {
    if (n == 0) return 0.0;
    double sum = 0.0;
    for (uint64_t i = 0; i < n; ++i) sum += x[i];
    double mean = sum / (double)n;
    double ss = 0.0;
    for (uint64_t i = 0; i < n; ++i) ss += (x[i] - mean) * (x[i] - mean);
    double sd = sqrt(ss / (double)n);
    if (sd == 0.0) return 0.0; /* profile.rs:1043-1045 -> empty -> mean 0.0 (:1143-1147) */
    double fs = 0.0; uint64_t fc = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (fabs((x[i] - mean) / sd) < 3.0) { fs += x[i]; ++fc; }
    return fc ? fs / (double)fc : 0.0;
}

int orc_hap_trio_stats(const orc_trio_table *t, uint32_t n_paths, const int64_t *trio_bases,
                       uint64_t *n_trio, uint64_t *n_nonzero, double *mean_filtered) {
    for (uint32_t h = 0; h < n_paths; ++h) {
        uint64_t b = t->hap_off[h], e = t->hap_off[h + 1];
        n_trio[h] = e - b;
        double *nz = (double *)malloc((e - b ? e - b : 1) * sizeof(double));
        uint64_t c = 0;
        for (uint64_t u = b; u < e; ++u) {
            double ab = (double)trio_bases[u] / (double)t->len[u]; /* profile.rs:1013-1014 */
            if (ab > 0.0) nz[c++] = ab;
        }
        n_nonzero[h] = c;
        mean_filtered[h] = zscore_mean(nz, c);
        free(nz);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* a10: coeff matrix + path_cov_ratio, profile.rs:1333-1361            */
/* ------------------------------------------------------------------ */
int orc_path_masks(const orc_graph *g, uint32_t n_cand, const uint32_t *cand,
                   const uint64_t *node_base_cov, uint64_t *mask_out, float *ratio_out) {
    /* row v of the 0/1 matrix = ORC_NW(n_cand) words at mask_out + v * nw (one word up to 64 columns) */
    const uint32_t nw = ORC_NW(n_cand);
    memset(mask_out, 0, (size_t)g->n_nodes * nw * sizeof(uint64_t));
    for (uint32_t k = 0; k < n_cand; ++k) {
        uint32_t h = cand[k];
        for (uint64_t q = g->path_off[h]; q < g->path_off[h + 1]; ++q)
            mask_out[(size_t)g->path_nodes[q] * nw + (k >> 6)] |= (1ull << (k & 63)); /* coeff_matrix[(v,pos)] = 1.0 */
    }
    for (uint32_t k = 0; k < n_cand; ++k) {
        float cov = 0.0f, len = 0.0f; /* f32 accumulation, profile.rs:1344-1357 */
        for (uint32_t v = 0; v < g->n_nodes; ++v)
            if (mask_out[(size_t)v * nw + (k >> 6)] >> (k & 63) & 1ull) { cov += (float)node_base_cov[v]; len += (float)g->node_len[v]; }
        ratio_out[k] = cov / len;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* a12: the PAO LP as an exact LAD active-set descent                   */
/* ------------------------------------------------------------------ */
typedef struct { const uint64_t *m; double a; } lrow; /* m: the node's mask words */
static int cmp_mask(const uint64_t *p, const uint64_t *q, uint32_t nw) {
    for (uint32_t w = nw; w-- > 0;) if (p[w] != q[w]) return p[w] < q[w] ? -1 : 1;
    return 0;
}
static int cmp_lrow(const void *x, const void *y, void *nwp) {
    const lrow *p = (const lrow *)x, *q = (const lrow *)y;
    int c = cmp_mask(p->m, q->m, *(const uint32_t *)nwp);
    if (c) return c;
    return p->a < q->a ? -1 : (p->a > q->a);
}
static int mask_any(const uint64_t *m, uint32_t nw) { for (uint32_t w = 0; w < nw; ++w) if (m[w]) return 1; return 0; }
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
typedef struct { double t, w; uint32_t k; uint64_t idx; } brk;
static int cmp_brk(const void *x, const void *y) {
    const brk *p = (const brk *)x, *q = (const brk *)y;
    if (p->t != q->t) return p->t < q->t ? -1 : 1;
    if (p->k != q->k) return p->k < q->k ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx);
}
enum { C_LB = 0, C_UB = 1, C_PAT = 2, C_FIXED = 3 };
typedef struct { int type; uint32_t j; uint32_t k; uint64_t i0, i1; } lcon;

static int invert(const double *N, double *W, int p) {
    /* Gauss-Jordan with partial pivoting; N, W are p x p row-major */
    double *M = (double *)malloc((size_t)p * 2 * p * sizeof(double));
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j) { M[i * 2 * p + j] = N[i * p + j]; M[i * 2 * p + p + j] = (i == j); }
    for (int c = 0; c < p; ++c) {
        int piv = c; double best = fabs(M[c * 2 * p + c]);
        for (int r = c + 1; r < p; ++r) if (fabs(M[r * 2 * p + c]) > best) { best = fabs(M[r * 2 * p + c]); piv = r; }
        if (best < 1e-12) { free(M); return -1; }
        if (piv != c) for (int j = 0; j < 2 * p; ++j) { double t = M[c * 2 * p + j]; M[c * 2 * p + j] = M[piv * 2 * p + j]; M[piv * 2 * p + j] = t; }
        double d = M[c * 2 * p + c];
        for (int j = 0; j < 2 * p; ++j) M[c * 2 * p + j] /= d;
        for (int r = 0; r < p; ++r) {
            if (r == c) continue;
            double f = M[r * 2 * p + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * p; ++j) M[r * 2 * p + j] -= f * M[c * 2 * p + j];
        }
    }
    for (int i = 0; i < p; ++i) for (int j = 0; j < p; ++j) W[i * p + j] = M[i * 2 * p + p + j];
    free(M);
    return 0;
}

static uint64_t lower_bound_a(const lrow *r, uint64_t lo, uint64_t hi, double v) {
    while (lo < hi) { uint64_t m = (lo + hi) / 2; if (r[m].a < v) lo = m + 1; else hi = m; }
    return lo;
}
static uint64_t upper_bound_a(const lrow *r, uint64_t lo, uint64_t hi, double v) {
    while (lo < hi) { uint64_t m = (lo + hi) / 2; if (r[m].a <= v) lo = m + 1; else hi = m; }
    return lo;
}
static double mdot(const uint64_t *mask, const double *x, int p) {
    double s = 0.0;
    for (int j = 0; j < p; ++j) if (mask[j >> 6] >> (j & 63) & 1ull) s += x[j];
    return s;
}

double orc_lad_objective(uint64_t n_nodes, const uint64_t *mask, const double *abund, uint32_t n_cand,
                         const double *x) {
    double s = 0.0; uint64_t n = 0;
    const uint32_t nw = ORC_NW(n_cand);
    for (uint64_t v = 0; v < n_nodes; ++v) {
        if (!(abund[v] > 0.0)) continue; /* profile.rs:1380-1385 */
        s += fabs(mdot(mask + v * nw, x, (int)n_cand) - abund[v]);
        ++n;
    }
    return n ? s / (double)n : 0.0; /* profile.rs:1450 */
}

int orc_lad_solve(uint64_t n_nodes, const uint64_t *mask, const double *abund, uint32_t n_cand,
                  const double *ub, double *x_out, double *obj_out, int32_t *iters_out, int32_t *status_out) {
    int p = (int)n_cand;
    uint32_t nw = ORC_NW(n_cand);
    uint64_t n = 0; double amax = 0.0;
    for (uint64_t v = 0; v < n_nodes; ++v) if (abund[v] > 0.0) { if (mask_any(mask + v * nw, nw)) ++n; if (abund[v] > amax) amax = abund[v]; }
    lrow *rows = (lrow *)malloc((n ? n : 1) * sizeof(lrow));
    uint64_t m = 0;
    for (uint64_t v = 0; v < n_nodes; ++v) if (abund[v] > 0.0 && mask_any(mask + v * nw, nw)) { rows[m].m = mask + v * nw; rows[m].a = abund[v]; ++m; }
    qsort_r(rows, n, sizeof(lrow), cmp_lrow, &nw);
    /* patterns = runs of equal mask */
    uint32_t K = 0;
    for (uint64_t i = 0; i < n; ++i) if (i == 0 || cmp_mask(rows[i].m, rows[i - 1].m, nw)) ++K;
    uint64_t *pst = (uint64_t *)malloc((K + 1) * sizeof(uint64_t));
    double *peps = (double *)malloc((K ? K : 1) * sizeof(double));
    int *pact = (int *)malloc((K ? K : 1) * sizeof(int));
    K = 0;
    for (uint64_t i = 0; i < n; ++i) if (i == 0 || cmp_mask(rows[i].m, rows[i - 1].m, nw)) pst[K++] = i;
    pst[K] = n;
    /* symbolic-perturbation substitute: shift every pattern's breakpoints by a distinct
     * tiny eps so that no two patterns tie at a vertex; removed again in the final x. */
    double delta = 1e-10 * (amax > 1.0 ? amax : 1.0);
    for (uint32_t k = 0; k < K; ++k) {
        uint64_t hk = rows[pst[k]].m[0];
        for (uint32_t w = 1; w < nw; ++w) hk = splitmix64(hk) ^ rows[pst[k]].m[w];
        peps[k] = delta * (0.25 + 0.5 * (double)(splitmix64(hk) >> 11) * (1.0 / 9007199254740992.0));
    }

    lcon *act = (lcon *)malloc((size_t)(p ? p : 1) * sizeof(lcon));
    double *N = (double *)calloc((size_t)(p ? p * p : 1), sizeof(double));
    double *W = (double *)calloc((size_t)(p ? p * p : 1), sizeof(double));
    double *x = (double *)calloc((size_t)(p ? p : 1), sizeof(double));
    double *c = (double *)calloc((size_t)(p ? p : 1), sizeof(double));
    double *gvec = (double *)calloc((size_t)(p ? p : 1), sizeof(double));
    double *lam = (double *)calloc((size_t)(p ? p : 1), sizeof(double));
    double *d = (double *)calloc((size_t)(p ? p : 1), sizeof(double));
    uint64_t *plo = (uint64_t *)malloc((K ? K : 1) * sizeof(uint64_t));
    uint64_t *pup = (uint64_t *)malloc((K ? K : 1) * sizeof(uint64_t));
    double *ps = (double *)malloc((K ? K : 1) * sizeof(double));
    brk *bl = (brk *)malloc((n ? n : 1) * sizeof(brk));
    for (int i = 0; i < p; ++i) {
        act[i].type = (ub[i] > 0.0) ? C_LB : C_FIXED; act[i].j = (uint32_t)i;
        N[i * p + i] = 1.0; W[i * p + i] = 1.0;
    }
    const double tol = 1e-7;
    int status = 0, it = 0;
    const int max_it = 200 * p + 2000;
    for (; it < max_it; ++it) {
        /* vertex of the perturbed problem */
        for (int i = 0; i < p; ++i) {
            switch (act[i].type) {
            case C_LB: case C_FIXED: c[i] = 0.0; break;
            case C_UB: c[i] = ub[act[i].j]; break;
            default: c[i] = rows[act[i].i0].a + peps[act[i].k];
            }
        }
        for (int j = 0; j < p; ++j) { double s = 0.0; for (int i = 0; i < p; ++i) s += W[j * p + i] * c[i]; x[j] = s; }
        for (uint32_t k = 0; k < K; ++k) pact[k] = -1;
        for (int i = 0; i < p; ++i) if (act[i].type == C_PAT) pact[act[i].k] = i;
        for (int j = 0; j < p; ++j) gvec[j] = 0.0;
        for (uint32_t k = 0; k < K; ++k) {
            uint64_t st = pst[k], en = pst[k + 1];
            const uint64_t *mk = rows[st].m;
            double sigma;
            if (pact[k] >= 0) {
                plo[k] = act[pact[k]].i0; pup[k] = act[pact[k]].i1;
            } else {
                ps[k] = mdot(mk, x, p);
                double sv = ps[k] - peps[k];
                plo[k] = lower_bound_a(rows, st, en, sv);
                pup[k] = upper_bound_a(rows, plo[k], en, sv);
            }
            sigma = (double)(plo[k] - st) - (double)(en - pup[k]);
            for (int j = 0; j < p; ++j) if (mk[j >> 6] >> (j & 63) & 1ull) gvec[j] += sigma;
        }
        /* multipliers: lam_i = -g . w_i, w_i = column i of W */
        int best = -1, bdir = 0; double bscore = -tol, bderiv = 0.0;
        for (int i = 0; i < p; ++i) {
            double s = 0.0, nrm = 0.0;
            for (int j = 0; j < p; ++j) { s += gvec[j] * W[j * p + i]; nrm += W[j * p + i] * W[j * p + i]; }
            lam[i] = -s; nrm = sqrt(nrm);
            double deriv = 0.0; int dir = 0;
            if (act[i].type == C_PAT) {
                double w = (double)(act[i].i1 - act[i].i0);
                if (lam[i] > w + tol) { dir = +1; deriv = w - lam[i]; }
                else if (lam[i] < -w - tol) { dir = -1; deriv = w + lam[i]; }
            } else if (act[i].type == C_LB) { if (lam[i] > tol) { dir = +1; deriv = -lam[i]; } }
            else if (act[i].type == C_UB) { if (lam[i] < -tol) { dir = -1; deriv = lam[i]; } }
            if (dir && deriv / nrm < bscore) { bscore = deriv / nrm; best = i; bdir = dir; bderiv = deriv; }
        }
        if (best < 0) break; /* optimal */
        for (int j = 0; j < p; ++j) d[j] = bdir * W[j * p + best];
        /* bound ratio test */
        double tmax = INFINITY; int bj = -1, btype = C_LB;
        for (int j = 0; j < p; ++j) {
            if (ub[j] <= 0.0) continue;
            if (d[j] < -1e-12) { double t = x[j] / (-d[j]); if (t < 0) t = 0; if (t < tmax) { tmax = t; bj = j; btype = C_LB; } }
            else if (d[j] > 1e-12) { double t = (ub[j] - x[j]) / d[j]; if (t < 0) t = 0; if (t < tmax) { tmax = t; bj = j; btype = C_UB; } }
        }
        /* breakpoints ahead */
        uint64_t nb = 0;
        for (uint32_t k = 0; k < K; ++k) {
            uint64_t st = pst[k], en = pst[k + 1];
            const uint64_t *mk = rows[st].m;
            double rho; double s0;
            if (pact[k] >= 0) {
                if (pact[k] != best) continue; /* stays tight: n_i . d = 0 */
                rho = (double)bdir; s0 = rows[act[best].i0].a + peps[k];
            } else {
                rho = mdot(mk, d, p); s0 = ps[k];
                if (fabs(rho) < 1e-12) continue;
                /* tie group the perturbation failed to split: enters at t = 0 with half weight */
                for (uint64_t i = plo[k]; i < pup[k]; ++i) { bl[nb].t = 0.0; bl[nb].w = fabs(rho); bl[nb].k = k; bl[nb].idx = i; ++nb; }
            }
            if (rho > 0) {
                for (uint64_t i = pup[k]; i < en; ++i) {
                    double t = (rows[i].a + peps[k] - s0) / rho; if (t < 0) t = 0;
                    if (t > tmax) break;
                    bl[nb].t = t; bl[nb].w = 2.0 * rho; bl[nb].k = k; bl[nb].idx = i; ++nb;
                }
            } else {
                for (uint64_t i = plo[k]; i-- > st;) {
                    double t = (rows[i].a + peps[k] - s0) / rho; if (t < 0) t = 0;
                    if (t > tmax) break;
                    bl[nb].t = t; bl[nb].w = -2.0 * rho; bl[nb].k = k; bl[nb].idx = i; ++nb;
                }
            }
        }
        qsort(bl, nb, sizeof(brk), cmp_brk);
        double slope = bderiv; int found = 0; lcon ent; memset(&ent, 0, sizeof(ent));
        for (uint64_t i = 0; i < nb; ++i) {
            slope += bl[i].w;
            if (slope >= -tol) {
                uint32_t k = bl[i].k; double av = rows[bl[i].idx].a;
                ent.type = C_PAT; ent.k = k;
                ent.i0 = lower_bound_a(rows, pst[k], pst[k + 1], av);
                ent.i1 = upper_bound_a(rows, ent.i0, pst[k + 1], av);
                found = 1; break;
            }
        }
        if (!found) {
            if (bj < 0) { status = 2; break; } /* unbounded ray: cannot happen for LAD */
            ent.type = btype; ent.j = (uint32_t)bj;
        }
        act[best] = ent;
        for (int j = 0; j < p; ++j) N[best * p + j] = 0.0;
        if (ent.type == C_PAT) { const uint64_t *mk = rows[pst[ent.k]].m; for (int j = 0; j < p; ++j) if (mk[j >> 6] >> (j & 63) & 1ull) N[best * p + j] = 1.0; }
        else N[best * p + ent.j] = 1.0;
        if (invert(N, W, p) != 0) { status = 3; break; }
    }
    if (it >= max_it) status = 1;
    /* final vertex with the UNPERTURBED right-hand sides */
    for (int i = 0; i < p; ++i) {
        switch (act[i].type) {
        case C_LB: case C_FIXED: c[i] = 0.0; break;
        case C_UB: c[i] = ub[act[i].j]; break;
        default: c[i] = rows[act[i].i0].a;
        }
    }
    for (int j = 0; j < p; ++j) {
        double s = 0.0; for (int i = 0; i < p; ++i) s += W[j * p + i] * c[i];
        if (s < 0.0) s = 0.0;
        if (s > ub[j]) s = ub[j];
        x_out[j] = s;
    }
    if (obj_out) *obj_out = orc_lad_objective(n_nodes, mask, abund, n_cand, x_out);
    if (iters_out) *iters_out = it;
    if (status_out) *status_out = status;
    free(rows); free(pst); free(peps); free(pact); free(act); free(N); free(W); free(x); free(c);
    free(gvec); free(lam); free(d); free(plo); free(pup); free(ps); free(bl);
    return 0;
}

/* ------------------------------------------------------------------ */
/* optimize_otu: profile.rs:2884-3026 with first_filter_paths           */
/* (:1080-1227) and second_filter_paths (:1229-1285)                    */
/* ------------------------------------------------------------------ */
static double round2(double x) { return round(x * 100.0) / 100.0; } /* f64::round = half away from zero */

/* ------------------------------------------------------------------ */
/* a11: sample_sorted, profile.rs:1287-1295 (rand 0.9.2, unpinned)     */
/* ------------------------------------------------------------------ */
#define ORC_ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define ORC_QR(a, b, c, d) \
    a += b; d ^= a; d = ORC_ROTL32(d, 16); c += d; b ^= c; b = ORC_ROTL32(b, 12); \
    a += b; d ^= a; d = ORC_ROTL32(d, 8);  c += d; b ^= c; b = ORC_ROTL32(b, 7)

void orc_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]) {
    uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u; /* "expand 32-byte k" */
    uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
    uint32_t x12 = (uint32_t)counter, x13 = (uint32_t)(counter >> 32), x14 = 0, x15 = 0;
    for (int i = 0; i < rounds / 2; ++i) {
        ORC_QR(x0, x4, x8, x12); ORC_QR(x1, x5, x9, x13); ORC_QR(x2, x6, x10, x14); ORC_QR(x3, x7, x11, x15);   /* columns */
        ORC_QR(x0, x5, x10, x15); ORC_QR(x1, x6, x11, x12); ORC_QR(x2, x7, x8, x13); ORC_QR(x3, x4, x9, x14);   /* diagonals */
    }
    out[0] = x0 + 0x61707865u; out[1] = x1 + 0x3320646eu; out[2] = x2 + 0x79622d32u; out[3] = x3 + 0x6b206574u;
    out[4] = x4 + key[0]; out[5] = x5 + key[1]; out[6] = x6 + key[2]; out[7] = x7 + key[3];
    out[8] = x8 + key[4]; out[9] = x9 + key[5]; out[10] = x10 + key[6]; out[11] = x11 + key[7];
    out[12] = x12 + (uint32_t)counter; out[13] = x13 + (uint32_t)(counter >> 32); out[14] = x14; out[15] = x15;
}

/* StdRng (rand 0.9) = ChaCha12Rng behind a 4-block (64-word) buffer; words are handed out in keystream order */
typedef struct { uint32_t key[8]; uint64_t block; uint32_t words[64]; int pos; } orc_stdrng;

static void orc_stdrng_seed_from_u64(orc_stdrng *r, uint64_t state) {
    /* rand_core SeedableRng::seed_from_u64: PCG32 output per 4 seed bytes (little endian) */
    for (int i = 0; i < 8; ++i) {
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
        r->key[i] = rot ? ((xs >> rot) | (xs << (32 - rot))) : xs;
    }
    r->block = 0; r->pos = 64;
}
static uint32_t orc_stdrng_u32(orc_stdrng *r) {
    if (r->pos == 64) {
        for (int b = 0; b < 4; ++b) orc_chacha_block(r->key, r->block++, 12, r->words + 16 * b);
        r->pos = 0;
    }
    return r->words[r->pos++];
}
/* UniformInt<u32>::sample_single_inclusive (Canon's method, one correction draw) */
static uint32_t orc_range_inclusive(orc_stdrng *r, uint32_t low, uint32_t high) {
    uint32_t range = high - low + 1u;
    if (range == 0) return orc_stdrng_u32(r);
    uint64_t wide = (uint64_t)orc_stdrng_u32(r) * (uint64_t)range;
    uint32_t result = (uint32_t)(wide >> 32), lo_order = (uint32_t)wide;
    if (lo_order > (uint32_t)(~range + 1u)) {
        uint32_t new_hi = (uint32_t)(((uint64_t)orc_stdrng_u32(r) * (uint64_t)range) >> 32);
        if ((uint32_t)(lo_order + new_hi) < lo_order) result += 1u; /* checked_add overflowed */
    }
    return low + result;
}
static int orc_cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }

int orc_sample_sorted_positions(uint32_t length, uint32_t amount, uint64_t seed, uint32_t *out) {
    if (amount > length) return -1; /* rand panics */
    orc_stdrng rng;
    orc_stdrng_seed_from_u64(&rng, seed);
    int big = length >= 500000u;
    int use_inplace, use_floyd = 0;
    if (amount < 163) { /* rand::seq::index::sample: f32 cost model */
        static const float C[2][2] = {{1.6f, 8.0f / 45.0f}, {10.0f, 70.0f / 9.0f}};
        float amount_fp = (float)amount, m4 = C[0][big] * amount_fp;
        use_inplace = amount > 11 && (float)length < (C[1][big] + m4) * amount_fp;
        use_floyd = !use_inplace;
    } else {
        static const float C[2] = {270.0f, 330.0f / 9.0f};
        use_inplace = (float)length < C[big] * (float)amount;
    }
    if (use_inplace) { /* sample_inplace: partial Fisher-Yates over 0..length */
        uint32_t *idx = (uint32_t *)malloc((size_t)(length ? length : 1) * sizeof(uint32_t));
        for (uint32_t i = 0; i < length; ++i) idx[i] = i;
        for (uint32_t i = 0; i < amount; ++i) {
            uint32_t j = orc_range_inclusive(&rng, i, length - 1u);
            uint32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t;
        }
        memcpy(out, idx, (size_t)amount * sizeof(uint32_t));
        free(idx);
    } else if (use_floyd) { /* sample_floyd */
        uint32_t n = 0;
        for (uint32_t j = length - amount; j < length; ++j) {
            uint32_t t = orc_range_inclusive(&rng, 0, j);
            for (uint32_t q = 0; q < n; ++q) if (out[q] == t) { out[q] = j; break; }
            out[n++] = t;
        }
    } else { /* sample_rejection: Uniform::new(0, length).sample until unseen */
        uint8_t *seen = (uint8_t *)calloc((size_t)length / 8 + 1, 1);
        uint32_t thresh = (uint32_t)(~length + 1u) % length;
        for (uint32_t n = 0; n < amount;) {
            uint64_t wide = (uint64_t)orc_stdrng_u32(&rng) * (uint64_t)length;
            if ((uint32_t)wide < thresh) continue;
            uint32_t pos = (uint32_t)(wide >> 32);
            if (seen[pos >> 3] & (1u << (pos & 7))) continue;
            seen[pos >> 3] |= (uint8_t)(1u << (pos & 7));
            out[n++] = pos;
        }
        free(seen);
    }
    qsort(out, amount, sizeof(uint32_t), orc_cmp_u32); /* sampled.sort_unstable() */
    return 0;
}


int orc_optimize_species(const orc_graph *g, const orc_trio_table *trio, const int64_t *bases_per_node,
                         const uint64_t *node_base_cov, const int64_t *trio_bases,
                         const orc_strain_config *cfg, orc_hap_metrics *met,
                         uint32_t *n_candidates_out, double *obj1_out, double *obj2_out) {
    uint32_t V = g->n_nodes, H = g->n_paths;
    double *ab = (double *)malloc((V ? V : 1) * sizeof(double));
    double amax = -INFINITY; /* profile.rs:1316-1319 fold(NEG_INFINITY, max) */
    for (uint32_t v = 0; v < V; ++v) { ab[v] = (double)bases_per_node[v] / (double)g->node_len[v]; if (ab[v] > amax) amax = ab[v]; }
    memset(met, 0, H * sizeof(orc_hap_metrics));
    uint32_t *cand = (uint32_t *)malloc((H ? H : 1) * sizeof(uint32_t));
    uint32_t nc = 0;
    int same_path = 0, second_opt = 0, rc = 0;
    uint64_t *mask = NULL; float *ratio = NULL; double *ub = NULL, *x1 = NULL, *x2 = NULL; uint8_t *keep = NULL;
    uint64_t U = trio->n_unique;
    if (obj1_out) *obj1_out = NAN;
    if (obj2_out) *obj2_out = NAN;

    if (H != 1 && U != 0) { /* profile.rs:1098 (hap2trio_nodes_m.len() = U*H) */
        uint64_t *nt = (uint64_t *)malloc(H * sizeof(uint64_t)), *nz = (uint64_t *)malloc(H * sizeof(uint64_t));
        double *mf = (double *)malloc(H * sizeof(double));
        orc_hap_trio_stats(trio, H, trio_bases, nt, nz, mf);
        for (uint32_t h = 0; h < H; ++h) {
            if (nt[h] == 0) continue; /* profile.rs:1119 */
            double frac = (double)nz[h] / (double)nt[h];
            met[h].unique_trio_nodes_fraction = round2(frac); met[h].has |= ORC_HAS_FRACTION; /* :1136-1138 */
            double fm = mf[h];
            if (cfg->shift) { /* profile.rs:1140-1165 */
                double sh;
                if (fm >= 1.0) { sh = cfg->unique_trio_nodes_fraction + (0.8 - cfg->unique_trio_nodes_fraction) * fm / 100.0; if (sh > 0.8) sh = 0.8; }
                else sh = cfg->unique_trio_nodes_fraction * fm;
                if (frac < sh) continue;
            } else {
                if (frac < cfg->unique_trio_nodes_fraction) continue; /* profile.rs:1168 */
            }
            met[h].frequencies_mean = fm; met[h].has |= ORC_HAS_FREQ_MEAN;
            cand[nc++] = h;
        }
        free(nt); free(nz); free(mf);
    } else {
        int all_same = 1;
        if (H != 1) { /* profile.rs:1187-1209 */
            uint64_t l0 = g->path_off[1] - g->path_off[0];
            for (uint32_t h = 1; h < H && all_same; ++h) {
                uint64_t lh = g->path_off[h + 1] - g->path_off[h];
                if (lh != l0 || memcmp(g->path_nodes + g->path_off[h], g->path_nodes + g->path_off[0], l0 * sizeof(uint32_t)) != 0) all_same = 0;
            }
        }
        if (H == 1 || all_same) { /* profile.rs:1191-1205, 1211-1224 */
            same_path = (H != 1);
            double s = 0.0; uint64_t c = 0;
            for (uint32_t v = 0; v < V; ++v) { double x = ab[v] > (double)cfg->min_depth ? ab[v] : 0.0; if (x > 0.0) { s += x; ++c; } } /* :2941-2944 */
            met[0].frequencies_mean = round2(c ? s / (double)c : 0.0); met[0].has |= ORC_HAS_FREQ_MEAN;
            cand[nc++] = 0;
        } else {
            for (uint32_t h = 0; h < H; ++h) cand[nc++] = h; /* profile.rs:1208 */
        }
    }
    *n_candidates_out = nc;
    if (nc > 0) {
        /* no cap on the columns (dense nvert x npaths matrix in the reference, profile.rs:1333-1342) */
        mask = (uint64_t *)malloc((size_t)(V ? V : 1) * ORC_NW(nc) * sizeof(uint64_t));
        ratio = (float *)malloc(nc * sizeof(float));
        ub = (double *)malloc(3 * (size_t)nc * sizeof(double)); x1 = ub + nc; x2 = x1 + nc;
        orc_path_masks(g, nc, cand, node_base_cov, mask, ratio);
        for (uint32_t k = 0; k < nc; ++k) { met[cand[k]].path_cov_ratio = (double)ratio[k]; met[cand[k]].has |= ORC_HAS_RATIO; ub[k] = 1.05 * amax; }
        /* a11: the LP sees only the sampled valid rows (profile.rs:2738-2752); amax and the ratios above do not */
        if (cfg->sample_nodes > 0) {
            uint32_t nv = 0;
            for (uint32_t v = 0; v < V; ++v) nv += ab[v] > 0.0;
            if (nv > (uint32_t)cfg->sample_nodes) {
                uint32_t *valid = (uint32_t *)malloc((size_t)nv * sizeof(uint32_t)), *pos = (uint32_t *)malloc((size_t)cfg->sample_nodes * sizeof(uint32_t));
                nv = 0;
                for (uint32_t v = 0; v < V; ++v) if (ab[v] > 0.0) valid[nv++] = v;
                orc_sample_sorted_positions(nv, (uint32_t)cfg->sample_nodes, 42, pos);
                uint32_t q = 0;
                for (uint32_t r = 0; r < nv; ++r) {
                    if (q < (uint32_t)cfg->sample_nodes && pos[q] == r) { ++q; continue; }
                    ab[valid[r]] = 0.0; /* not an LP row; ab is not read below except by the LP and its objective */
                }
                free(valid); free(pos);
            }
        }
        int32_t it, st;
        orc_lad_solve(V, mask, ab, nc, ub, x1, obj1_out, &it, &st);
        if (st != 0) { rc = -1; goto done; }
        for (uint32_t k = 0; k < nc; ++k) { met[cand[k]].first_sol = x1[k]; met[cand[k]].has |= ORC_HAS_FIRST; }
        /* second_filter_paths, profile.rs:1229-1285 */
        keep = (uint8_t *)calloc(nc, 1);
        if (H != 1 && U > 0) {
            second_opt = 1;
            for (uint32_t k = 0; k < nc; ++k) {
                orc_hap_metrics *mm = &met[cand[k]];
                double fm = (mm->has & ORC_HAS_FREQ_MEAN) ? mm->frequencies_mean : 0.0;
                if (fm == 0.0) continue;
                double sol = mm->first_sol;
                double fr = round2(fabs(sol - fm) / (sol + fm));
                mm->divergence = fr; mm->has |= ORC_HAS_DIVERGENCE;
                if (fr > cfg->unique_trio_nodes_mean_count_f) {
                    if (fr <= 0.6) {
                        double sc = mm->unique_trio_nodes_fraction * mm->path_cov_ratio;
                        if (sc < cfg->single_cov_ratio || sol == 0.0) continue;
                        mm->is_rescue = 1; mm->has |= ORC_HAS_RESCUE; keep[k] = 1;
                    }
                } else if (sol != 0.0) keep[k] = 1;
            }
        } else if ((H != 1 && U == 0 && same_path) || H == 1) {
            double fm = met[0].frequencies_mean;
            if (fm > 0.0) {
                double sol = met[0].first_sol;
                met[0].divergence = round2(fabs(sol - fm) / (sol + fm)); met[0].has |= ORC_HAS_DIVERGENCE;
                met[0].second_sol = sol; met[0].has |= ORC_HAS_SECOND;
            }
        } else {
            for (uint32_t k = 0; k < nc; ++k) { met[cand[k]].second_sol = met[cand[k]].first_sol; met[cand[k]].has |= ORC_HAS_SECOND; }
        }
        if (second_opt) { /* profile.rs:1482-1508 (Gurobi semantics); :2849-2879 (highs_opt) */
            for (uint32_t k = 0; k < nc; ++k) if (!keep[k]) ub[k] = 0.0;
            orc_lad_solve(V, mask, ab, nc, ub, x2, obj2_out, &it, &st);
            if (st != 0) { rc = -1; goto done; }
            /* highs_opt: `sols2 = &all_sols[..min(len, second_possible_paths_idx.len())]`, then `possible_paths_idx.iter().zip(sols2)` (:2865, :2871):
             * only the first K candidate positions are visited, K = number of survivors */
            uint32_t k_lim = nc;
            if (cfg->solver_semantics == 1) { k_lim = 0; for (uint32_t k = 0; k < nc; ++k) k_lim += keep[k] ? 1u : 0u; }
            for (uint32_t k = 0; k < nc && k < k_lim; ++k) if (keep[k]) { met[cand[k]].second_sol = x2[k]; met[cand[k]].has |= ORC_HAS_SECOND; }
        }
    }
done:
    free(ab); free(cand); free(mask); free(ratio); free(ub); free(keep);
    return rc;
}

/* profile.rs:3028-3070 */
int orc_abundance_constraint(double species_cov, uint32_t H, orc_hap_metrics *met) {
    double sum = 0.0, mx = -INFINITY;
    for (uint32_t h = 0; h < H; ++h) {
        if ((met[h].has & ORC_HAS_RESCUE) && met[h].is_rescue && (met[h].has & ORC_HAS_FIRST) && (met[h].has & ORC_HAS_SECOND))
            if (met[h].first_sol < met[h].second_sol) met[h].second_sol = met[h].first_sol;
        double v = (met[h].has & ORC_HAS_SECOND) ? met[h].second_sol : 0.0;
        sum += v; if (v > mx) mx = v;
    }
    double diff = fabs(sum - species_cov) / ((sum + species_cov) / 2.0);
    for (uint32_t h = 0; h < H; ++h) { met[h].total_cov_diff = diff; met[h].has |= ORC_HAS_TOTAL_DIFF; }
    if (H && mx > 1.05 * species_cov) {
        double f = species_cov / sum;
        for (uint32_t h = 0; h < H; ++h)
            if (!((met[h].has & ORC_HAS_RESCUE) && met[h].is_rescue) && (met[h].has & ORC_HAS_SECOND)) met[h].second_sol *= f;
    }
    return 0;
}


/* ------------------------------------------------------------------ */
/* filter_max_alignment_mt, gaf_filter.rs:44-97           */
/* ------------------------------------------------------------------ */
typedef struct { const char *id; uint32_t id_len; uint64_t line; int32_t matches, mapq, span; double identity; } orc_aln;

static int orc_is_space(char c) { return c == ' ' || (c >= 9 && c <= 13); }
/* str::parse::<i32>() */
static int orc_parse_i32(const char *b, const char *e, int32_t *out) {
    if (b == e) return 0;
    int neg = 0;
    if (*b == '+' || *b == '-') { neg = *b == '-'; ++b; if (b == e) return 0; }
    long long v = 0;
    for (; b < e; ++b) { if (*b < '0' || *b > '9') return 0; v = v * 10 + (*b - '0'); if (v > 2147483648LL) return 0; }
    if (neg) v = -v;
    if (v > 2147483647LL) return 0;
    *out = (int32_t)v;
    return 1;
}
static int orc_word_ci(const char *b, const char *e, const char *w) {
    size_t n = strlen(w);
    if ((size_t)(e - b) != n) return 0;
    for (size_t i = 0; i < n; ++i) if ((b[i] | 0x20) != w[i]) return 0;
    return 1;
}
/* f64::from_str: [+-] (inf | infinity | nan | digits [. digits] [e [+-] digits]); value by the C library's correctly
 * rounded strtod once the spelling is known to be one Rust accepts */
static int orc_parse_f64(const char *b, const char *e, double *out) {
    const char *p = b;
    if (p < e && (*p == '+' || *p == '-')) ++p;
    if (p == e) return 0;
    if (!(orc_word_ci(p, e, "inf") || orc_word_ci(p, e, "infinity") || orc_word_ci(p, e, "nan"))) {
        int nd = 0;
        while (p < e && *p >= '0' && *p <= '9') { ++p; ++nd; }
        if (p < e && *p == '.') { ++p; while (p < e && *p >= '0' && *p <= '9') { ++p; ++nd; } }
        if (!nd) return 0;
        if (p < e && (*p == 'e' || *p == 'E')) {
            ++p;
            if (p < e && (*p == '+' || *p == '-')) ++p;
            if (p == e) return 0;
            while (p < e) { if (*p < '0' || *p > '9') return 0; ++p; }
        }
        if (p != e) return 0;
    }
    char tmp[512];
    size_t n = (size_t)(e - b);
    char *z = n < sizeof(tmp) ? tmp : (char *)malloc(n + 1);
    memcpy(z, b, n); z[n] = 0;
    *out = strtod(z, NULL);
    if (z != tmp) free(z);
    return 1;
}
static int orc_aln_cmp(const void *a, const void *b) {
    const orc_aln *x = (const orc_aln *)a, *y = (const orc_aln *)b;
    uint32_t n = x->id_len < y->id_len ? x->id_len : y->id_len;
    int c = memcmp(x->id, y->id, n);
    if (c) return c;
    if (x->id_len != y->id_len) return x->id_len < y->id_len ? -1 : 1;
    return x->line < y->line ? -1 : x->line > y->line;
}

int64_t orc_gaf_filter(const char *text, uint64_t size, uint8_t *keep_out, uint64_t *n_records_out) {
    uint64_t n_lines = 0, cap = 1024, n = 0;
    orc_aln *rec = (orc_aln *)malloc(cap * sizeof(orc_aln));
    for (uint64_t pos = 0; pos < size; ++n_lines) {
        const char *nl = (const char *)memchr(text + pos, '\n', size - pos);
        uint64_t end = nl ? (uint64_t)(nl - text) : size;
        const char *b = text + pos, *e = text + end;
        pos = end + 1;
        keep_out[n_lines] = 0;
        while (b < e && orc_is_space(*b)) ++b;             /* trim (also eats the '\r' lines() would drop) */
        while (e > b && orc_is_space(e[-1])) --e;
        const char *fb[16], *fe[16];
        int nf = 0;
        for (const char *q = b;;) {
            const char *t = q;
            while (t < e && *t != '\t') ++t;
            fb[nf] = q; fe[nf] = t; ++nf;
            if (t >= e || nf == 16) break;
            q = t + 1;
        }
        if (nf < 16) continue;
        orc_aln a;
        int32_t s3, s2;
        const char *ib = fb[15];
        for (const char *c = fb[15]; c < fe[15]; ++c) if (*c == ':') ib = c + 1;
        if (!orc_parse_i32(fb[9], fe[9], &a.matches) || !orc_parse_f64(ib, fe[15], &a.identity) || !orc_parse_i32(fb[11], fe[11], &a.mapq) ||
            !orc_parse_i32(fb[3], fe[3], &s3) || !orc_parse_i32(fb[2], fe[2], &s2)) continue;
        a.span = (int32_t)((uint32_t)s3 - (uint32_t)s2);
        a.id = fb[0]; a.id_len = (uint32_t)(fe[0] - fb[0]); a.line = n_lines;
        if (n == cap) { cap *= 2; rec = (orc_aln *)realloc(rec, cap * sizeof(orc_aln)); }
        rec[n++] = a;
    }
    if (n_records_out) *n_records_out = n;
    qsort(rec, n, sizeof(orc_aln), orc_aln_cmp);
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && rec[j].id_len == rec[i].id_len && memcmp(rec[j].id, rec[i].id, rec[i].id_len) == 0) ++j;
        int32_t bm = rec[i].matches; double bi = rec[i].identity; int have = !(bi != bi);
        for (uint64_t q = i; q < j; ++q) {                  /* best = max (matches, identity); a NaN identity never wins */
            const orc_aln *r = &rec[q];
            if (r->identity != r->identity) { if (r->matches > bm) { bm = r->matches; bi = r->identity; have = 0; } continue; }
            if (r->matches > bm || (r->matches == bm && (!have || r->identity > bi))) { bm = r->matches; bi = r->identity; have = 1; }
        }
        if (have)
            for (uint64_t q = i; q < j; ++q)
                if (rec[q].mapq > 20 && rec[q].span > 1000 && rec[q].matches == bm && rec[q].identity == bi) { keep_out[rec[q].line] = 1; break; }
        i = j;
    }
    free(rec);
    return (int64_t)n_lines;
}
