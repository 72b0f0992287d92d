/* synth_set.c -- the synthetic PanTax inputs of SURVEY.md section 8d, generated natively (test / bench data generator, NOT
 * part of the product library; built by __graft_entry__.build() into tools/native/libsynthset.so).
 *
 * Same statistical model as pantax_amd/synth.py:make_species / make_reads (core segments, SNP bubbles of two 1-bp alleles,
 * accessory segments; strain membership by clades of a random binary tree; min(1024, 1 + Geometric(mean 32)) segment
 * lengths; 20 % of the strains present at LogNormal(ln 8, 1) depth; error-free reads of a fixed length or HiFi-shaped,
 * 85 % MAPQ 60, 0.1 % adversarial records), but every quantity is a pure function of (seed, species) or of (seed, chunk,
 * read): a counter-based generator (splitmix64 of key + counter) replaces numpy's sequential streams.  So
 *   - any rank can generate any species' graph and any chunk of the reads without generating the rest, and the set is the
 *     same for every number of ranks and threads (bench.py --scaling strong, the bounded sample of the CPU baseline);
 *   - the 1k-species / 100M-read configuration takes seconds instead of minutes.
 * Reads are drawn independently (strain by inverse CDF over depth x genome length), so a chunk is already interleaved over
 * the species like a real GAF. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t seed;
    uint32_t H;            /* strains per species (<= 63) */
    int64_t genome_len;    /* target length of strain 0's walk */
    double frac_snp, frac_acc, mean_len, present_frac, depth_mu, depth_sigma;
} synth_params;

static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define GOLDEN 0x9E3779B97F4A7C15ull
static inline uint64_t stream_key(uint64_t seed, uint64_t kind, uint64_t a, uint64_t b) {
    return mix64(mix64(mix64(mix64(seed + GOLDEN) ^ (kind * 0xD1342543DE82EF95ull)) + a * GOLDEN) + b * 0xA24BAED4963EE407ull);
}
static inline uint64_t rnd(uint64_t key, uint64_t i) { return mix64(key + (i + 1) * GOLDEN); }
static inline double u01(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

/* ------------------------------------------------------------------------------------------------ species graphs */
typedef struct { uint64_t key, ctr; } seq_rng;
static inline uint64_t seq_next(seq_rng *r) { return rnd(r->key, r->ctr++); }
static inline uint32_t seq_below(seq_rng *r, uint32_t n) { return (uint32_t)(u01(seq_next(r)) * n); }

/* random binary tree over H strains -> clade bitmasks (both children of every split), at most 2H - 2 */
static int make_clades(seq_rng *r, uint32_t H, uint64_t *clades) {
    uint32_t members[64], stack_b[128], stack_e[128];
    int sp = 0, n = 0;
    for (uint32_t h = 0; h < H; ++h) members[h] = h;
    stack_b[sp] = 0; stack_e[sp] = H; ++sp;
    while (sp) {
        --sp;
        const uint32_t b = stack_b[sp], e = stack_e[sp], len = e - b;
        if (len <= 1) continue;
        for (uint32_t i = len - 1; i > 0; --i) {            /* shuffle the segment */
            const uint32_t j = seq_below(r, i + 1);
            const uint32_t t = members[b + i]; members[b + i] = members[b + j]; members[b + j] = t;
        }
        const uint32_t k = 1 + seq_below(r, len - 1);
        uint64_t ml = 0, mr = 0;
        for (uint32_t i = b; i < b + k; ++i) ml |= 1ull << members[i];
        for (uint32_t i = b + k; i < e; ++i) mr |= 1ull << members[i];
        clades[n++] = ml; clades[n++] = mr;
        stack_b[sp] = b; stack_e[sp] = b + k; ++sp;
        stack_b[sp] = b + k; stack_e[sp] = e; ++sp;
    }
    return n;
}

/* the node sequence of species s: calls emit(v, len, member) for node v = 0, 1, ...; returns V */
typedef void (*emit_fn)(void *user, uint64_t v, int64_t len, uint64_t member);
static uint64_t species_nodes(const synth_params *p, uint32_t s, emit_fn emit, void *user) {
    const uint32_t H = p->H;
    const uint64_t full = (H >= 64) ? ~0ull : ((1ull << H) - 1);
    uint64_t clades[128];
    seq_rng tr = { stream_key(p->seed, 1, s, 0), 0 };
    int nc = H > 1 ? make_clades(&tr, H, clades) : 0;
    const uint64_t key = stream_key(p->seed, 1, s, 1);
    const double lg = log(1.0 - 1.0 / p->mean_len);
    uint64_t v = 0;
    int64_t cum0 = 0;
    for (uint64_t site = 0;; ++site) {
        const double uk = u01(rnd(key, 3 * site));
        int kind = 0;                                        /* 0 core 1 snp 2 accessory */
        if (site > 0 && nc > 0) { if (uk < p->frac_snp) kind = 1; else if (uk < p->frac_snp + p->frac_acc) kind = 2; }
        double ul = u01(rnd(key, 3 * site + 1));
        if (ul < 1e-300) ul = 1e-300;
        int64_t seg = 1 + (int64_t)floor(log(ul) / lg) + 1;   /* 1 + Geometric(1/mean_len), support of the geometric starts at 1 */
        if (seg > 1024) seg = 1024;
        const uint64_t cl = nc ? clades[(uint32_t)(u01(rnd(key, 3 * site + 2)) * nc)] : full;
        if (kind == 0) { emit(user, v++, seg, full); cum0 += seg; }
        else if (kind == 1) {
            emit(user, v++, 1, cl); emit(user, v++, 1, full ^ cl);
            cum0 += 1;                                        /* strain 0 is on exactly one allele */
        } else { emit(user, v++, seg, cl); if (cl & 1) cum0 += seg; }
        if (cum0 >= p->genome_len && v >= 8) break;
    }
    return v;
}

typedef struct { uint32_t H; uint64_t *plen; int64_t *glen; } dims_acc;
static void dims_emit(void *user, uint64_t v, int64_t len, uint64_t member) {
    dims_acc *a = (dims_acc *)user;
    (void)v;
    while (member) { const int h = __builtin_ctzll(member); member &= member - 1; a->plen[h] += 1; a->glen[h] += len; }
}

/* V, P, per-strain walk length (nodes) and genome length (bases), truth depth -- without storing the graph */
int synth_species_dims(const synth_params *p, uint32_t s, uint64_t *V, uint64_t *P, uint64_t *path_len, int64_t *genome_len, double *depth) {
    if (p->H < 1 || p->H > 63) return -1;
    dims_acc a = { p->H, path_len, genome_len };
    for (uint32_t h = 0; h < p->H; ++h) { path_len[h] = 0; genome_len[h] = 0; depth[h] = 0.0; }
    *V = species_nodes(p, s, dims_emit, &a);
    uint64_t tot = 0;
    for (uint32_t h = 0; h < p->H; ++h) tot += path_len[h];
    *P = tot;
    /* present strains: n_present of H without replacement, LogNormal depth */
    seq_rng r = { stream_key(p->seed, 1, s, 2), 0 };
    uint32_t idx[64];
    for (uint32_t h = 0; h < p->H; ++h) idx[h] = h;
    uint32_t np_ = (uint32_t)floor(p->present_frac * p->H + 0.5);
    if (np_ < 1) np_ = 1;
    if (np_ > p->H) np_ = p->H;
    for (uint32_t i = 0; i < np_; ++i) {
        const uint32_t j = i + seq_below(&r, p->H - i);
        const uint32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t;
        double u1 = u01(seq_next(&r)), u2 = u01(seq_next(&r));
        if (u1 < 1e-300) u1 = 1e-300;
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        depth[idx[i]] = exp(p->depth_mu + p->depth_sigma * z);
    }
    return 0;
}

typedef struct { int64_t *node_len; uint64_t *member; } fill_acc;
static void fill_emit(void *user, uint64_t v, int64_t len, uint64_t member) {
    fill_acc *a = (fill_acc *)user;
    a->node_len[v] = len; a->member[v] = member;
}

/* node_len [V], path_off [H+1], path_nodes [P] (local 0-based ids, walks in strain order); V, P from synth_species_dims */
int synth_species_fill(const synth_params *p, uint32_t s, uint64_t V, int64_t *node_len, uint64_t *path_off, uint32_t *path_nodes) {
    uint64_t *member = (uint64_t *)malloc(sizeof(uint64_t) * (V ? V : 1));
    if (!member) return -2;
    fill_acc a = { node_len, member };
    const uint64_t v2 = species_nodes(p, s, fill_emit, &a);
    if (v2 != V) { free(member); return -3; }
    uint64_t o = 0;
    for (uint32_t h = 0; h < p->H; ++h) {
        path_off[h] = o;
        const uint64_t bit = 1ull << h;
        for (uint64_t v = 0; v < V; ++v) if (member[v] & bit) path_nodes[o++] = (uint32_t)v;
    }
    path_off[p->H] = o;
    free(member);
    return 0;
}

/* cum_end[i] = bases of the walk up to and including its i-th node (u32: genomes < 4 Gbp) */
void synth_walk_cum(const int64_t *node_len, const uint32_t *path_nodes, uint64_t n, uint32_t *cum_end) {
    uint64_t c = 0;
    for (uint64_t i = 0; i < n; ++i) { c += (uint64_t)node_len[path_nodes[i]]; cum_end[i] = (uint32_t)c; }
}

/* ------------------------------------------------------------------------------------------------ reads */
typedef struct {
    uint64_t seed;
    uint32_t n_strains;              /* present strains */
    const double *cum_w;             /* [n_strains] cumulative read share, last = 1 */
    const uint64_t *walk_nodes;      /* [n_strains] address of the strain's walk (uint32 local ids) */
    const uint64_t *walk_cum;        /* [n_strains] address of its cum_end (uint32) */
    const uint64_t *walk_len;        /* [n_strains] nodes on the walk */
    const int64_t *genome_len;       /* [n_strains] */
    const int64_t *range_start;      /* [n_strains] first global node id of the strain's species */
    uint32_t n_species;
    const int64_t *species_start;    /* [n_species] (adversarial cross-species walks) */
    int32_t long_reads;
    int64_t read_len;
    double adversarial_frac;
} synth_read_ctx;

static inline uint64_t ub_u32(const uint32_t *a, uint64_t n, uint64_t x) {   /* first i with a[i] > x */
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if ((uint64_t)a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
#define DRAWS 12

typedef struct { uint32_t st; int rev; uint64_t i0, i1; int64_t rl, pos, mapq; int adv; double uadv; } read_draw;
static inline void draw_read(const synth_read_ctx *c, uint64_t key, uint64_t j, read_draw *d) {
    const double us = u01(rnd(key, DRAWS * j));
    uint32_t lo = 0, hi = c->n_strains - 1;                  /* first strain with cum_w > us */
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (c->cum_w[mid] <= us) lo = mid + 1; else hi = mid; }
    d->st = lo;
    const int64_t glen = c->genome_len[lo];
    int64_t rl = c->read_len;
    if (c->long_reads) {
        double u1 = u01(rnd(key, DRAWS * j + 5)), u2 = u01(rnd(key, DRAWS * j + 6));
        if (u1 < 1e-300) u1 = 1e-300;
        double x = 15000.0 + 3000.0 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        if (x < 2000.0) x = 2000.0;
        if (x > 25000.0) x = 25000.0;
        rl = (int64_t)x;
    }
    if (rl > glen) rl = glen;
    d->rl = rl;
    d->pos = (int64_t)(u01(rnd(key, DRAWS * j + 1)) * (double)(glen - rl + 1));
    d->rev = u01(rnd(key, DRAWS * j + 2)) < 0.5;
    d->mapq = u01(rnd(key, DRAWS * j + 3)) < 0.85 ? 60 : (int64_t)(u01(rnd(key, DRAWS * j + 4)) * 60.0);
    d->adv = u01(rnd(key, DRAWS * j + 7)) < c->adversarial_frac ? 1 + (int)(u01(rnd(key, DRAWS * j + 8)) * 3.0) : 0;
    d->uadv = u01(rnd(key, DRAWS * j + 9));
    const uint32_t *cum = (const uint32_t *)(uintptr_t)c->walk_cum[lo];
    const uint64_t n = c->walk_len[lo];
    d->i0 = ub_u32(cum, n, (uint64_t)d->pos);
    d->i1 = ub_u32(cum, n, (uint64_t)(d->pos + rl - 1));
    if (d->i1 >= n) d->i1 = n - 1;
}

/* pass 1: walk lengths of reads [j0, j1) of chunk `chunk` -> n_steps[0 .. j1-j0) */
void synth_reads_count(const synth_read_ctx *c, uint64_t chunk, uint64_t j0, uint64_t j1, uint32_t *n_steps) {
    const uint64_t key = stream_key(c->seed, 2, chunk, 0);
    read_draw d;
    for (uint64_t j = j0; j < j1; ++j) { draw_read(c, key, j, &d); n_steps[j - j0] = (uint32_t)(d.i1 - d.i0 + 1); }
}

/* pass 2: the records.  step_off[0 .. n] are the ABSOLUTE offsets of these reads in node_id / strand; the per-read columns
 * are written at [0 .. n). */
void synth_reads_fill(const synth_read_ctx *c, uint64_t chunk, uint64_t j0, uint64_t j1, const uint64_t *step_off, uint32_t *node_id,
                      uint8_t *strand, int64_t *pstart, int64_t *pend, int64_t *qlen, int64_t *mapq) {
    const uint64_t key = stream_key(c->seed, 2, chunk, 0);
    read_draw d;
    for (uint64_t j = j0; j < j1; ++j) {
        draw_read(c, key, j, &d);
        const uint64_t r = j - j0, b = step_off[r], k = d.i1 - d.i0 + 1;
        const uint32_t *walk = (const uint32_t *)(uintptr_t)c->walk_nodes[d.st];
        const uint32_t *cum = (const uint32_t *)(uintptr_t)c->walk_cum[d.st];
        const uint32_t base = (uint32_t)c->range_start[d.st];
        int64_t ps;
        if (!d.rev) {
            for (uint64_t i = 0; i < k; ++i) { node_id[b + i] = walk[d.i0 + i] + base; strand[b + i] = 0; }
            ps = d.pos - (int64_t)(d.i0 ? cum[d.i0 - 1] : 0);
        } else {
            for (uint64_t i = 0; i < k; ++i) { node_id[b + i] = walk[d.i1 - i] + base; strand[b + i] = 1; }
            ps = (int64_t)cum[d.i1] - (d.pos + d.rl);          /* unused tail of the walk's last node */
        }
        int64_t pe = ps + d.rl;
        if (d.adv == 1 && k == 1) { ps += 50; pe = ps - (1 + (int64_t)(d.uadv * 49.0)); }      /* end < start on a single node (vg issue 4249) */
        else if (d.adv == 2 && k >= 3) node_id[b + 2] = node_id[b];                              /* a, b, a repeat */
        else if (d.adv == 3 && k >= 2 && c->n_species > 1)                                       /* walk leaves its species -> "U" */
            node_id[b + k - 1] = (uint32_t)c->species_start[(uint32_t)(d.uadv * c->n_species)];
        pstart[r] = ps; pend[r] = pe; qlen[r] = d.rl; mapq[r] = d.mapq;
    }
}

/* ------------------------------------------------------------------------------------------------ GAF text */
static char *put_u(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
static char *put_i(char *p, int64_t v) {
    if (v < 0) { *p++ = '-'; return put_u(p, (uint64_t)(-(v + 1)) + 1u); }
    return put_u(p, (uint64_t)v);
}
static inline int ndig(uint64_t v) { int n = 1; while (v >= 10) { v /= 10; ++n; } return n; }
static inline int ndig_i(int64_t v) { return v < 0 ? 1 + ndig((uint64_t)(-(v + 1)) + 1u) : ndig((uint64_t)v); }

/* bytes of the GAF lines of reads [r0, r1) (the format of synth_gaf.c / synth.py:write_gaf; plen == qlen) */
uint64_t synth_gaf_size(uint64_t r0, uint64_t r1, uint64_t id_base, const uint64_t *step_off, const uint32_t *node_id, const int64_t *pstart,
                        const int64_t *pend, const int64_t *qlen, const int64_t *mapq, uint64_t tags_len) {
    uint64_t n = 0;
    for (uint64_t r = r0; r < r1; ++r) {
        const uint64_t b = step_off[r], e = step_off[r + 1];
        n += 3 + ndig(id_base + r) + 3 + 5 * ndig_i(qlen[r]) + 3 + 3 + ndig_i(pstart[r]) + ndig_i(pend[r]) + ndig_i(mapq[r]) + 7 + tags_len + 1;
        if (e == b) n += 1;
        for (uint64_t i = b; i < e; ++i) n += 1 + ndig(node_id[i]);
    }
    return n;
}

/* the same lines written at byte `offset` of an existing file (pwrite-style through fseeko); returns bytes written or < 0 */
int64_t synth_gaf_write_at(const char *path, uint64_t offset, uint64_t r0, uint64_t r1, uint64_t id_base, const uint64_t *step_off,
                           const uint32_t *node_id, const uint8_t *strand, const int64_t *pstart, const int64_t *pend, const int64_t *qlen,
                           const int64_t *mapq, const char *tags) {
    FILE *f = fopen(path, "r+b");
    if (!f) return -1;
    if (fseeko(f, (off_t)offset, SEEK_SET) != 0) { fclose(f); return -6; }
    const size_t cap = 1u << 23;
    char *buf = (char *)malloc(cap);
    if (!buf) { fclose(f); return -2; }
    const size_t tl = strlen(tags);
    char *p = buf;
    int64_t total = 0;
    for (uint64_t r = r0; r < r1; ++r) {
        const uint64_t b = step_off[r], e = step_off[r + 1];
        const size_t need = 256 + tl + 12 * (size_t)(e - b);
        if (need > cap) { free(buf); fclose(f); return -4; }
        if ((size_t)(p - buf) + need > cap) {
            if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); fclose(f); return -3; }
            total += p - buf;
            p = buf;
        }
        memcpy(p, "S0R", 3); p += 3; p = put_u(p, id_base + r); memcpy(p, "/1\t", 3); p += 3;
        p = put_i(p, qlen[r]); memcpy(p, "\t0\t", 3); p += 3; p = put_i(p, qlen[r]); memcpy(p, "\t+\t", 3); p += 3;
        if (e == b) *p++ = '*';
        for (uint64_t i = b; i < e; ++i) { *p++ = strand[i] ? '<' : '>'; p = put_u(p, node_id[i]); }
        *p++ = '\t'; p = put_i(p, qlen[r]); *p++ = '\t'; p = put_i(p, pstart[r]); *p++ = '\t'; p = put_i(p, pend[r]);
        *p++ = '\t'; p = put_i(p, qlen[r]); *p++ = '\t'; p = put_i(p, qlen[r]); *p++ = '\t'; p = put_i(p, mapq[r]);
        *p++ = '\t'; memcpy(p, tags, tl); p += tl; *p++ = '\n';
    }
    if (p > buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); fclose(f); return -3; }
    total += p - buf;
    free(buf);
    if (fclose(f) != 0) return -5;
    return total;
}
