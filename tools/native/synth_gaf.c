/* synth_gaf.c -- fast writer of the synthetic GAF text (test / bench data generator, NOT part of the product library).
 * Byte-for-byte the same lines as pantax_amd/synth.py:write_gaf for reads without explicit ids:
 *   S0R<r>/1 \t qlen \t 0 \t qlen \t + \t <walk> \t plen \t pstart \t pend \t qlen \t qlen \t mapq \t <tags> \n
 * Built by __graft_entry__.build() into tools/native/libsynthgaf.so (gcc -O2 -shared). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

/* r0..r1: read range; id_base: number printed in the read id of read r is id_base + r.  Returns bytes written, <0 on error. */
int64_t synth_write_gaf(const char *path, int append, uint64_t r0, uint64_t r1, uint64_t id_base, const uint64_t *step_off, const uint32_t *node_id,
                        const uint8_t *strand, const int64_t *pstart, const int64_t *pend, const int64_t *qlen,
                        const int64_t *mapq, const int64_t *plen, const char *tags) {
    FILE *f = fopen(path, append ? "ab" : "wb");
    if (!f) return -1;
    const size_t cap = 1u << 23;
    char *buf = (char *)malloc(cap);
    if (!buf) { fclose(f); return -2; }
    const size_t tl = strlen(tags);
    char *p = buf;
    int64_t total = 0;
    for (uint64_t r = r0; r < r1; ++r) {
        const uint64_t b = step_off[r], e = step_off[r + 1];
        const size_t need = 256 + tl + 12 * (size_t)(e - b);
        if (need > cap) { free(buf); fclose(f); return -4; }   /* walk longer than the buffer */
        if ((size_t)(p - buf) + need > cap) {
            if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); fclose(f); return -3; }
            total += p - buf;
            p = buf;
        }
        memcpy(p, "S0R", 3); p += 3; p = put_u(p, id_base + r); memcpy(p, "/1\t", 3); p += 3;
        p = put_i(p, qlen[r]); memcpy(p, "\t0\t", 3); p += 3; p = put_i(p, qlen[r]); memcpy(p, "\t+\t", 3); p += 3;
        if (e == b) *p++ = '*';
        for (uint64_t i = b; i < e; ++i) { *p++ = strand[i] ? '<' : '>'; p = put_u(p, node_id[i]); }
        *p++ = '\t'; p = put_i(p, plen[r]); *p++ = '\t'; p = put_i(p, pstart[r]); *p++ = '\t'; p = put_i(p, pend[r]);
        *p++ = '\t'; p = put_i(p, qlen[r]); *p++ = '\t'; p = put_i(p, qlen[r]); *p++ = '\t'; p = put_i(p, mapq[r]);
        *p++ = '\t'; memcpy(p, tags, tl); p += tl; *p++ = '\n';
    }
    if (p > buf && fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); fclose(f); return -3; }
    total += p - buf;
    free(buf);
    if (fclose(f) != 0) return -5;
    return total;
}
