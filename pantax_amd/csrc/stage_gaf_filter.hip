// stage_gaf_filter.hip -- SURVEY 8f-3: the long-read best-alignment filter that sits between the aligner and the
// profile (filter_max_alignment_mt, gaf_filter.rs:44-97; called from alignment.rs:171-175).  Per GAF line
// (parse_line, gaf_filter.rs:21-42):
//   * the line is trimmed (ASCII white space here) and split on tabs; fewer than 16 fields => not a record
//   * read_id = field 0; matches = field 9 (i32); mapq = field 11 (i32); span = field 3 - field 2 (i32 each);
//     identity = the text after the last ':' of field 15, parsed as f64; any parse failure => not a record
// Then (gaf_filter.rs:60-93): best[read_id] = max over its records of (matches, identity), lexicographic; a record
// is written when mapq > 20, span > 1000 and (matches, identity) == best, and only ONE line per read id.
// The reference runs both loops under rayon: which of several equal-best lines of a read is written, and the order of
// the output lines, are scheduling accidents there.  Here: the first such line in file order, output in file order.
// A NaN identity never compares greater or equal, so such a record is never written (the reference agrees unless it
// happens to be the first record rayon inserts for its read).
//
// On the device: newline scan (shared with the tokenizer) -> one thread per line parses the five fields and hashes the
// read id (64-bit FNV-1a + avalanche, as the tokenizer) -> stable radix sort of (hash, line) -> one thread per run of
// equal hashes finds the best key and the first passing line that holds it.  f64 text goes through the exact
// fast path (<= 19 digits giving an integer < 2^53 and |exponent| <= 22: one correctly rounded multiply or divide);
// the rare other spellings are marked and converted by the host's strtod before the grouping.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "common.hpp"
#include "gaf_scan.hpp"
#include "host_io.hpp"
#include "primitives.hpp"

namespace ptx {

namespace {

struct FilterRec {   // structure of arrays, one entry per raw line
    uint64_t *hash, *ident;       // identity as ordered bits (see order_bits); 0 = NaN
    uint32_t *matches;            // i32 biased to unsigned order
    uint32_t *f15_b, *f15_e;      // identity text span, for the host fallback
    uint8_t *state;               // 0 = not a record, 1 = record, 2 = record whose identity needs the host
    uint8_t *pass;                // mapq > 20 && span > 1000
};

__device__ __forceinline__ bool is_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// str::parse::<i32>: optional sign, at least one digit, digits only, no overflow
__device__ __forceinline__ bool dev_parse_i32(const uint8_t *__restrict__ t, uint32_t b, uint32_t e, int32_t &out) {
    if (b == e) return false;
    bool neg = false;
    if (t[b] == '+' || t[b] == '-') { neg = t[b] == '-'; ++b; if (b == e) return false; }
    int64_t v = 0;
    for (uint32_t p = b; p < e; ++p) {
        const uint8_t c = t[p];
        if (c < '0' || c > '9') return false;
        v = v * 10 + (c - '0');
        if (v > 2147483648ll) return false;
    }
    if (neg) v = -v;
    if (v > 2147483647ll) return false;
    out = (int32_t)v;
    return true;
}

// total order of the non-NaN doubles as unsigned integers, -0 == +0; NaN -> 0 (below everything, equal to nothing used)
__host__ __device__ __forceinline__ uint64_t order_bits(double x) {
    if (x != x) return 0ull;
    if (x == 0.0) x = 0.0;
    uint64_t b;
    memcpy(&b, &x, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ bool ieq(uint8_t c, char lower) { return (c | 0x20) == (uint8_t)lower; }

// f64::from_str.  Returns 0 = not a number, 1 = value in `out`, 2 = well-formed but outside the exact fast path.
__device__ int dev_parse_f64(const uint8_t *__restrict__ t, uint32_t b, uint32_t e, double &out) {
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19,
                                   1e20, 1e21, 1e22};
    if (b == e) return 0;
    bool neg = false;
    if (t[b] == '+' || t[b] == '-') { neg = t[b] == '-'; ++b; }
    if (b == e) return 0;
    const uint32_t n = e - b;
    if ((n == 3 && ieq(t[b], 'i') && ieq(t[b + 1], 'n') && ieq(t[b + 2], 'f')) ||
        (n == 8 && ieq(t[b], 'i') && ieq(t[b + 1], 'n') && ieq(t[b + 2], 'f') && ieq(t[b + 3], 'i') && ieq(t[b + 4], 'n') && ieq(t[b + 5], 'i') &&
         ieq(t[b + 6], 't') && ieq(t[b + 7], 'y'))) { out = neg ? -INFINITY : INFINITY; return 1; }
    if (n == 3 && ieq(t[b], 'n') && ieq(t[b + 1], 'a') && ieq(t[b + 2], 'n')) { out = NAN; return 1; }
    uint64_t w = 0;
    int nd = 0, sig = 0;          // digits seen, significant digits kept in w
    int e10 = 0;
    bool trunc = false;
    uint32_t p = b;
    for (; p < e && t[p] >= '0' && t[p] <= '9'; ++p) {
        ++nd;
        if (sig < 19) { w = w * 10 + (t[p] - '0'); if (w) ++sig; }
        else { ++e10; if (t[p] != '0') trunc = true; }
    }
    if (p < e && t[p] == '.') {
        ++p;
        for (; p < e && t[p] >= '0' && t[p] <= '9'; ++p) {
            ++nd;
            if (sig < 19) { w = w * 10 + (t[p] - '0'); if (w) ++sig; --e10; }
            else if (t[p] != '0') trunc = true;
        }
    }
    if (nd == 0) return 0;
    if (p < e && (t[p] == 'e' || t[p] == 'E')) {
        ++p;
        bool eneg = false;
        if (p < e && (t[p] == '+' || t[p] == '-')) { eneg = t[p] == '-'; ++p; }
        if (p == e) return 0;
        int ex = 0;
        for (; p < e; ++p) {
            if (t[p] < '0' || t[p] > '9') return 0;
            if (ex < 100000) ex = ex * 10 + (t[p] - '0');
        }
        e10 += eneg ? -ex : ex;
    }
    if (p != e) return 0;
    if (w == 0 && !trunc) { out = neg ? -0.0 : 0.0; return 1; }
    if (trunc || w >= (1ull << 53) || e10 < -22 || e10 > 22) return 2;
    double v = (double)w;                    // exact
    v = e10 >= 0 ? v * P10[e10] : v / P10[-e10];   // one correctly rounded operation on exact operands
    out = neg ? -v : v;
    return 1;
}

__global__ void __launch_bounds__(256) filter_parse_kernel(const uint8_t *__restrict__ txt, uint64_t N, uint32_t n_raw, uint32_t n_nl,
                                                           const uint32_t *__restrict__ nl_pos, FilterRec o, uint32_t *__restrict__ line_idx,
                                                           uint32_t *__restrict__ n_slow) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_raw) return;
    line_idx[i] = i;
    uint32_t p = i ? nl_pos[i - 1] + 1 : 0u;
    uint32_t le = i < n_nl ? nl_pos[i] : (uint32_t)N;
    while (p < le && is_ws(txt[p])) ++p;                // line.trim()
    while (le > p && is_ws(txt[le - 1])) --le;
    uint32_t fb[16], fe[16];
    int nf = 0;
    {                                                   // the first 16 fields; more may follow.  Tabs found eight bytes per step (TxtWin), static indices
        TxtWin tw(txt);
        uint32_t q = p;
        bool more = true;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            fb[f] = fe[f] = le;
            if (more) {
                const uint32_t t = tw.find(q, le, '\t');
                fb[f] = q; fe[f] = t; nf = f + 1;
                more = t < le;
                q = t + 1;
            }
        }
    }
    uint8_t state = 0, pass = 0;
    uint64_t h = ((uint64_t)i + 1) * 0x9E3779B97F4A7C15ull, ident = 0;   // non-records: singletons in the sort (odd multiplier = bijection)
    uint32_t m_ord = 0;
    if (nf == 16) {
        int32_t matches = 0, mapq = 0, s3 = 0, s2 = 0;
        uint32_t ib = fb[15];
        for (uint32_t c = fb[15]; c < fe[15]; ++c) if (txt[c] == ':') ib = c + 1;   // rsplit(':').next()
        double idv = 0.0;
        // the reference evaluates (and `?`-returns) in this order: matches, identity, mapq, span
        const bool ok_m = dev_parse_i32(txt, fb[9], fe[9], matches);
        const int ok_i = ok_m ? dev_parse_f64(txt, ib, fe[15], idv) : 0;
        const bool ok = ok_m && ok_i && dev_parse_i32(txt, fb[11], fe[11], mapq) && dev_parse_i32(txt, fb[3], fe[3], s3) &&
                        dev_parse_i32(txt, fb[2], fe[2], s2);
        if (ok) {
            state = ok_i == 2 ? 2 : 1;
            if (ok_i == 2) { atomicAdd(n_slow, 1u); o.f15_b[i] = ib; o.f15_e[i] = fe[15]; }
            else ident = order_bits(idv);
            const int32_t span = (int32_t)((uint32_t)s3 - (uint32_t)s2);   // release-build i32 subtraction wraps
            pass = (mapq > 20 && span > 1000) ? 1 : 0;
            m_ord = (uint32_t)matches ^ 0x80000000u;
            h = 0xcbf29ce484222325ull;
            for (uint32_t c = fb[0]; c < fe[0]; ++c) { h ^= (uint64_t)txt[c]; h *= 0x100000001b3ull; }
            h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
        }
    }
    o.state[i] = state; o.pass[i] = pass; o.hash[i] = h; o.ident[i] = ident; o.matches[i] = m_ord;
}

// sorted by (hash, line): one thread per run head walks its run
__global__ void __launch_bounds__(256) filter_pick_kernel(uint32_t n, const uint64_t *__restrict__ s_hash, const uint32_t *__restrict__ s_line,
                                                          FilterRec r, uint8_t *__restrict__ keep, uint32_t *__restrict__ n_kept) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t mine = 0;
    if (i < n && (i == 0 || s_hash[i - 1] != s_hash[i])) {
        const uint64_t h = s_hash[i];
        uint32_t bm = 0; uint64_t bi = 0; bool any = false;
        uint32_t j = i;
        for (; j < n && s_hash[j] == h; ++j) {
            const uint32_t l = s_line[j];
            if (!r.state[l]) continue;
            const uint32_t m = r.matches[l]; const uint64_t id = r.ident[l];
            if (!any || m > bm || (m == bm && id > bi)) { bm = m; bi = id; any = true; }
        }
        if (any && bi != 0) {                           // bi == 0: the best identity is NaN, nothing equals it
            for (uint32_t q = i; q < j; ++q) {          // lines of a run are in file order (stable sort)
                const uint32_t l = s_line[q];
                if (r.state[l] && r.pass[l] && r.matches[l] == bm && r.ident[l] == bi) { keep[l] = 1; mine = 1; break; }
            }
        }
    }
    if (__any(mine != 0)) {
        const uint32_t c = (uint32_t)__popcll(__ballot(mine != 0));
        if ((threadIdx.x & 63) == 0) atomicAdd(n_kept, c);
    }
}

}  // namespace

namespace {
struct FilterPiece {   // the parsed lines of one piece of the text (< 4 GiB), on the device
    uint32_t n_raw = 0;
    DevBuf<uint64_t> hash, ident;
    DevBuf<uint32_t> matches;
    DevBuf<uint8_t> state, pass;
};
__global__ void __launch_bounds__(256) iota_u32_kernel(uint32_t n, uint32_t *__restrict__ v) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = i;
}
}  // namespace

// keep_out[i] = 1 when raw line i (0-based, lines split at '\n') is written by the filter; line_end[i] = offset of the
// line's '\n' (or the text size for an unterminated last line).  Texts of 4 GiB and more are parsed in pieces cut at
// line ends (PANTAX_GAF_PIECE_BYTES lowers the piece size for tests); the grouping by read id runs over all lines.
int gaf_filter_device(Ctx *ctx, const char *text, uint64_t size, std::vector<uint8_t> &keep_out, std::vector<uint64_t> &line_end, uint64_t *n_records,
                      uint64_t *n_kept_out, int fd) {
    keep_out.clear(); line_end.clear();
    if (n_records) *n_records = 0;
    if (n_kept_out) *n_kept_out = 0;
    if (size == 0) return 0;
    uint64_t piece_max = 0xE0000000ull;   // 3.5 GiB
    if (ctx->cfg.gaf_piece_bytes && ctx->cfg.gaf_piece_bytes < piece_max) piece_max = ctx->cfg.gaf_piece_bytes;
    std::vector<std::unique_ptr<FilterPiece>> pcs;
    uint64_t n_lines = 0, nrec = 0;
    for (uint64_t off = 0; off < size;) {
        uint64_t end = std::min<uint64_t>(size, off + piece_max);
        if (end < size) {
            const void *nl = memrchr(text + off, '\n', (size_t)(end - off));
            if (!nl) return fail(ctx, PANTAX_HIP_E_LIMIT, "gaf_filter: a line of more than %llu bytes at offset %llu", (unsigned long long)piece_max, (unsigned long long)off);
            end = (uint64_t)(static_cast<const char *>(nl) - text) + 1;
        }
        const uint64_t psize = end - off;
        const char *ptext = text + off;
        DevBuf<uint8_t> d_txt;
        DevBuf<uint32_t> nl_pos, f15b, f15e, line_tmp, cnt;
        uint32_t n_nl = 0;
        PTX_TRY(gaf_upload_and_scan(ctx, ptext, psize, d_txt, nl_pos, &n_nl, fd, off));
        const uint32_t n_raw = n_nl + (ptext[psize - 1] != '\n' ? 1u : 0u);
        pcs.emplace_back(new FilterPiece());
        FilterPiece &pc = *pcs.back();
        pc.n_raw = n_raw;
        const size_t nr = n_raw ? n_raw : 1;
        PTX_HIP(ctx, pc.hash.alloc(nr)); PTX_HIP(ctx, pc.ident.alloc(nr)); PTX_HIP(ctx, pc.matches.alloc(nr)); PTX_HIP(ctx, pc.state.alloc(nr)); PTX_HIP(ctx, pc.pass.alloc(nr));
        PTX_HIP(ctx, f15b.alloc(nr)); PTX_HIP(ctx, f15e.alloc(nr)); PTX_HIP(ctx, line_tmp.alloc(nr)); PTX_HIP(ctx, cnt.alloc(1));
        PTX_HIP(ctx, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t), ctx->stream));
        FilterRec rec{pc.hash.p, pc.ident.p, pc.matches.p, f15b.p, f15e.p, pc.state.p, pc.pass.p};
        {
            KTimer t(ctx, "filter_parse_kernel");
            hipLaunchKernelGGL(filter_parse_kernel, dim3((n_raw + 255) / 256 ? (n_raw + 255) / 256 : 1), dim3(256), 0, ctx->stream, d_txt.p, psize, n_raw, n_nl,
                               nl_pos.p, rec, line_tmp.p, cnt.p);
        }
        uint32_t n_slow = 0;
        std::vector<uint8_t> h_state(n_raw);
        std::vector<uint32_t> h_nl(n_nl);
        PTX_TRY(download(ctx, &n_slow, cnt.p, 1));
        PTX_TRY(download(ctx, h_state.data(), pc.state.p, n_raw));
        PTX_TRY(download(ctx, h_nl.data(), nl_pos.p, n_nl));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint8_t st : h_state) nrec += st != 0;
        for (uint32_t i = 0; i < n_nl; ++i) line_end.push_back(off + h_nl[i]);
        if (n_raw > n_nl) line_end.push_back(size);
        if (n_slow) {   // identities outside the exact fast path: the host's correctly rounded strtod, then back
            std::vector<uint32_t> b(n_raw), e(n_raw);
            std::vector<uint64_t> idb(n_raw);
            PTX_TRY(download(ctx, b.data(), f15b.p, n_raw)); PTX_TRY(download(ctx, e.data(), f15e.p, n_raw));
            PTX_TRY(download(ctx, idb.data(), pc.ident.p, n_raw));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (uint32_t i = 0; i < n_raw; ++i) {
                if (h_state[i] != 2) continue;
                const std::string num(ptext + b[i], ptext + e[i]);   // already validated against f64::from_str's grammar
                idb[i] = order_bits(std::strtod(num.c_str(), nullptr));
            }
            PTX_HIP(ctx, hipMemcpyAsync(pc.ident.p, idb.data(), n_raw * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        n_lines += n_raw;
        off = end;
    }
    if (n_lines >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "gaf_filter: %llu lines exceed 32-bit line numbers", (unsigned long long)n_lines);
    const uint32_t n_raw = (uint32_t)n_lines;
    const size_t nr = n_raw ? n_raw : 1;
    // all lines side by side (a single piece is used as it is)
    DevBuf<uint64_t> hash, ident, hash_b;
    DevBuf<uint32_t> matches, line_a, line_b, table, scan_tmp, cnt;
    DevBuf<uint8_t> state, pass, keep;
    if (pcs.size() == 1) {
        hash.take(pcs[0]->hash); ident.take(pcs[0]->ident); matches.take(pcs[0]->matches); state.take(pcs[0]->state); pass.take(pcs[0]->pass);
    } else {
        PTX_HIP(ctx, hash.alloc(nr)); PTX_HIP(ctx, ident.alloc(nr)); PTX_HIP(ctx, matches.alloc(nr)); PTX_HIP(ctx, state.alloc(nr)); PTX_HIP(ctx, pass.alloc(nr));
        uint64_t l0 = 0;
        for (auto &pcp : pcs) {
            FilterPiece &pc = *pcp;
            if (pc.n_raw) {
                PTX_HIP(ctx, hipMemcpyAsync(hash.p + l0, pc.hash.p, pc.n_raw * 8ull, hipMemcpyDeviceToDevice, ctx->stream));
                PTX_HIP(ctx, hipMemcpyAsync(ident.p + l0, pc.ident.p, pc.n_raw * 8ull, hipMemcpyDeviceToDevice, ctx->stream));
                PTX_HIP(ctx, hipMemcpyAsync(matches.p + l0, pc.matches.p, pc.n_raw * 4ull, hipMemcpyDeviceToDevice, ctx->stream));
                PTX_HIP(ctx, hipMemcpyAsync(state.p + l0, pc.state.p, pc.n_raw, hipMemcpyDeviceToDevice, ctx->stream));
                PTX_HIP(ctx, hipMemcpyAsync(pass.p + l0, pc.pass.p, pc.n_raw, hipMemcpyDeviceToDevice, ctx->stream));
            }
            l0 += pc.n_raw;
        }
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        pcs.clear();
    }
    PTX_HIP(ctx, hash_b.alloc(nr)); PTX_HIP(ctx, line_a.alloc(nr)); PTX_HIP(ctx, line_b.alloc(nr)); PTX_HIP(ctx, keep.alloc(nr));
    PTX_HIP(ctx, table.alloc(sort_table_elems(nr))); PTX_HIP(ctx, scan_tmp.alloc(scan_tmp_elems(nr))); PTX_HIP(ctx, cnt.alloc(2));
    PTX_HIP(ctx, hipMemsetAsync(cnt.p, 0, 2 * sizeof(uint32_t), ctx->stream));
    PTX_HIP(ctx, hipMemsetAsync(keep.p, 0, nr, ctx->stream));
    const uint32_t grid = (n_raw + 255) / 256 ? (n_raw + 255) / 256 : 1;
    hipLaunchKernelGGL(iota_u32_kernel, dim3(grid), dim3(256), 0, ctx->stream, n_raw, line_a.p);
    FilterRec rec{hash.p, ident.p, matches.p, nullptr, nullptr, state.p, pass.p};
    SortBufs A, B;
    A.nw = B.nw = 1; A.k[0] = hash.p; B.k[0] = hash_b.p; A.v = line_a.p; B.v = line_b.p;
    std::vector<SortPass> passes;
    add_passes(passes, 0, 0, 64);
    bool in_b = false;
    PTX_TRY(radix_sort(ctx, A, B, n_raw, passes.data(), (int)passes.size(), table.p, scan_tmp.p, &in_b, nullptr));
    {
        KTimer t(ctx, "filter_pick_kernel");
        hipLaunchKernelGGL(filter_pick_kernel, dim3(grid), dim3(256), 0, ctx->stream, n_raw, in_b ? hash_b.p : hash.p, in_b ? line_b.p : line_a.p, rec,
                           keep.p, cnt.p + 1);
    }
    PTX_HIP(ctx, hipGetLastError());
    keep_out.resize(n_raw);
    uint32_t n_kept = 0;
    PTX_TRY(download(ctx, keep_out.data(), keep.p, n_raw));
    PTX_TRY(download(ctx, &n_kept, cnt.p + 1, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_records) *n_records = nrec;
    if (n_kept_out) *n_kept_out = n_kept;
    return 0;
}

}  // namespace ptx

using namespace ptx;

extern "C" int pantax_hip_gaf_filter(pantax_hip_ctx *ctx, const char *gaf_path, const char *out_path, uint64_t *n_lines, uint64_t *n_records,
                                     uint64_t *n_written) {
    if (!ctx || !gaf_path) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    MappedFile mf;
    const std::string err = mf.open(gaf_path);
    if (!err.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", err.c_str());
    std::string out = out_path ? out_path : "";
    if (out.empty()) {   // <stem>_filtered.gaf beside the input (gaf_filter.rs:46-49)
        const std::string in(gaf_path);
        const size_t slash = in.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "" : in.substr(0, slash + 1);
        std::string name = slash == std::string::npos ? in : in.substr(slash + 1);
        const size_t dot = name.find_last_of('.');
        if (dot != std::string::npos && dot != 0) name = name.substr(0, dot);
        out = dir + name + "_filtered.gaf";
    }
    std::vector<uint8_t> keep;
    std::vector<uint64_t> nl;   // end of every raw line
    uint64_t nrec = 0, nkept = 0;
    PTX_TRY(gaf_filter_device(ctx, mf.data, mf.size, keep, nl, &nrec, &nkept, mf.fd));
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) return fail(ctx, PANTAX_HIP_E_IO, "cannot write %s", out.c_str());
    std::vector<char> buf;
    buf.reserve(1 << 20);
    bool io_ok = true;
    for (size_t i = 0; i < keep.size(); ++i) {
        if (!keep[i]) continue;
        const uint64_t b = i ? nl[i - 1] + 1 : 0, e0 = nl[i];
        uint64_t e = e0;
        if (e > b && mf.data[e - 1] == '\r') --e;      // BufRead::lines drops "\r\n"; writeln! adds '\n'
        if (buf.size() + (e - b) + 1 > (1u << 20) && !buf.empty()) { io_ok = io_ok && std::fwrite(buf.data(), 1, buf.size(), f) == buf.size(); buf.clear(); }
        buf.insert(buf.end(), mf.data + b, mf.data + e);
        buf.push_back('\n');
    }
    if (!buf.empty()) io_ok = io_ok && std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    io_ok = (std::fclose(f) == 0) && io_ok;
    if (!io_ok) return fail(ctx, PANTAX_HIP_E_IO, "short write to %s", out.c_str());
    if (n_lines) *n_lines = keep.size();
    if (n_records) *n_records = nrec;
    if (n_written) *n_written = nkept;
    return 0;
}
