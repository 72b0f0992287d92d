// gaf_prune.cc -- host side of the GAF load: the columns the path never reads do not travel over PCIe.
//
// load_gaf_file_lazy (rcls.rs:119-137) keeps columns 1, 2, 6, 7, 8, 9 and 12 of a GAF line; the device tokenizer
// (stage_gaf.hip: gaf_parse_kernel) accordingly looks at fields 0, 1, 5, 6, 7, 8 and 11 of the first twelve
// tab-separated fields of a line and at nothing behind them.  prune_lines() rewrites every line that HAS twelve fields as
//     f0 \t f1 \t \t \t \t f5 \t f6 \t f7 \t f8 \t \t \t f11 \n
// -- the unread fields left empty, the tags cut off -- and copies every other line (fewer fields, comments, empty lines)
// byte for byte.  The tokenizer therefore parses the pruned text with the rules it applies to the original text, field
// for field, and yields the same columns: the format quirks (`*`, ragged rows, `@`, CR LF) need no second
// implementation.  What changes are the byte positions of the read ids, so a caller that wants the id spans (the binning
// report) loads the text unpruned.  Plain C++ (compiled without -x hip) so that the scan can use AVX2.
#include <cstdint>
#include <cstring>
#include <immintrin.h>

namespace ptx {

namespace {
// positions of '\t' and '\n' in [p, e): 32 bytes per step; calls on_sep(position, is_newline) in order, stops when it returns false
template <class F>
__attribute__((target("avx2"))) inline const uint8_t *scan_seps_avx2(const uint8_t *p, const uint8_t *e, F &&on_sep) {
    const __m256i vt = _mm256_set1_epi8('\t'), vn = _mm256_set1_epi8('\n');
    while (p + 32 <= e) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v, vt), _mm256_cmpeq_epi8(v, vn)));
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            if (!on_sep(p + b, p[b] == '\n')) return p + b + 1;
        }
        p += 32;
    }
    for (; p < e; ++p)
        if (*p == '\t' || *p == '\n') { if (!on_sep(p, *p == '\n')) return p + 1; }
    return e;
}
template <class F>
inline const uint8_t *scan_seps_scalar(const uint8_t *p, const uint8_t *e, F &&on_sep) {
    for (; p < e; ++p)
        if (*p == '\t' || *p == '\n') { if (!on_sep(p, *p == '\n')) return p + 1; }
    return e;
}

struct LineState {
    const uint8_t *line = nullptr;   // start of the current line
    const uint8_t *tab[11];          // its first eleven tabs
    int n_tab = 0;
};

// one complete line [b, nl) (nl = its '\n', or the end of the text when the last line has none) with its first n_tab <= 11
// tab positions -> out; returns the new end of out.  COUNT: nothing is written, `out` only advances (the size pass of the two-pass
// upload: every thread learns where its part of the chunk starts before anything is copied)
template <bool COUNT>
inline uint8_t *emit_line(const uint8_t *b, const uint8_t *nl, bool has_nl, const uint8_t *const *tab, int n_tab, uint8_t *out) {
    const uint8_t *le = nl;
    if (le > b && le[-1] == '\r') --le;                       // the tokenizer strips it before it splits the line
    if (n_tab < 11 || le == b || *b == '@' || tab[10] >= le) {   // fewer than twelve fields, empty, comment: byte for byte
        const size_t n = (size_t)(nl - b) + (has_nl ? 1u : 0u);
        if (!COUNT) std::memcpy(out, b, n);
        return out + n;
    }
    auto put = [&](const uint8_t *fb, const uint8_t *fe) { if (!COUNT) std::memcpy(out, fb, (size_t)(fe - fb)); out += fe - fb; };
    put(b, tab[0]); if (!COUNT) *out = '\t'; ++out;           // f0
    put(tab[0] + 1, tab[1]);                                  // f1
    if (!COUNT) std::memcpy(out, "\t\t\t\t", 4); out += 4;    // f2 f3 f4 empty
    put(tab[4] + 1, tab[8]);                                  // f5 \t f6 \t f7 \t f8
    if (!COUNT) std::memcpy(out, "\t\t\t", 3); out += 3;      // f9 f10 empty
    const uint8_t *f11e = static_cast<const uint8_t *>(std::memchr(tab[10] + 1, '\t', (size_t)(le - (tab[10] + 1))));
    put(tab[10] + 1, f11e ? f11e : le);                       // f11, the tags behind it cut off
    if (has_nl) { if (!COUNT) *out = '\n'; ++out; }
    return out;
}
}  // namespace

// Lines that START in [begin, end) of text [0, size), which must begin at a line start when begin == 0 -- the caller says whether
// `begin` is a line start otherwise (byte begin - 1 is '\n').  A line that starts before `end` is finished beyond it.  out must hold
// (end of the last such line) - (start of the first) bytes; returns the bytes written.  out == nullptr: the same count, nothing written.
template <bool COUNT>
static uint64_t prune_range(const uint8_t *text, uint64_t size, uint64_t begin, uint64_t end, bool begin_is_line_start, uint8_t *out) {
    if (begin >= size || begin >= end) return 0;
    const uint8_t *p = text + begin, *const e_all = text + size, *const e_own = text + (end < size ? end : size);
    if (!begin_is_line_start) {                               // skip the tail of a line that belongs to the range before
        const void *nl = std::memchr(p, '\n', (size_t)(e_all - p));
        if (!nl) return 0;
        p = static_cast<const uint8_t *>(nl) + 1;
    }
    static const bool avx2 = __builtin_cpu_supports("avx2");
    uint8_t *o = out;
    LineState st;
    while (p < e_own) {                                       // p = start of a line this range owns
        st.line = p; st.n_tab = 0;
        const uint8_t *nl = nullptr;
        auto on_sep = [&](const uint8_t *at, bool is_nl) {
            if (is_nl) { nl = at; return false; }
            if (st.n_tab < 11) st.tab[st.n_tab++] = at;
            return true;
        };
        // first the twelve fields (tabs and the newline in one pass), then -- tags can be long -- memchr for the line end
        const uint8_t *q = p;
        while (!nl && q < e_all && st.n_tab < 11) {
            const uint8_t *stop = q + 256 < e_all ? q + 256 : e_all;
            auto lim = [&](const uint8_t *at, bool is_nl) { const bool go = on_sep(at, is_nl); return go && st.n_tab < 11; };
            q = avx2 ? scan_seps_avx2(q, stop, lim) : scan_seps_scalar(q, stop, lim);
        }
        if (!nl && q < e_all) nl = static_cast<const uint8_t *>(std::memchr(q, '\n', (size_t)(e_all - q)));
        const bool has_nl = nl != nullptr;
        if (!nl) nl = e_all;
        o = emit_line<COUNT>(st.line, nl, has_nl, st.tab, st.n_tab, o);
        p = nl + 1;
    }
    return (uint64_t)(o - out);
}
uint64_t gaf_prune_range(const uint8_t *text, uint64_t size, uint64_t begin, uint64_t end, bool begin_is_line_start, uint8_t *out) {
    // (the count pass advances a pointer that is never dereferenced: it starts at the text so that the arithmetic stays inside an object)
    if (!out) return prune_range<true>(text, size, begin, end, begin_is_line_start, const_cast<uint8_t *>(text));
    return prune_range<false>(text, size, begin, end, begin_is_line_start, out);
}

}  // namespace ptx
