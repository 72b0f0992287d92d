// api_sort.cpp -- pantax_hip_sort_rows: the two device sorts of the LP row grouping (a10/a12) behind a host-buffer
// entry point.  Used by the tests to pin both of them against a host sort on crafted inputs (heavy ties,
// oversize buckets, all sizes); a caller may use it as a plain 3-word key sort.
#include <algorithm>
#include <vector>
#include "primitives.hpp"

using namespace ptx;

extern "C" int pantax_hip_sort_rows(pantax_hip_ctx *ctx, uint64_t n, uint64_t *k0, uint64_t *k1, uint64_t *k2, int algo) {
    if (!ctx || (n && (!k0 || !k1 || !k2))) return PANTAX_HIP_E_INVALID;
    if (n >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "sort_rows: %llu rows exceed 32-bit positions", (unsigned long long)n);
    PTX_ENTER(ctx);
    if (n == 0) return 0;
    if (algo == 2 && n > SS_MAX_N) return fail(ctx, PANTAX_HIP_E_LIMIT, "sort_rows: the sample sort takes at most %llu rows", (unsigned long long)SS_MAX_N);
    if (algo == 3) {   // segmented: k0 = segment id (rows of a segment adjacent, ids ascending), (k1, k2) sorted inside every segment
        std::vector<uint32_t> off, cnt;
        uint64_t bound = 0;
        for (uint64_t i = 0; i < n;) {
            uint64_t j = i;
            while (j < n && k0[j] == k0[i]) ++j;
            if (j < n && k0[j] < k0[i]) return fail(ctx, PANTAX_HIP_E_INVALID, "sort_rows: algo 3 takes rows grouped by ascending k0");
            off.push_back((uint32_t)i); cnt.push_back((uint32_t)(j - i));
            bound = std::max<uint64_t>(bound, j - i);
            i = j;
        }
        DevBuf<uint64_t> a1, a2, b1, b2;
        DevBuf<uint32_t> d_off, d_cnt, wsb;
        PTX_TRY(upload(ctx, a1, k1, n)); PTX_TRY(upload(ctx, a2, k2, n));
        PTX_HIP(ctx, b1.alloc(n)); PTX_HIP(ctx, b2.alloc(n));
        PTX_TRY(upload(ctx, d_off, off.data(), off.size())); PTX_TRY(upload(ctx, d_cnt, cnt.data(), cnt.size()));
        PTX_HIP(ctx, wsb.alloc(sample_sort_seg_ws_elems((uint32_t)off.size(), n)));
        PTX_TRY(sample_sort_seg(ctx, a1.p, a2.p, b1.p, b2.p, (uint32_t)off.size(), bound, n, d_off.p, d_cnt.p, wsb.p));
        PTX_TRY(download(ctx, k1, a1.p, n)); PTX_TRY(download(ctx, k2, a2.p, n));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    const bool sample = algo == 2 || (algo == 0 && n <= SS_MAX_N);
    DevBuf<uint64_t> a[3], b[3];
    DevBuf<uint32_t> ws, tmp, dn;
    uint64_t *h[3] = {k0, k1, k2};
    for (int w = 0; w < 3; ++w) { PTX_TRY(upload(ctx, a[w], h[w], n)); PTX_HIP(ctx, b[w].alloc(n)); }
    const uint32_t n32 = (uint32_t)n;
    PTX_TRY(upload(ctx, dn, &n32, 1));
    SortBufs A, B;
    A.nw = B.nw = 3;
    for (int w = 0; w < 3; ++w) { A.k[w] = a[w].p; B.k[w] = b[w].p; }
    bool in_b = false;
    if (sample) {
        PTX_HIP(ctx, ws.alloc(sample_sort_ws_elems(n)));
        PTX_TRY(sample_sort3(ctx, A, B, n, ws.p, dn.p));
    } else {
        std::vector<SortPass> passes;
        add_passes(passes, 2, 0, 64);
        add_passes(passes, 1, 0, 64);
        add_passes(passes, 0, 0, 64);
        PTX_HIP(ctx, ws.alloc(sort_table_elems(n)));
        PTX_HIP(ctx, tmp.alloc(16));
        PTX_TRY(radix_sort(ctx, A, B, n, passes.data(), (int)passes.size(), ws.p, tmp.p, &in_b, dn.p));
    }
    const SortBufs &R = in_b ? B : A;
    for (int w = 0; w < 3; ++w) PTX_TRY(download(ctx, h[w], R.k[w], n));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
