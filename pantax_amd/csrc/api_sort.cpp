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
    if (algo == 3) return fail(ctx, PANTAX_HIP_E_INVALID, "sort_rows: algo 3 (round 3's segmented sort) is gone; 4 / 5 are the batched sort of the many-species step");
    if (algo == 4 || algo == 5) {   // the node-order sort: an entry with k1 == 0 or k2 not the bits of a positive double is no row and is dropped
        std::vector<uint32_t> base{0};
        std::vector<uint64_t> seg_val;
        uint64_t bound = 0;
        for (uint64_t i = 0; i < n;) {
            uint64_t j = i;
            while (j < n && k0[j] == k0[i]) ++j;
            if (j < n && k0[j] < k0[i]) return fail(ctx, PANTAX_HIP_E_INVALID, "sort_rows: algo 4 takes entries grouped by ascending k0");
            seg_val.push_back(k0[i]); base.push_back((uint32_t)j);
            bound = std::max<uint64_t>(bound, j - i);
            i = j;
        }
        const uint32_t S = (uint32_t)seg_val.size();
        const int pack_shift = algo == 5 ? 8 : -1;   // algo 5: masks below 256, the segment number travels in the mask word
        if (algo == 5) for (uint64_t i = 0; i < n; ++i) if (k1[i] >> 8) return fail(ctx, PANTAX_HIP_E_INVALID, "sort_rows: algo 5 takes k1 < 256");
        DevBuf<uint64_t> dm, da, r16, osp, om, oa;
        DevBuf<uint32_t> d_base, wsb, dn;
        PTX_TRY(upload(ctx, dm, k1, n)); PTX_TRY(upload(ctx, da, k2, n));
        PTX_TRY(upload(ctx, d_base, base.data(), base.size()));
        PTX_HIP(ctx, r16.alloc(4 * n)); PTX_HIP(ctx, osp.alloc(n)); PTX_HIP(ctx, om.alloc(n)); PTX_HIP(ctx, oa.alloc(n)); PTX_HIP(ctx, dn.alloc(1));
        PTX_HIP(ctx, wsb.alloc(sample_sort_nodes_ws_elems(S, bound, n)));
        PTX_TRY(sample_sort_nodes(ctx, reinterpret_cast<const double *>(da.p), dm.p, d_base.p, S, bound, n, r16.p, pack_shift >= 0 ? (uint64_t *)nullptr : osp.p, om.p, oa.p,
                                  pack_shift, wsb.p, dn.p));
        uint32_t nv = 0;
        PTX_TRY(download(ctx, &nv, dn.p, 1));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> hs(nv), hm(nv), ha(nv);
        if (nv) {
            if (pack_shift < 0) PTX_TRY(download(ctx, hs.data(), osp.p, nv));
            PTX_TRY(download(ctx, hm.data(), om.p, nv)); PTX_TRY(download(ctx, ha.data(), oa.p, nv));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        for (uint64_t i = 0; i < n; ++i) {
            if (i < nv) {
                const uint64_t sp = pack_shift >= 0 ? (hm[i] >> pack_shift) : hs[i];
                if (sp >= S) return fail(ctx, PANTAX_HIP_E_STATE, "sort_rows: segment %llu of %u in the output", (unsigned long long)sp, S);
                k0[i] = seg_val[sp]; k1[i] = pack_shift >= 0 ? (hm[i] & 0xFFull) : hm[i]; k2[i] = ha[i];
            } else k0[i] = k1[i] = k2[i] = 0;
        }
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
