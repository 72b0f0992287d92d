// api_core.cpp -- upload of the resident DB / reads and the a2/a3/a8 entry points of
// include/pantax_hip.h.  Host code only; kernels live in the stage_*.hip files.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include "common.hpp"

using namespace ptx;

extern "C" {

int pantax_hip_db_upload(pantax_hip_ctx *ctx, const pantax_hip_graphs *g, pantax_hip_db **out) {
    if (!ctx || !g || !out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    uint32_t S = g->n_species;
    if (S == 0) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: n_species == 0");
    std::unique_ptr<pantax_hip_db> db(new pantax_hip_db());
    db->S = S;
    if (!g->node_len) {
        // ranges-only db: enough for read binning (a2/a3) over ALL species of species_range.txt without
        // loading any graph; the strain stages need a full db of the selected species
        std::vector<uint32_t> order(S), rs(S), re(S);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return g->range_start[a] < g->range_start[b]; });
        bool disjoint = true;
        for (uint32_t i = 0; i < S; ++i)
            if (g->range_start[i] < 0 || g->range_end[i] > 0xFFFFFFFFll) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u has an invalid node id range", i);
        for (uint32_t i = 0; i + 1 < S; ++i) if (g->range_end[order[i]] >= g->range_start[order[i + 1]]) disjoint = false;
        db->ranges_sorted_disjoint = disjoint;
        if (!disjoint) std::iota(order.begin(), order.end(), 0u);
        for (uint32_t i = 0; i < S; ++i) { rs[i] = (uint32_t)g->range_start[order[i]]; re[i] = (uint32_t)g->range_end[order[i]]; }
        db->h_range_start.assign(g->range_start, g->range_start + S);
        db->h_range_end.assign(g->range_end, g->range_end + S);
        PTX_TRY(upload(ctx, db->d_rng_start, rs.data(), S));
        PTX_TRY(upload(ctx, db->d_rng_end, re.data(), S));
        PTX_TRY(upload(ctx, db->d_rng_idx, order.data(), S));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *out = db.release();
        return 0;
    }
    // full db: one part per species, pointing into the caller's flat arrays
    std::vector<GraphPart> parts(S);
    for (uint32_t s = 0; s < S; ++s) {
        const uint64_t h0 = g->hap_off[s], h1 = g->hap_off[s + 1], nn = g->node_off[s + 1] - g->node_off[s];
        GraphPart &pt = parts[s];
        pt.n_nodes = nn; pt.n_haps = h1 - h0; pt.path_off = g->path_off + h0;
        pt.len_seg.src = g->node_len + g->node_off[s]; pt.len_seg.out_bytes = 4 * nn; pt.len_seg.narrow = true;
        UploadSeg w;
        w.src = g->path_nodes + g->path_off[h0]; w.out_bytes = 4 * (g->path_off[h1] - g->path_off[h0]);
        pt.walk_segs.push_back(w);
    }
    db.reset();
    return db_upload_parts(ctx, S, g->range_start, g->range_end, parts.data(), nullptr, out);
}

int pantax_hip_db_upload_parts(pantax_hip_ctx *ctx, uint32_t n_species, const int64_t *range_start, const int64_t *range_end,
                               const pantax_hip_graph_part *gp, pantax_hip_db **out) {
    if (!ctx || !range_start || !range_end || !gp || !out || n_species == 0) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    std::vector<GraphPart> parts(n_species);
    for (uint32_t s = 0; s < n_species; ++s) {
        if (!gp[s].path_off || (gp[s].n_nodes && !gp[s].node_len)) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload_parts: species %u has a null array", s);
        GraphPart &pt = parts[s];
        pt.n_nodes = gp[s].n_nodes; pt.n_haps = gp[s].n_haps; pt.path_off = gp[s].path_off;
        pt.len_seg.src = gp[s].node_len; pt.len_seg.out_bytes = 4 * gp[s].n_nodes; pt.len_seg.narrow = true;
        UploadSeg w;
        w.src = gp[s].path_nodes; w.out_bytes = 4 * (gp[s].path_off[gp[s].n_haps] - gp[s].path_off[0]);
        if (w.out_bytes && !w.src) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload_parts: species %u has a null array", s);
        pt.walk_segs.push_back(w);
    }
    return db_upload_parts(ctx, n_species, range_start, range_end, parts.data(), nullptr, out);
}

}  // extern "C"

namespace ptx {

// The resident DB from one part per species (the file seam hands over where every species' arrays lie: nothing is parsed,
// concatenated or checked on the host).  part.path_off[h] - part.path_off[0] indexes the species' walks.
int db_upload_parts(Ctx *ctx, uint32_t S, const int64_t *range_start, const int64_t *range_end, const GraphPart *parts, const std::string *files,
                    pantax_hip_db **out) {
    *out = nullptr;
    pantax_hip_db *db = nullptr;
    PTX_TRY(db_upload_begin(ctx, S, range_start, range_end, parts, &db));
    std::unique_ptr<pantax_hip_db> hold(db);
    const auto t0 = std::chrono::steady_clock::now();
    PTX_TRY(db_upload_arrays(ctx, db, parts, files, nullptr));
    if (ctx->cfg.trace) std::fprintf(stderr, "[db_upload]            %-28s %9.3f ms\n", "graph arrays -> HBM", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    PTX_TRY(db_upload_finish(ctx, db));
    *out = hold.release();
    return 0;
}

int db_upload_begin(Ctx *ctx, uint32_t S, const int64_t *range_start, const int64_t *range_end, const GraphPart *parts, pantax_hip_db **out) {
    *out = nullptr;
    std::unique_ptr<pantax_hip_db> db(new pantax_hip_db());
    db->S = S;
    db->h_node_off.assign(S + 1, 0); db->h_hap_off.assign(S + 1, 0);
    for (uint32_t s = 0; s < S; ++s) { db->h_node_off[s + 1] = db->h_node_off[s] + parts[s].n_nodes; db->h_hap_off[s + 1] = db->h_hap_off[s] + parts[s].n_haps; }
    db->V = db->h_node_off[S];
    db->H = db->h_hap_off[S];
    db->h_path_off.assign(db->H + 1, 0);
    for (uint32_t s = 0; s < S; ++s)
        for (uint64_t h = 0; h < parts[s].n_haps; ++h) {
            const uint64_t gh = db->h_hap_off[s] + h;
            db->h_path_off[gh + 1] = db->h_path_off[gh] + (parts[s].path_off[h + 1] - parts[s].path_off[h]);
        }
    db->P = db->h_path_off[db->H];
    // a row's owner within its species travels in 16 bits (d_trio_hap, the statistics by key) and the basis inverse of the LP is indexed with 32:
    // the bound lad_prepare states, checked here so that no stage entry point ever sees a truncated owner
    for (uint32_t s = 0; s < S; ++s)
        if (parts[s].n_haps > 30000ull)
            return fail(ctx, PANTAX_HIP_E_LIMIT, "db_upload: species %u has %llu haplotypes (limit 30000 per species)", s, (unsigned long long)parts[s].n_haps);
    if (db->V >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "db_upload: %llu nodes on one GPU exceeds the 32-bit node index", (unsigned long long)db->V);
    db->h_range_start.assign(range_start, range_start + S);
    db->h_range_end.assign(range_end, range_end + S);
    for (uint32_t s = 0; s < S; ++s) {
        if (range_start[s] < 1 || range_end[s] > 0xFFFFFFFFll || range_end[s] < range_start[s])
            return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u has an invalid node id range [%lld,%lld]", s, (long long)range_start[s], (long long)range_end[s]);
        // optimize_otu derives nvert from the range (profile.rs:2938); the graph must agree
        if ((uint64_t)(range_end[s] - range_start[s] + 1) != parts[s].n_nodes)
            return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u range spans %lld ids but its graph has %llu nodes", s, (long long)(range_end[s] - range_start[s] + 1), (unsigned long long)parts[s].n_nodes);
    }
    // binning table: sorted by start when the ranges are pairwise disjoint (sort_range.rs:25-33
    // builds them contiguous), otherwise file order + linear scan
    std::vector<uint32_t> order(S);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return range_start[a] < range_start[b]; });
    bool disjoint = true;
    for (uint32_t i = 0; i + 1 < S; ++i)
        if (range_end[order[i]] >= range_start[order[i + 1]]) disjoint = false;
    db->ranges_sorted_disjoint = disjoint;
    if (!disjoint) std::iota(order.begin(), order.end(), 0u);
    std::vector<uint32_t> rs(S), re(S), first_id(S), node_base(S + 1), hap_species(db->H);
    for (uint32_t i = 0; i < S; ++i) {
        rs[i] = (uint32_t)range_start[order[i]];
        re[i] = (uint32_t)range_end[order[i]];
        first_id[i] = (uint32_t)range_start[i];
        node_base[i] = (uint32_t)db->h_node_off[i];
        for (uint64_t h = db->h_hap_off[i]; h < db->h_hap_off[i + 1]; ++h) hap_species[h] = i;
    }
    node_base[S] = (uint32_t)db->V;
    PTX_TRY(upload(ctx, db->d_rng_start, rs.data(), S));
    PTX_TRY(upload(ctx, db->d_rng_end, re.data(), S));
    PTX_TRY(upload(ctx, db->d_rng_idx, order.data(), S));
    PTX_TRY(upload(ctx, db->d_sp_first_id, first_id.data(), S));
    PTX_TRY(upload(ctx, db->d_node_base, node_base.data(), S + 1));
    PTX_TRY(upload(ctx, db->d_path_off, db->h_path_off.data(), db->H + 1));
    PTX_TRY(upload(ctx, db->d_hap_species, hap_species.data(), db->H));
    PTX_TRY(upload(ctx, db->d_hap_off, db->h_hap_off.data(), S + 1));
    db->h_all_same.assign(S, 0);
    for (uint32_t s = 0; s < S; ++s) db->h_all_same[s] = parts[s].n_haps >= 2;   // until a walk differs (walks_same_kernel)
    PTX_TRY(upload(ctx, db->d_all_same, db->h_all_same.data(), S));
    PTX_HIP(ctx, db->d_bit_off.alloc(db->V + 1)); PTX_HIP(ctx, db->d_node_len.alloc(db->V)); PTX_HIP(ctx, db->d_node_rec.alloc(db->V));
    PTX_HIP(ctx, db->d_path_nodes.alloc(db->P));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the small uploads above are over: whoever loads the arrays (another thread, another stream) finds the tables in place
    *out = db.release();
    return 0;
}

int db_upload_arrays(Ctx *ctx, pantax_hip_db *db, const GraphPart *parts, const std::string *files, hipStream_t stream_arg) {
    const uint32_t S = db->S;
    const hipStream_t stream = stream_arg ? stream_arg : ctx->stream;
    // small tables go through plain asynchronous copies from vectors that live until the closing wait (not through the ctx's staging buffers:
    // this may be a loader thread beside the thread that owns them)
    auto put = [&](void *dst, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream) : hipSuccess; };
    {   // the two big arrays: every species' stretch of them, back to back, through ONE chunk pipeline each.  Species that come PACKED (image
        // format 4: 16-bit lengths, walks as blocks of deltas) send their packed sections through pipelines of their own into scratch, and
        // kernels unpack them into their stretches of the arrays; in the plain pipelines those stretches are holes (nothing is filled for them).
        std::vector<UploadSeg> segs;
        std::vector<uint32_t> seg_species;
        uint32_t n_len16 = 0, n_packed = 0;
        for (uint32_t s = 0; s < S; ++s) { n_len16 += parts[s].len16 ? 1u : 0u; n_packed += parts[s].packed ? 1u : 0u; }
        int64_t bad = -1;
        // ---- node lengths: the plain pipeline first (holes travel as garbage), then the 16-bit stretches and the kernel that widens them into place
        if (n_len16 < S) {
            for (uint32_t s = 0; s < S; ++s) {
                UploadSeg sg = parts[s].len_seg;
                if (parts[s].len16) { sg = UploadSeg(); sg.hole = true; sg.out_bytes = 4 * parts[s].n_nodes; }
                else if (sg.out_bytes != 4 * parts[s].n_nodes) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u: the node lengths do not cover its %llu nodes", s, (unsigned long long)parts[s].n_nodes);
                segs.push_back(sg); seg_species.push_back(s);
            }
            PTX_TRY(upload_segments(ctx, db->d_node_len.p, segs.data(), segs.size(), files, &bad, stream));
            if (bad >= 0) return fail(ctx, PANTAX_HIP_E_LIMIT, "db_upload: species %u holds a node of negative length or longer than 2^32 - 1 (reference asserts > 0, profile.rs:494)", seg_species[bad]);
            segs.clear(); seg_species.clear();
        }
        DevBuf<uint16_t> d_len16;
        DevBuf<WidenSpecies> d_wt;
        if (n_len16) {
            std::vector<WidenSpecies> wt;
            uint64_t at = 0;
            for (uint32_t s = 0; s < S; ++s) {
                if (!parts[s].len16) continue;
                if (parts[s].len_seg.out_bytes != ((2 * parts[s].n_nodes + 3) & ~3ull)) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u: the 16-bit node lengths do not cover its %llu nodes", s, (unsigned long long)parts[s].n_nodes);
                wt.push_back(WidenSpecies{at, db->h_node_off[s], parts[s].n_nodes});
                segs.push_back(parts[s].len_seg);
                at += parts[s].len_seg.out_bytes / 2;
            }
            PTX_HIP(ctx, d_len16.alloc(at ? at : 1));
            PTX_HIP(ctx, d_wt.alloc(wt.size()));
            PTX_HIP(ctx, put(d_wt.p, wt.data(), wt.size() * sizeof(WidenSpecies)));
            PTX_TRY(upload_segments(ctx, d_len16.p, segs.data(), segs.size(), files, nullptr, stream));
            PTX_TRY(lens_widen_launch(ctx, d_wt.p, (uint32_t)wt.size(), at, d_len16.p, db->d_node_len.p, stream_arg));
            PTX_HIP(ctx, hipStreamSynchronize(stream));   // the scratch and its table go out of scope
            segs.clear();
        }
        // ---- walks
        DevBuf<uint32_t> d_pk_first, d_pk_off;
        DevBuf<uint8_t> d_pk_payload;
        DevBuf<UnpackSpecies> d_ut;
        std::vector<UnpackSpecies> ut;
        uint64_t nb_tot = 0, noff_tot = 0, pay_units = 0;
        if (n_packed) {
            std::vector<UploadSeg> s_first, s_off, s_pay;
            for (uint32_t s = 0; s < S; ++s) {
                if (!parts[s].packed) continue;
                const PackedWalks &pk = parts[s].pk;
                const uint64_t Ps = db->h_path_off[db->h_hap_off[s + 1]] - db->h_path_off[db->h_hap_off[s]];
                if (pk.n_blocks != (Ps + PK_BLOCK - 1) / PK_BLOCK || pk.payload_bytes % PK_UNIT || pk.first_seg.out_bytes != 4 * pk.n_blocks ||
                    pk.off_seg.out_bytes != 4 * (pk.n_blocks + 1) || pk.payload_seg.out_bytes != pk.payload_bytes)
                    return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u: the packed walks do not cover its path steps", s);
                if (nb_tot + pk.n_blocks >= 0xFFFFFFFFull || pay_units + pk.payload_bytes / PK_UNIT >= 0xFFFFFFFFull)
                    return fail(ctx, PANTAX_HIP_E_LIMIT, "db_upload: more than 2^32 blocks of packed walks");
                ut.push_back(UnpackSpecies{(uint32_t)nb_tot, (uint32_t)noff_tot, (uint32_t)pay_units, (uint32_t)db->h_path_off[db->h_hap_off[s]], (uint32_t)Ps});
                s_first.push_back(pk.first_seg); s_off.push_back(pk.off_seg); s_pay.push_back(pk.payload_seg);
                nb_tot += pk.n_blocks; noff_tot += pk.n_blocks + 1; pay_units += pk.payload_bytes / PK_UNIT;
            }
            PTX_HIP(ctx, d_pk_first.alloc(nb_tot ? nb_tot : 1)); PTX_HIP(ctx, d_pk_off.alloc(noff_tot ? noff_tot : 1)); PTX_HIP(ctx, d_pk_payload.alloc(pay_units ? pay_units * PK_UNIT : 1));
            PTX_HIP(ctx, d_ut.alloc(ut.size()));
            PTX_HIP(ctx, put(d_ut.p, ut.data(), ut.size() * sizeof(UnpackSpecies)));
            PTX_TRY(upload_segments(ctx, d_pk_first.p, s_first.data(), s_first.size(), files, nullptr, stream));
            PTX_TRY(upload_segments(ctx, d_pk_off.p, s_off.data(), s_off.size(), files, nullptr, stream));
            PTX_TRY(upload_segments(ctx, d_pk_payload.p, s_pay.data(), s_pay.size(), files, nullptr, stream));
        }
        if (n_packed < S) {
            for (uint32_t s = 0; s < S; ++s) {
                const uint64_t want = 4 * (db->h_path_off[db->h_hap_off[s + 1]] - db->h_path_off[db->h_hap_off[s]]);
                if (parts[s].packed) { UploadSeg sg; sg.hole = true; sg.out_bytes = want; segs.push_back(sg); seg_species.push_back(s); continue; }
                uint64_t bytes = 0;
                for (const UploadSeg &w : parts[s].walk_segs) { segs.push_back(w); seg_species.push_back(s); bytes += w.out_bytes; }
                if (bytes != want) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: species %u: the walks do not cover its path steps", s);
            }
            PTX_TRY(upload_segments(ctx, db->d_path_nodes.p, segs.data(), segs.size(), files, &bad, stream));
            if (bad >= 0) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: a haplotype of species %u walks a node index beyond 2^32", seg_species[bad]);
        }
        if (n_packed) {   // (behind the plain pipeline: the holes' garbage is overwritten)
            PTX_TRY(walks_unpack_launch(ctx, d_ut.p, (uint32_t)ut.size(), (uint32_t)nb_tot, d_pk_first.p, d_pk_off.p, d_pk_payload.p, db->d_path_nodes.p, stream_arg));
            PTX_HIP(ctx, hipStreamSynchronize(stream));   // the scratch goes out of scope
        }
    }
    PTX_HIP(ctx, hipStreamSynchronize(stream));
    return 0;
}

int db_upload_finish(Ctx *ctx, pantax_hip_db *db_raw) {
    Db *const db = db_raw;
    const uint32_t S = db->S;
    const bool trace = ctx->cfg.trace;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[db_upload]            %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    {   // node tables, the zero-length check, the walk check and the identical-walk test (first_filter_paths, profile.rs:1188-1190:
        // a property of the graphs, done once) on the device
        DevBuf<uint32_t> d_flags;
        PTX_HIP(ctx, d_flags.alloc(2));
        PTX_HIP(ctx, hipMemsetAsync(d_flags.p, 0, sizeof(uint32_t), ctx->stream));
        PTX_HIP(ctx, hipMemsetAsync(d_flags.p + 1, 0xFF, sizeof(uint32_t), ctx->stream));
        PTX_TRY(node_tables_launch(ctx, db, d_flags.p));
        uint32_t fl[2] = {0, 0};
        uint64_t L = 0;
        PTX_TRY(download(ctx, fl, d_flags.p, 2));
        PTX_TRY(download(ctx, &L, db->d_bit_off.p + db->V, 1));
        PTX_TRY(download(ctx, db->h_all_same.data(), db->d_all_same.p, S));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (fl[0]) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: a graph holds a node of length 0 (reference asserts > 0, profile.rs:494)");
        if (fl[1] != 0xFFFFFFFFu) return fail(ctx, PANTAX_HIP_E_INVALID, "db_upload: hap %u walks a node outside its species graph", fl[1] - 1);
        db->L = L;
        if (db->L >= NODE_REC_MAX_BITS) return fail(ctx, PANTAX_HIP_E_LIMIT, "db_upload: %llu graph bases on one GPU (limit 2^40)", (unsigned long long)db->L);
    }
    lap("node tables + checks (device)");
    {   // path tiles, ordered (species, chunk group, hap, chunk in group): workgroup b runs on XCD b % 8, so the
        // tiles of ONE chunk (all haplotypes, largely collinear: same node buckets, same path neighbourhood) get
        // workgroup ids that are congruent mod 8 -- their bucket / mask writes meet in one XCD's L2 instead of
        // being written back piecemeal from eight
        std::vector<uint2> tiles;
        constexpr uint64_t XCD = 8;
        for (uint32_t s = 0; s < S; ++s) {
            uint64_t maxlen = 0;
            for (uint64_t h = db->h_hap_off[s]; h < db->h_hap_off[s + 1]; ++h) maxlen = std::max<uint64_t>(maxlen, db->h_path_off[h + 1] - db->h_path_off[h]);
            const uint64_t n_chunks = (maxlen + PATH_TILE - 1) / PATH_TILE;
            for (uint64_t c0 = 0; c0 < n_chunks; c0 += XCD) {
                // pad the workgroup ids so that the group starts on XCD 0
                while (tiles.size() % XCD) tiles.push_back(make_uint2(0xFFFFFFFFu, 0u));
                for (uint64_t h = db->h_hap_off[s]; h < db->h_hap_off[s + 1]; ++h)
                    for (uint64_t c = c0; c < c0 + XCD; ++c) {
                        const bool live = c < n_chunks && c * PATH_TILE < db->h_path_off[h + 1] - db->h_path_off[h];
                        tiles.push_back(live ? make_uint2((uint32_t)h, (uint32_t)c) : make_uint2(0xFFFFFFFFu, 0u));
                    }
            }
        }
        db->n_tiles = tiles.size();
        PTX_TRY(upload(ctx, db->d_tiles, tiles.data(), tiles.size()));
        // the same tiles numbered in path order (hap-major): rows of the trio table are numbered in that order
        std::vector<uint32_t> hap_tile_off(db->H + 1, 0), tile_rank(tiles.size());
        for (uint64_t h = 0; h < db->H; ++h)
            hap_tile_off[h + 1] = hap_tile_off[h] + (uint32_t)((db->h_path_off[h + 1] - db->h_path_off[h] + PATH_TILE - 1) / PATH_TILE);
        for (size_t i = 0; i < tiles.size(); ++i) tile_rank[i] = tiles[i].x == 0xFFFFFFFFu ? hap_tile_off[db->H] : hap_tile_off[tiles[i].x] + tiles[i].y;   // pads count into the spare last slot
        PTX_TRY(upload(ctx, db->d_tile_rank, tile_rank.data(), tile_rank.size()));
        PTX_TRY(upload(ctx, db->d_hap_tile_off, hap_tile_off.data(), hap_tile_off.size()));
    }
    {   // node tiles of the LP row compaction (2048 nodes each): which species they start and end in
        const uint64_t TILE = 2048, nt = (db->V + TILE - 1) / TILE;
        std::vector<uint2> tsp(nt ? nt : 1, make_uint2(0u, 0u));
        uint32_t sp = 0;
        for (uint64_t t = 0; t < nt; ++t) {
            const uint64_t v0 = t * TILE, v1 = std::min<uint64_t>(db->V, v0 + TILE) - 1;
            while (sp + 1 < S && db->h_node_off[sp + 1] <= v0) ++sp;
            uint32_t sl = sp;
            while (sl + 1 < S && db->h_node_off[sl + 1] <= v1) ++sl;
            tsp[t] = make_uint2(sp, sl);
        }
        PTX_TRY(upload(ctx, db->d_emit_tile_sp, tsp.data(), tsp.size()));
    }
    lap("tiles");
    PTX_TRY(trio_visits_build(ctx, db));
    lap("visit table");
    PTX_TRY(trio_runs_build(ctx, db));
    lap("node-block runs");
    PTX_TRY(node_haps_build(ctx, db));
    lap("node -> haplotypes");
    PTX_HIP(ctx, db->d_trio_first.alloc(1));
    PTX_HIP(ctx, db->d_trio_ent.alloc(2));
    PTX_HIP(ctx, db->d_trio_bases.alloc(1));
    PTX_HIP(ctx, db->d_active.alloc(S));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors go out of scope
    return 0;
}

}  // namespace ptx

extern "C" {

void pantax_hip_db_free(pantax_hip_ctx *ctx, pantax_hip_db *db) {
    std::unique_lock<std::recursive_mutex> lk;
    if (ctx) lk = std::unique_lock<std::recursive_mutex>(ctx->mu);
    // the side stream may still be building this db's trio index (a step that failed after the fork)
    bool idle = false;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        idle = hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ctx->stream2) idle = hipStreamSynchronize(ctx->stream2) == hipSuccess && idle;
    }
    // a db is used by its ctx's two streams only (its arrays arrived on the loader's stream, which the loader waited for): behind the two waits its blocks are idle
    if (idle) { DevCacheIdleFrees scope; delete db; }
    else delete db;
}

int pantax_hip_reads_upload(pantax_hip_ctx *ctx, const pantax_hip_packed_reads *r, pantax_hip_reads **out) {
    if (!ctx || !r || !out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    if (r->n_steps >= 0xFFFFFFFFull) return fail(ctx, PANTAX_HIP_E_LIMIT, "reads_upload: %llu steps exceed the 32-bit step offset; split the batch", (unsigned long long)r->n_steps);
    if (r->n_reads && (r->step_off[0] != 0 || r->step_off[r->n_reads] != r->n_steps))
        return fail(ctx, PANTAX_HIP_E_INVALID, "reads_upload: step_off must start at 0 and end at n_steps");
    std::unique_ptr<pantax_hip_reads> rd(new pantax_hip_reads());
    rd->R = r->n_reads;
    rd->T = r->n_steps;
    static const uint32_t zero = 0;
    PTX_TRY(upload(ctx, rd->d_step_off, r->n_reads ? r->step_off : &zero, r->n_reads + 1));
    PTX_TRY(upload(ctx, rd->d_node_id, r->node_id, r->n_steps));
    PTX_TRY(upload(ctx, rd->d_pstart, r->pstart, r->n_reads));
    PTX_TRY(upload(ctx, rd->d_pend, r->pend, r->n_reads));
    PTX_TRY(upload(ctx, rd->d_qlen, r->qlen, r->n_reads));
    PTX_TRY(upload(ctx, rd->d_mapq, r->mapq, r->n_reads));
    rd->has_flags = r->flags != nullptr;
    if (rd->has_flags) PTX_TRY(upload(ctx, rd->d_flags, r->flags, r->n_reads));
    uint32_t max_id = 0;
    for (uint64_t i = 0; i < r->n_steps; ++i) max_id = std::max(max_id, r->node_id[i]);
    PTX_TRY(build_step_read(ctx, rd.get(), max_id));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = rd.release();
    return 0;
}

void pantax_hip_reads_free(pantax_hip_ctx *ctx, pantax_hip_reads *reads) {
    std::unique_lock<std::recursive_mutex> lk;
    if (ctx) lk = std::unique_lock<std::recursive_mutex>(ctx->mu);
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    delete reads;
}

int pantax_hip_bin_reads(pantax_hip_ctx *ctx, const pantax_hip_db *db, pantax_hip_reads *reads, int32_t *species_idx_out,
                         int64_t *read_count_out, int64_t *base_sum_out, int64_t *less_multi_out, int64_t *uniq_count_out) {
    if (!ctx || !db || !reads) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    uint32_t S = db->S;
    DevBuf<unsigned long long> &d_cnt = const_cast<pantax_hip_db *>(db)->d_counters;
    PTX_HIP(ctx, d_cnt.alloc(bin_counter_words(S)));
    PTX_TRY(bin_reads_launch(ctx, db, reads, d_cnt.p));
    // one pinned download: the sums and the head of (species, qlen) for species_profile's first-1000 test
    const size_t res_bytes = bin_result_words(S) * sizeof(unsigned long long);
    PTX_HIP(ctx, ctx->pin_down.reserve(res_bytes));
    PTX_HIP(ctx, hipMemcpyAsync(ctx->pin_down.p, d_cnt.p, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (species_idx_out) { PTX_TRY(species_ensure(ctx, reads)); PTX_TRY(download(ctx, species_idx_out, reads->d_species.p, reads->R)); }
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long *h = reinterpret_cast<const unsigned long long *>(ctx->pin_down.p);
    const uint64_t npre = std::min<uint64_t>(reads->R, BIN_PREFIX);
    const int32_t *pre_sp = reinterpret_cast<const int32_t *>(h + 4ull * S);
    const uint32_t *pre_q = reinterpret_cast<const uint32_t *>(pre_sp + BIN_PREFIX);
    reads->h_pre_species.assign(pre_sp, pre_sp + npre);
    reads->h_pre_qlen.assign(pre_q, pre_q + npre);
    int64_t *outs[4] = {read_count_out, base_sum_out, less_multi_out, uniq_count_out};
    for (int k = 0; k < 4; ++k)
        if (outs[k]) for (uint32_t s = 0; s < S; ++s) outs[k][s] = (int64_t)h[(size_t)k * S + s];
    return 0;
}

int pantax_hip_node_coverage(pantax_hip_ctx *ctx, pantax_hip_db *db, pantax_hip_reads *reads, const uint8_t *species_active,
                             int64_t *bases_per_node_out, uint64_t *node_base_cov_out, int64_t *trio_bases_out,
                             uint64_t *n_abort_out) {
    if (!ctx || !db || !reads) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    if (!reads->binned) return fail(ctx, PANTAX_HIP_E_STATE, "node_coverage: call pantax_hip_bin_reads on these reads first");
    if (trio_bases_out && !db->trio_built) return fail(ctx, PANTAX_HIP_E_STATE, "node_coverage: trio_bases requested but pantax_hip_trio_index has not run");
    const uint8_t *d_active = nullptr;
    if (species_active) {
        PTX_TRY(upload_small(ctx, db->d_active, species_active, db->S));
        d_active = db->d_active.p;
    }
    // trio_bases are handed out in the table's export order, (species, hap, position) -- the order of pantax_hip_trio_get --, not in the order
    // the rows are filed in: the permutation is made before the pass (it may rebuild the index with the window starts: same rows)
    if (trio_bases_out && db->U) PTX_TRY(trio_export_ensure(ctx, db));
    PTX_TRY(coverage_launch(ctx, db, reads, d_active, db->trio_built));
    unsigned long long *d_abort = db->d_abort;
    unsigned long long h_abort = 0;
    std::vector<uint32_t> cov32;
    DevBuf<unsigned long long> tb_export;
    if (bases_per_node_out) PTX_TRY(download(ctx, (unsigned long long *)bases_per_node_out, db->d_bases.p, db->V));
    if (node_base_cov_out) { cov32.resize(db->V); PTX_TRY(download(ctx, cov32.data(), db->d_cov.p, db->V)); }
    if (trio_bases_out && db->U) {
        PTX_HIP(ctx, tb_export.alloc(db->U));
        PTX_TRY(trio_export_u64(ctx, db, db->d_trio_bases.p, tb_export.p));
        PTX_TRY(download(ctx, (unsigned long long *)trio_bases_out, tb_export.p, db->U));
    }
    if (!bases_per_node_out && !node_base_cov_out && !trio_bases_out && !n_abort_out) return 0;   // everything stays on the device: no host sync
    PTX_TRY(download(ctx, &h_abort, d_abort, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (node_base_cov_out) for (uint64_t v = 0; v < db->V; ++v) node_base_cov_out[v] = cov32[v];
    if (n_abort_out) *n_abort_out = h_abort;
    return 0;
}

}  // extern "C"
