// stage_gaf.hip -- a1 on the device: GAF text -> packed reads.  Same contract as the host
// tokenizer host_io.cpp:parse_chunk (load_gaf_file_lazy rcls.rs:119-146, walk regex rcls.rs:242-245), which is the
// checker in the tests:
//   * lines end at '\n' (a trailing '\r' is dropped); empty lines and lines starting with '@' are skipped
//   * up to 12 tab-separated fields; 1 = read_len, 5 = path, 6 = path_len, 7 = path_start, 8 = path_end, 11 = mapq
//   * a numeric field is null when it is empty or holds any non-digit ("*"); values clamp at 2^32-1
//   * null path / path_len / path_start / path_end => flag bit 0 (row dropped by the strain step); a null path has
//     no steps; null mapq = 255, mapq > 255 = 255; null read_len = 0
//   * the walk is every maximal digit run of the path field
//   * id_hash = FNV-1a64 of the read id with a final avalanche (duplicate detection), id_span = its place in the text
//
// Five launches over the text resident in HBM:
//   gaf_nl_count   newlines per 4-KiB tile            -> chained scan -> tile bases
//   gaf_nl_emit    position of every newline           (line i = (nl[i-1], nl[i]))
//   gaf_parse      one thread per raw line: fields, numbers, hash, step count, valid flag
//                  -> chained scans of valid (read index) and of the step counts (step_off)
//   gaf_fill       one thread per raw line: packed columns at the read index, walk -> node_id
// The text is read three times (byte-granular gathers per line: L2-friendly, each line is contiguous).
// Algorithmic bytes: 3 N (text) + 4 T + 30 R (outputs).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#include "common.hpp"
#include "gaf_scan.hpp"
#include "host_io.hpp"
#include "primitives.hpp"
#include "wave.hpp"

namespace ptx {

constexpr int GAF_TILE = 4096;   // bytes per workgroup of the newline kernels (256 threads x 16)

__device__ __forceinline__ uint32_t count_nl16(const uint8_t *__restrict__ txt, uint64_t base, uint64_t N, uint32_t &mask) {
    mask = 0;
    if (base + 16 <= N) {
        const uint4 v = *reinterpret_cast<const uint4 *>(txt + base);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) if (((w[i] >> (8 * b)) & 0xFFu) == '\n') mask |= 1u << (4 * i + b);
    } else {
        for (int b = 0; b < 16; ++b) if (base + b < N && txt[base + b] == '\n') mask |= 1u << b;
    }
    return (uint32_t)__popc(mask);
}

__global__ void __launch_bounds__(256) gaf_nl_count_kernel(const uint8_t *__restrict__ txt, uint64_t N, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * GAF_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t mask;
    uint32_t c = base < N ? count_nl16(txt, base, N, mask) : 0u;
    c = wave_reduce(c, [](uint32_t x, uint32_t y) { return x + y; });
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}
__global__ void __launch_bounds__(256) gaf_nl_emit_kernel(const uint8_t *__restrict__ txt, uint64_t N, const uint32_t *__restrict__ tile_base,
                                                          uint32_t *__restrict__ nl_pos) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * GAF_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t mask = 0;
    const uint32_t c = base < N ? count_nl16(txt, base, N, mask) : 0u;
    uint32_t tot;
    uint32_t off = tile_base[blockIdx.x] + block_excl_scan<256>(c, s_wave, &tot);
    while (mask) {
        const int b = __ffs((int)mask) - 1;
        mask &= mask - 1;
        nl_pos[off++] = (uint32_t)(base + b);
    }
}

struct GafRaw {   // per raw line, before the comment / empty lines are squeezed out
    uint32_t *path_b, *path_e, *ql, *ps, *pe, *steps, *id_off, *id_len;
    uint64_t *id_hash;
    uint8_t *mq, *fl, *valid;
};

__device__ __forceinline__ bool dev_parse_u32(const uint8_t *__restrict__ txt, uint32_t b, uint32_t e, uint32_t &out) {
    if (b == e) return false;
    uint64_t v = 0;
    for (uint32_t p = b; p < e; ++p) {
        const uint8_t ch = txt[p];
        if (ch < '0' || ch > '9') return false;
        v = v * 10 + (uint64_t)(ch - '0');
        if (v > 0xFFFFFFFFull) v = 0xFFFFFFFFull;
    }
    out = (uint32_t)v;
    return true;
}

__global__ void __launch_bounds__(256) gaf_parse_kernel(const uint8_t *__restrict__ txt, uint64_t N, uint32_t n_raw, uint32_t n_nl,
                                                        const uint32_t *__restrict__ nl_pos, GafRaw o) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_raw) return;
    const uint32_t p = i ? nl_pos[i - 1] + 1 : 0u;
    uint32_t le = i < n_nl ? nl_pos[i] : (uint32_t)N;
    if (le > p && txt[le - 1] == '\r') --le;
    const bool valid = le > p && txt[p] != '@';
    o.valid[i] = valid ? 1 : 0;
    if (!valid) { o.steps[i] = 0; return; }
    uint32_t fb[12], fe[12];
    int nf = 0;
    uint32_t q = p;
    while (nf < 12) {
        uint32_t t = q;
        while (t < le && txt[t] != '\t') ++t;
        fb[nf] = q; fe[nf] = t; ++nf;
        if (t >= le) break;
        q = t + 1;
    }
    uint8_t flag = 0;
    uint32_t ql = 0, ps = 0, pe = 0, pl = 0, mq = 255, steps = 0;
    if (nf > 1) dev_parse_u32(txt, fb[1], fe[1], ql);
    const bool path_null = nf <= 5 || (fe[5] - fb[5] == 1 && txt[fb[5]] == '*');
    if (!path_null) {
        bool in_run = false;
        for (uint32_t c = fb[5]; c < fe[5]; ++c) {
            const uint8_t ch = txt[c];
            const bool dig = ch >= '0' && ch <= '9';
            steps += (dig && !in_run) ? 1u : 0u;
            in_run = dig;
        }
    } else flag |= 1;
    if (!(nf > 6 && dev_parse_u32(txt, fb[6], fe[6], pl))) flag |= 1;
    if (!(nf > 7 && dev_parse_u32(txt, fb[7], fe[7], ps))) flag |= 1;
    if (!(nf > 8 && dev_parse_u32(txt, fb[8], fe[8], pe))) flag |= 1;
    if (nf > 11) { uint32_t m; if (dev_parse_u32(txt, fb[11], fe[11], m)) mq = m > 255 ? 255 : m; }
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t c = fb[0]; c < fe[0]; ++c) { h ^= (uint64_t)txt[c]; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    o.path_b[i] = path_null ? 0u : fb[5]; o.path_e[i] = path_null ? 0u : fe[5];
    o.ql[i] = ql; o.ps[i] = ps; o.pe[i] = pe; o.steps[i] = steps;
    o.mq[i] = (uint8_t)mq; o.fl[i] = flag;
    o.id_off[i] = fb[0]; o.id_len[i] = fe[0] - fb[0]; o.id_hash[i] = h;
}

struct GafOut {
    uint32_t *step_off, *node_id, *pstart, *pend, *qlen, *id_off, *id_len;
    uint8_t *mapq, *flags;
    uint64_t *id_hash;
};
__global__ void __launch_bounds__(256) gaf_fill_kernel(const uint8_t *__restrict__ txt, uint32_t n_raw, GafRaw r, const uint32_t *__restrict__ ridx,
                                                       const uint32_t *__restrict__ soff, uint32_t n_reads, uint32_t n_steps, GafOut o) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) o.step_off[n_reads] = n_steps;
    if (i >= n_raw || !r.valid[i]) return;
    const uint32_t k = ridx[i];
    uint32_t w = soff[i];
    o.step_off[k] = w;
    o.pstart[k] = r.ps[i]; o.pend[k] = r.pe[i]; o.qlen[k] = r.ql[i]; o.mapq[k] = r.mq[i]; o.flags[k] = r.fl[i];
    o.id_off[k] = r.id_off[i]; o.id_len[k] = r.id_len[i]; o.id_hash[k] = r.id_hash[i];
    const uint32_t pb = r.path_b[i], pe = r.path_e[i];
    bool in_run = false;
    uint64_t v = 0;
    for (uint32_t c = pb; c < pe; ++c) {
        const uint8_t ch = txt[c];
        const bool dig = ch >= '0' && ch <= '9';
        if (dig) { v = in_run ? v * 10 + (uint64_t)(ch - '0') : (uint64_t)(ch - '0'); if (v > 0xFFFFFFFFull) v = 0xFFFFFFFFull; }
        else if (in_run) o.node_id[w++] = (uint32_t)v;
        in_run = dig;
    }
    if (in_run) o.node_id[w++] = (uint32_t)v;
}

int gaf_upload_and_scan(Ctx *ctx, const char *text, uint64_t size, DevBuf<uint8_t> &d_txt, DevBuf<uint32_t> &nl_pos, uint32_t *n_nl_out, int fd, uint64_t file_off) {
    PTX_HIP(ctx, d_txt.alloc(size + 16));
    if (fd >= 0) PTX_TRY(upload_file(ctx, d_txt.p, fd, file_off, size));
    else PTX_TRY(upload_big(ctx, d_txt.p, text, size));
    const uint32_t n_tiles = (uint32_t)((size + GAF_TILE - 1) / GAF_TILE);
    DevBuf<uint32_t> tile_cnt, tile_base, tot, scan_tmp;
    PTX_HIP(ctx, tile_cnt.alloc(n_tiles)); PTX_HIP(ctx, tile_base.alloc(n_tiles)); PTX_HIP(ctx, tot.alloc(4)); PTX_HIP(ctx, scan_tmp.alloc(16));
    {
        KTimer t(ctx, "gaf_nl_count_kernel");
        hipLaunchKernelGGL(gaf_nl_count_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, d_txt.p, size, tile_cnt.p);
    }
    PTX_TRY(exclusive_scan_u32(ctx, tile_cnt.p, tile_base.p, n_tiles, scan_tmp.p, tot.p));
    uint32_t n_nl = 0;
    PTX_TRY(download(ctx, &n_nl, tot.p, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PTX_HIP(ctx, nl_pos.alloc(n_nl ? n_nl : 1));
    {
        KTimer t(ctx, "gaf_nl_emit_kernel");
        hipLaunchKernelGGL(gaf_nl_emit_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, d_txt.p, size, tile_base.p, nl_pos.p);
    }
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the tile tables are released on return
    *n_nl_out = n_nl;
    return 0;
}

// after sorting the id hashes: how many adjacent pairs are equal (0 = every read id is distinct, the usual case, and
// the host can skip its hash-set pass of the duplicate-id rule, profile.rs:361-437)
__global__ void __launch_bounds__(256) dup_count_kernel(uint64_t n, const uint64_t *__restrict__ sorted, uint32_t *__restrict__ out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x + 1; i < n; i += (uint64_t)gridDim.x * 256) c += sorted[i] == sorted[i - 1] ? 1u : 0u;
    c = wave_reduce(c, [](uint32_t x, uint32_t y) { return x + y; });
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

__global__ void __launch_bounds__(256) max_u32_kernel(uint64_t n, const uint32_t *__restrict__ v, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) m = max(m, v[i]);
    m = wave_reduce(m, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// one tokenised piece of the text (< 4 GiB: positions inside a piece are 32-bit), still on the device
struct GafPiece {
    uint64_t R = 0, T = 0;
    DevBuf<uint32_t> o32[7];   // step_off (piece-local), node_id, pstart, pend, qlen, id_off (piece-local), id_len
    DevBuf<uint8_t> o8[2];     // mapq, flags
    DevBuf<uint64_t> o_hash;
};

static int tokenize_piece(Ctx *ctx, const char *text, uint64_t size, int fd, uint64_t file_off, GafPiece &pc) {
    DevBuf<uint8_t> d_txt;
    DevBuf<uint32_t> nl_pos, tot, scan_tmp;
    uint32_t n_nl = 0;
    const bool trace = std::getenv("PANTAX_HIP_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gaf_tokenize]   piece: %-25s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    PTX_TRY(gaf_upload_and_scan(ctx, text, size, d_txt, nl_pos, &n_nl, fd, file_off));
    lap("text upload + newline scan");
    PTX_HIP(ctx, tot.alloc(4)); PTX_HIP(ctx, scan_tmp.alloc(16));
    const uint32_t n_raw = n_nl + (text[size - 1] != '\n' ? 1u : 0u);
    DevBuf<uint32_t> r32[8], ridx, soff;
    DevBuf<uint64_t> r_hash;
    DevBuf<uint8_t> r8[3];
    for (auto &b : r32) PTX_HIP(ctx, b.alloc(n_raw ? n_raw : 1));
    for (auto &b : r8) PTX_HIP(ctx, b.alloc(n_raw ? n_raw : 1));
    PTX_HIP(ctx, r_hash.alloc(n_raw ? n_raw : 1)); PTX_HIP(ctx, ridx.alloc(n_raw ? n_raw : 1)); PTX_HIP(ctx, soff.alloc(n_raw ? n_raw : 1));
    GafRaw raw{r32[0].p, r32[1].p, r32[2].p, r32[3].p, r32[4].p, r32[5].p, r32[6].p, r32[7].p, r_hash.p, r8[0].p, r8[1].p, r8[2].p};
    const uint32_t grid = (n_raw + 255) / 256 ? (n_raw + 255) / 256 : 1;
    {
        KTimer t(ctx, "gaf_parse_kernel");
        hipLaunchKernelGGL(gaf_parse_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_txt.p, size, n_raw, n_nl, nl_pos.p, raw);
    }
    PTX_TRY(exclusive_scan_u8(ctx, raw.valid, ridx.p, n_raw, scan_tmp.p, tot.p + 1));
    PTX_TRY(exclusive_scan_u32(ctx, raw.steps, soff.p, n_raw, scan_tmp.p, tot.p + 2));
    uint32_t rt[2] = {0, 0};
    PTX_TRY(download(ctx, rt, tot.p + 1, 2));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t R = rt[0], T = rt[1];
    pc.R = R; pc.T = T;
    PTX_HIP(ctx, pc.o32[0].alloc(R + 1)); PTX_HIP(ctx, pc.o32[1].alloc(T ? T : 1));
    for (int k = 2; k < 7; ++k) PTX_HIP(ctx, pc.o32[k].alloc(R ? R : 1));
    for (auto &b : pc.o8) PTX_HIP(ctx, b.alloc(R ? R : 1));
    PTX_HIP(ctx, pc.o_hash.alloc(R ? R : 1));
    GafOut go{pc.o32[0].p, pc.o32[1].p, pc.o32[2].p, pc.o32[3].p, pc.o32[4].p, pc.o32[5].p, pc.o32[6].p, pc.o8[0].p, pc.o8[1].p, pc.o_hash.p};
    {
        KTimer t(ctx, "gaf_fill_kernel");
        hipLaunchKernelGGL(gaf_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_txt.p, n_raw, raw, ridx.p, soff.p, (uint32_t)R, (uint32_t)T, go);
    }
    PTX_HIP(ctx, hipGetLastError());
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the piece's text and raw columns are released on return
    lap("parse + scans + fill");
    return 0;
}

__global__ void __launch_bounds__(256) add_u32_offset_kernel(uint64_t n, uint32_t *__restrict__ v, uint32_t add) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) v[i] += add;
}

// text (host, `size` bytes) -> HostReads, tokenised on the device.  With `resident` the packed reads stay in HBM
// (the object is ready for pantax_hip_bin_reads) and only the columns host code needs come back: read_len, mapq,
// flags, id hashes and id spans -- the walks (node_id, step_off, path_start, path_end) are not downloaded.
// Texts of 4 GiB and more are cut at line ends into pieces (32-bit positions inside a piece) whose packed columns are
// joined on the device; PANTAX_GAF_PIECE_BYTES lowers the piece size (tests).
// `file_base`: where `text` starts in the file behind `fd` (a rank's byte range of a shared GAF); `group` = false leaves
// the resident reads as plain columns (no locus-grouped copy: a slice that is binned and then routed away, stage_route.hip).
// `want_id_spans`: where every read id sits in the text (the binning report writes the ids back); 160 MB of host memory and
// ~20 ms per 10 M reads that nobody else needs.
int gaf_tokenize_device(Ctx *ctx, const char *text, uint64_t size, HostReads &out, Reads *resident, int fd, uint64_t file_base, bool group,
                        bool want_id_spans) {
    out = HostReads();
    if (resident) resident->grouped = group;
    // PANTAX_HIP_TRACE=1: where the load spends its time (stderr)
    const bool trace = std::getenv("PANTAX_HIP_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gaf_tokenize] %-34s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (size == 0) {
        if (resident) {
            static const uint32_t zero = 0;
            resident->R = resident->T = 0;
            PTX_TRY(upload(ctx, resident->d_step_off, &zero, 1));
            if (group) PTX_TRY(build_step_read(ctx, resident, 0));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        return 0;
    }
    uint64_t piece_max = 0xE0000000ull;   // 3.5 GiB
    if (const char *ev = std::getenv("PANTAX_GAF_PIECE_BYTES")) { const long long v = std::atoll(ev); if (v > 0 && (uint64_t)v < piece_max) piece_max = (uint64_t)v; }
    std::vector<std::unique_ptr<GafPiece>> pcs;
    std::vector<uint64_t> piece_off;
    for (uint64_t off = 0; off < size;) {
        uint64_t end = std::min<uint64_t>(size, off + piece_max);
        if (end < size) {   // back to the last line end inside the piece
            const void *nl = memrchr(text + off, '\n', (size_t)(end - off));
            if (!nl) return fail(ctx, PANTAX_HIP_E_LIMIT, "gaf_tokenize: a line of more than %llu bytes at offset %llu", (unsigned long long)piece_max, (unsigned long long)off);
            end = (uint64_t)(static_cast<const char *>(nl) - text) + 1;
        }
        pcs.emplace_back(new GafPiece());
        piece_off.push_back(off);
        PTX_TRY(tokenize_piece(ctx, text + off, end - off, fd, file_base + off, *pcs.back()));
        off = end;
    }
    lap("pieces: upload + scan + parse + fill");
    uint64_t R = 0, T = 0;
    for (auto &pc : pcs) { R += pc->R; T += pc->T; }
    if (R >= 0xFFFFFFFFull || T >= 0xFFFFFFFFull)
        return fail(ctx, PANTAX_HIP_E_LIMIT, "gaf_tokenize: %llu reads / %llu walk steps exceed the 32-bit offsets of one batch; split the input", (unsigned long long)R, (unsigned long long)T);
    // id spans: piece-local 32-bit positions -> positions in the whole text
    if (want_id_spans) {
        out.id_span.resize(R);
        uint64_t r0 = 0;
        std::vector<uint32_t> id_off, id_len;
        for (size_t k = 0; k < pcs.size(); ++k) {
            GafPiece &pc = *pcs[k];
            id_off.resize(pc.R); id_len.resize(pc.R);
            PTX_TRY(download(ctx, id_off.data(), pc.o32[5].p, pc.R)); PTX_TRY(download(ctx, id_len.data(), pc.o32[6].p, pc.R));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const uint64_t base = piece_off[k], rr = r0;
            parallel_for(pc.R, 8, [&](uint64_t i0, uint64_t i1) { for (uint64_t i = i0; i < i1; ++i) out.id_span[rr + i] = {base + (uint64_t)id_off[i], id_len[i]}; });
            r0 += pc.R;
        }
    }
    // join the pieces (a single piece is used as it is)
    DevBuf<uint32_t> o32[5];
    DevBuf<uint8_t> o8[2];
    DevBuf<uint64_t> o_hash;
    if (pcs.size() == 1) {
        for (int k = 0; k < 5; ++k) o32[k].take(pcs[0]->o32[k]);
        o8[0].take(pcs[0]->o8[0]); o8[1].take(pcs[0]->o8[1]); o_hash.take(pcs[0]->o_hash);
    } else {
        PTX_HIP(ctx, o32[0].alloc(R + 1)); PTX_HIP(ctx, o32[1].alloc(T ? T : 1));
        for (int k = 2; k < 5; ++k) PTX_HIP(ctx, o32[k].alloc(R ? R : 1));
        for (auto &b : o8) PTX_HIP(ctx, b.alloc(R ? R : 1));
        PTX_HIP(ctx, o_hash.alloc(R ? R : 1));
        uint64_t r0 = 0, t0 = 0;
        for (auto &pcp : pcs) {
            GafPiece &pc = *pcp;
            auto d2d = [&](void *dst, const void *src, uint64_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream) : hipSuccess; };
            PTX_HIP(ctx, d2d(o32[0].p + r0, pc.o32[0].p, pc.R * 4)); PTX_HIP(ctx, d2d(o32[1].p + t0, pc.o32[1].p, pc.T * 4));
            for (int k = 2; k < 5; ++k) PTX_HIP(ctx, d2d(o32[k].p + r0, pc.o32[k].p, pc.R * 4));
            PTX_HIP(ctx, d2d(o8[0].p + r0, pc.o8[0].p, pc.R)); PTX_HIP(ctx, d2d(o8[1].p + r0, pc.o8[1].p, pc.R));
            PTX_HIP(ctx, d2d(o_hash.p + r0, pc.o_hash.p, pc.R * 8));
            if (pc.R && t0) hipLaunchKernelGGL(add_u32_offset_kernel, dim3(grid_for(pc.R, 256, ctx->n_cu * 4)), dim3(256), 0, ctx->stream, pc.R, o32[0].p + r0, (uint32_t)t0);
            r0 += pc.R; t0 += pc.T;
        }
        const uint32_t t32 = (uint32_t)T;
        PTX_HIP(ctx, hipMemcpyAsync(o32[0].p + R, &t32, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        pcs.clear();
    }
    DevBuf<uint32_t> tot, scan_tmp;
    PTX_HIP(ctx, tot.alloc(4)); PTX_HIP(ctx, scan_tmp.alloc(16));
    out.qlen.resize(R); out.mapq.resize(R); out.flags.resize(R); out.id_hash.resize(R);
    if (!resident) {
        out.step_off.resize(R + 1); out.node_id.resize(T); out.pstart.resize(R); out.pend.resize(R);
        PTX_TRY(download(ctx, out.step_off.data(), o32[0].p, R + 1));
        PTX_TRY(download(ctx, out.node_id.data(), o32[1].p, T));
        PTX_TRY(download(ctx, out.pstart.data(), o32[2].p, R)); PTX_TRY(download(ctx, out.pend.data(), o32[3].p, R));
    } else {
        PTX_HIP(ctx, hipMemsetAsync(tot.p + 3, 0, sizeof(uint32_t), ctx->stream));
        if (T) hipLaunchKernelGGL(max_u32_kernel, dim3(grid_for(T, 256, ctx->n_cu * 4)), dim3(256), 0, ctx->stream, T, o32[1].p, tot.p + 3);
    }
    lap("join pieces / id spans");
    // are all read ids distinct?  sort a copy of the hashes, count equal neighbours
    DevBuf<uint64_t> hs_a, hs_b;
    DevBuf<uint32_t> hs_table, dup_cnt;
    uint32_t n_dup = 0;
    if (R > 1) {
        PTX_HIP(ctx, hs_a.alloc(R)); PTX_HIP(ctx, hs_b.alloc(R)); PTX_HIP(ctx, hs_table.alloc(sort_table_elems(R))); PTX_HIP(ctx, dup_cnt.alloc(1));
        PTX_HIP(ctx, hipMemcpyAsync(hs_a.p, o_hash.p, R * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        PTX_HIP(ctx, hipMemsetAsync(dup_cnt.p, 0, sizeof(uint32_t), ctx->stream));
        SortBufs A, B;
        A.nw = B.nw = 1; A.k[0] = hs_a.p; B.k[0] = hs_b.p;
        std::vector<SortPass> passes;
        add_passes(passes, 0, 0, 64);
        bool in_b = false;
        PTX_TRY(radix_sort(ctx, A, B, R, passes.data(), (int)passes.size(), hs_table.p, scan_tmp.p, &in_b, nullptr));
        hipLaunchKernelGGL(dup_count_kernel, dim3(grid_for(R, 256, ctx->n_cu * 4)), dim3(256), 0, ctx->stream, R, in_b ? hs_b.p : hs_a.p, dup_cnt.p);
        PTX_TRY(download(ctx, &n_dup, dup_cnt.p, 1));
    }
    lap("id-hash sort");
    uint32_t max_id = 0;
    PTX_TRY(download(ctx, out.qlen.data(), o32[4].p, R));
    PTX_TRY(download(ctx, out.mapq.data(), o8[0].p, R)); PTX_TRY(download(ctx, out.flags.data(), o8[1].p, R));
    PTX_TRY(download(ctx, out.id_hash.data(), o_hash.p, R));
    if (resident) PTX_TRY(download(ctx, &max_id, tot.p + 3, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lap("host columns download");
    if (resident) {
        resident->R = R; resident->T = T;
        resident->d_step_off.take(o32[0]); resident->d_node_id.take(o32[1]); resident->d_pstart.take(o32[2]); resident->d_pend.take(o32[3]);
        resident->d_qlen.take(o32[4]); resident->d_mapq.take(o8[0]); resident->d_flags.take(o8[1]);
        resident->has_flags = true;
        if (group) PTX_TRY(build_step_read(ctx, resident, max_id));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        lap("locus-grouped copy");
    }
    out.n_lines = R;
    out.ids_distinct = n_dup == 0 ? 1 : 0;
    return 0;
}

}  // namespace ptx

using namespace ptx;
extern "C" int pantax_hip_gaf_load_device(pantax_hip_ctx *ctx, const char *path, pantax_hip_gaf **out) {
    if (!ctx || !path || !out) return PANTAX_HIP_E_INVALID;
    *out = nullptr;
    PTX_ENTER(ctx);
    pantax_hip_gaf *g = new pantax_hip_gaf();
    std::string e = g->mf.open(path);
    if (!e.empty()) { delete g; return fail(ctx, PANTAX_HIP_E_IO, "%s", e.c_str()); }
    const int rc = gaf_tokenize_device(ctx, g->mf.data, g->mf.size, g->reads, nullptr, g->mf.fd);
    if (rc != 0) { delete g; return rc; }
    *out = g;
    return 0;
}

extern "C" int pantax_hip_reads_load_gaf(pantax_hip_ctx *ctx, const char *path, pantax_hip_reads **reads_out, pantax_hip_gaf **gaf_out) {
    if (!ctx || !path || !reads_out) return PANTAX_HIP_E_INVALID;
    *reads_out = nullptr;
    if (gaf_out) *gaf_out = nullptr;
    PTX_ENTER(ctx);
    std::unique_ptr<pantax_hip_gaf> g(new pantax_hip_gaf());
    std::unique_ptr<pantax_hip_reads> rd(new pantax_hip_reads());
    std::string e = g->mf.open(path);
    if (!e.empty()) return fail(ctx, PANTAX_HIP_E_IO, "%s", e.c_str());
    PTX_TRY(gaf_tokenize_device(ctx, g->mf.data, g->mf.size, g->reads, rd.get(), g->mf.fd));
    *reads_out = rd.release();
    if (gaf_out) *gaf_out = g.release();
    return 0;
}

extern "C" int pantax_hip_reads_set_flags(pantax_hip_ctx *ctx, pantax_hip_reads *reads, const uint8_t *flags) {
    if (!ctx || !reads) return PANTAX_HIP_E_INVALID;
    PTX_ENTER(ctx);
    reads->has_flags = flags != nullptr;
    if (flags) { PTX_TRY(upload(ctx, reads->d_flags, flags, reads->R)); PTX_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
    reads->g_flags_valid = false;
    reads->binned = false;   // the per-slot species carry the drop flags: bin again
    return 0;
}
