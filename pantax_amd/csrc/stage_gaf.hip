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
// The text is read three times (per line, through an 8-byte register window -- TxtWin: each line is contiguous).
// Algorithmic bytes: 3 N (text) + 4 T + 30 R (outputs).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
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

__device__ __forceinline__ bool dev_parse_u32(TxtWin &tw, uint32_t b, uint32_t e, uint32_t &out) {
    if (b == e) return false;
    uint64_t v = 0;
    for (uint32_t p = b; p < e; ++p) {
        const uint32_t ch = tw.at(p);
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
    TxtWin tw(txt);
    if (le > p && tw.at(le - 1) == '\r') --le;
    const bool valid = le > p && tw.at(p) != '@';
    o.valid[i] = valid ? 1 : 0;
    if (!valid) { o.steps[i] = 0; return; }
    // the first twelve fields (static indices: the arrays stay in registers)
    uint32_t fb[12], fe[12];
    int nf = 0;
    {
        uint32_t q = p;
        bool more = true;
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            fb[f] = fe[f] = le;
            if (more) {
                const uint32_t t = tw.find(q, le, '\t');
                fb[f] = q; fe[f] = t; nf = f + 1;
                more = t < le;
                q = t + 1;
            }
        }
    }
    uint8_t flag = 0;
    uint32_t ql = 0, ps = 0, pe = 0, pl = 0, mq = 255, steps = 0;
    if (nf > 1) dev_parse_u32(tw, fb[1], fe[1], ql);
    const bool path_null = nf <= 5 || fe[5] == fb[5] || (fe[5] - fb[5] == 1 && tw.at(fb[5]) == '*');   // '*' and the empty field are null
    if (!path_null) {
        bool in_run = false;
        for (uint32_t c = fb[5]; c < fe[5]; ++c) {
            const uint32_t ch = tw.at(c);
            const bool dig = ch >= '0' && ch <= '9';
            steps += (dig && !in_run) ? 1u : 0u;
            in_run = dig;
        }
    } else flag |= 1;
    if (!(nf > 6 && dev_parse_u32(tw, fb[6], fe[6], pl))) flag |= 1;
    if (!(nf > 7 && dev_parse_u32(tw, fb[7], fe[7], ps))) flag |= 1;
    if (!(nf > 8 && dev_parse_u32(tw, fb[8], fe[8], pe))) flag |= 1;
    if (nf > 11) { uint32_t m; if (dev_parse_u32(tw, fb[11], fe[11], m)) mq = m > 255 ? 255 : m; }
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t c = fb[0]; c < fe[0]; ++c) { h ^= (uint64_t)tw.at(c); h *= 0x100000001b3ull; }
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
// `o` points at this piece's place in the joined columns; t_base = walk steps of the pieces before it (step offsets are global)
__global__ void __launch_bounds__(256) gaf_fill_kernel(const uint8_t *__restrict__ txt, uint32_t n_raw, GafRaw r, const uint32_t *__restrict__ ridx,
                                                       const uint32_t *__restrict__ soff, uint32_t n_reads, uint32_t n_steps, uint32_t t_base, GafOut o) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) o.step_off[n_reads] = t_base + n_steps;
    if (i >= n_raw || !r.valid[i]) return;
    const uint32_t k = ridx[i];
    uint32_t w = soff[i];
    o.step_off[k] = t_base + w;
    o.pstart[k] = r.ps[i]; o.pend[k] = r.pe[i]; o.qlen[k] = r.ql[i]; o.mapq[k] = r.mq[i]; o.flags[k] = r.fl[i];
    o.id_off[k] = r.id_off[i]; o.id_len[k] = r.id_len[i]; o.id_hash[k] = r.id_hash[i];
    const uint32_t pb = r.path_b[i], pe = r.path_e[i];
    bool in_run = false;
    uint64_t v = 0;
    TxtWin tw(txt);
    for (uint32_t c = pb; c < pe; ++c) {
        const uint32_t ch = tw.at(c);
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
    return gaf_scan_newlines(ctx, size, d_txt, nl_pos, n_nl_out);
}

// newline positions of a text that is already in HBM
int gaf_scan_newlines(Ctx *ctx, uint64_t size, DevBuf<uint8_t> &d_txt, DevBuf<uint32_t> &nl_pos, uint32_t *n_nl_out) {
    DevBuf<uint32_t> tile_cnt, tile_base, tot, scan_tmp;
    PTX_TRY(gaf_scan_newlines_ws(ctx, size, d_txt.p, nl_pos, n_nl_out, tile_cnt, tile_base, tot, scan_tmp));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the tile tables are released on return
    return 0;
}
// ... with the caller's (grow-only) work buffers: nothing is released here, so no wait at the end
int gaf_scan_newlines_ws(Ctx *ctx, uint64_t size, const uint8_t *d_txt, DevBuf<uint32_t> &nl_pos, uint32_t *n_nl_out, DevBuf<uint32_t> &tile_cnt,
                         DevBuf<uint32_t> &tile_base, DevBuf<uint32_t> &tot, DevBuf<uint32_t> &scan_tmp) {
    const uint32_t n_tiles = (uint32_t)((size + GAF_TILE - 1) / GAF_TILE);
    PTX_HIP(ctx, tile_cnt.alloc(n_tiles)); PTX_HIP(ctx, tile_base.alloc(n_tiles)); PTX_HIP(ctx, tot.alloc(4)); PTX_HIP(ctx, scan_tmp.alloc(16));
    {
        KTimer t(ctx, "gaf_nl_count_kernel");
        hipLaunchKernelGGL(gaf_nl_count_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, d_txt, size, tile_cnt.p);
    }
    PTX_TRY(exclusive_scan_u32(ctx, tile_cnt.p, tile_base.p, n_tiles, scan_tmp.p, tot.p));
    uint32_t n_nl = 0;
    PTX_TRY(download(ctx, &n_nl, tot.p, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PTX_HIP(ctx, nl_pos.alloc(n_nl ? n_nl : 1));
    {
        KTimer t(ctx, "gaf_nl_emit_kernel");
        hipLaunchKernelGGL(gaf_nl_emit_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, d_txt, size, tile_base.p, nl_pos.p);
    }
    PTX_HIP(ctx, hipGetLastError());
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

__global__ void __launch_bounds__(256) id_set_insert_kernel(uint32_t n, const uint64_t *__restrict__ hashes, unsigned long long *__restrict__ tab, uint64_t mask,
                                                           uint32_t *__restrict__ dup) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long h = hashes[i];
    if (!h) h = 1ull;                                  // 0 marks an empty slot (a false "equal" of the hashes 0 and 1 only sends the caller to its exact pass)
    uint64_t slot = (h >> 5) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long old = atomicCAS(&tab[slot], 0ull, h);
        if (old == 0ull) return;
        if (old == h) { atomicAdd(dup, 1u); return; }
        slot = (slot + 1) & mask;
    }
    atomicAdd(dup, 1u);                                // a full table cannot happen (load <= 0.7): never silent
}

__global__ void __launch_bounds__(256) max_u32_kernel(uint64_t n, const uint32_t *__restrict__ v, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) m = max(m, v[i]);
    m = wave_reduce(m, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// The joined columns of all pieces (grow-only; sized from the first piece's density, so later pieces are written in place),
// and the work buffers one piece after the other reuses -- no allocation is made or released between two pieces: the device
// allocation cache hands a block out again only behind a device-wide wait, which would stall the upload that runs beside.
struct GafJoined {
    DevBuf<uint32_t> o32[7];   // step_off, node_id, pstart, pend, qlen, id_off (piece-local), id_len
    DevBuf<uint8_t> o8[2];     // mapq, flags
    DevBuf<uint64_t> o_hash;
    // "are all read ids distinct?" piece by piece, beside the upload: every piece's id hashes are inserted into an open-addressing set
    // (sized from the first piece's density); an insert that meets its own value counts a duplicate.  A text that turns out denser than
    // the set was sized for falls back to sorting the hashes after the last piece (round 3's path: 9 ms behind the last byte at 1e8 reads).
    DevBuf<uint64_t> id_set;
    DevBuf<uint32_t> dup_cnt;
    uint64_t set_slots = 0;
    bool set_ok = false;
    uint64_t R = 0, T = 0, cap_r = 0, cap_t = 0;
    std::vector<uint64_t> piece_r0;   // first read of every piece (id spans are piece-local)
};
struct GafWork {
    DevBuf<uint32_t> nl_pos, tile_cnt, tile_base, tot, scan_tmp, r32[8], ridx, soff;
    DevBuf<uint64_t> r_hash;
    DevBuf<uint8_t> r8[3];
};

// capacity for R reads / T steps in all (the present contents are kept)
static int joined_reserve(Ctx *ctx, GafJoined &J, uint64_t need_r, uint64_t need_t) {
    auto grow32 = [&](DevBuf<uint32_t> &b, uint64_t have, uint64_t want) -> int {
        DevBuf<uint32_t> nb;
        PTX_HIP(ctx, nb.alloc(want));
        if (have) PTX_HIP(ctx, hipMemcpyAsync(nb.p, b.p, have * 4, hipMemcpyDeviceToDevice, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        b.take(nb);
        return 0;
    };
    if (need_t > J.cap_t || !J.o32[1].p) {
        const uint64_t want = std::max<uint64_t>(need_t, J.cap_t + J.cap_t / 2);
        PTX_TRY(grow32(J.o32[1], J.T, want ? want : 1));
        J.cap_t = want;
    }
    if (need_r > J.cap_r || !J.o32[0].p) {
        const uint64_t want = std::max<uint64_t>(need_r, J.cap_r + J.cap_r / 2);
        PTX_TRY(grow32(J.o32[0], J.R ? J.R + 1 : 0, want + 1));
        for (int k = 2; k < 7; ++k) PTX_TRY(grow32(J.o32[k], J.R, want ? want : 1));
        for (auto &b : J.o8) {
            DevBuf<uint8_t> nb;
            PTX_HIP(ctx, nb.alloc(want ? want : 1));
            if (J.R) PTX_HIP(ctx, hipMemcpyAsync(nb.p, b.p, J.R, hipMemcpyDeviceToDevice, ctx->stream));
            PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            b.take(nb);
        }
        DevBuf<uint64_t> nh;
        PTX_HIP(ctx, nh.alloc(want ? want : 1));
        if (J.R) PTX_HIP(ctx, hipMemcpyAsync(nh.p, J.o_hash.p, J.R * 8, hipMemcpyDeviceToDevice, ctx->stream));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        J.o_hash.take(nh);
        J.cap_r = want;
    }
    return 0;
}

// d_txt: the piece's text, uploaded (size + 16 bytes allocated); last_is_nl: its last byte is a line end; rest_bytes: text that
// follows this piece (sizes the joined columns from this piece's density when they are first allocated)
static int tokenize_piece(Ctx *ctx, const uint8_t *d_txt, uint64_t size, bool last_is_nl, uint64_t rest_bytes, GafWork &W, GafJoined &J) {
    uint32_t n_nl = 0;
    const bool trace = ctx->cfg.trace;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gaf_tokenize]   piece: %-25s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    PTX_TRY(gaf_scan_newlines_ws(ctx, size, d_txt, W.nl_pos, &n_nl, W.tile_cnt, W.tile_base, W.tot, W.scan_tmp));
    lap("newline scan");
    const uint32_t n_raw = n_nl + (last_is_nl ? 0u : 1u);
    for (auto &b : W.r32) PTX_HIP(ctx, b.alloc(n_raw ? n_raw : 1));
    for (auto &b : W.r8) PTX_HIP(ctx, b.alloc(n_raw ? n_raw : 1));
    PTX_HIP(ctx, W.r_hash.alloc(n_raw ? n_raw : 1)); PTX_HIP(ctx, W.ridx.alloc(n_raw ? n_raw : 1)); PTX_HIP(ctx, W.soff.alloc(n_raw ? n_raw : 1));
    GafRaw raw{W.r32[0].p, W.r32[1].p, W.r32[2].p, W.r32[3].p, W.r32[4].p, W.r32[5].p, W.r32[6].p, W.r32[7].p, W.r_hash.p, W.r8[0].p, W.r8[1].p, W.r8[2].p};
    const uint32_t grid = (n_raw + 255) / 256 ? (n_raw + 255) / 256 : 1;
    {
        KTimer t(ctx, "gaf_parse_kernel");
        hipLaunchKernelGGL(gaf_parse_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_txt, size, n_raw, n_nl, W.nl_pos.p, raw);
    }
    PTX_TRY(exclusive_scan_u8(ctx, raw.valid, W.ridx.p, n_raw, W.scan_tmp.p, W.tot.p + 1));
    PTX_TRY(exclusive_scan_u32(ctx, raw.steps, W.soff.p, n_raw, W.scan_tmp.p, W.tot.p + 2));
    uint32_t rt[2] = {0, 0};
    PTX_TRY(download(ctx, rt, W.tot.p + 1, 2));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t R = rt[0], T = rt[1];
    if (J.R + R >= 0xFFFFFFFFull || J.T + T >= 0xFFFFFFFFull)
        return fail(ctx, PANTAX_HIP_E_LIMIT, "gaf_tokenize: %llu reads / %llu walk steps exceed the 32-bit offsets of one batch; split the input",
                    (unsigned long long)(J.R + R), (unsigned long long)(J.T + T));
    // room in the joined columns: first piece -> the whole text at this piece's density + 3 %; later pieces fit unless they are denser
    uint64_t need_r = J.R + R, need_t = J.T + T;
    if (need_r > J.cap_r || need_t > J.cap_t || !J.o32[0].p) {
        const double f = 1.03 * (double)(size + rest_bytes) / (double)(size ? size : 1);
        need_r = std::max<uint64_t>(need_r, std::min<uint64_t>(0xFFFFFFFEull, J.R + (uint64_t)((double)R * f) + 1024));
        need_t = std::max<uint64_t>(need_t, std::min<uint64_t>(0xFFFFFFFEull, J.T + (uint64_t)((double)T * f) + 1024));
        PTX_TRY(joined_reserve(ctx, J, need_r, need_t));
    }
    if (!J.id_set.p) {   // first piece: the set for the whole text, twice the reads the joined columns were sized for
        uint64_t slots = 1024;
        while (slots < 2 * J.cap_r) slots <<= 1;
        PTX_HIP(ctx, J.id_set.alloc(slots)); PTX_HIP(ctx, J.dup_cnt.alloc(1));
        PTX_TRY(zero_fill(ctx, J.id_set.p, slots * sizeof(uint64_t)));
        PTX_HIP(ctx, hipMemsetAsync(J.dup_cnt.p, 0, sizeof(uint32_t), ctx->stream));
        J.set_slots = slots; J.set_ok = true;
    }
    if ((double)(J.R + R) > 0.7 * (double)J.set_slots) J.set_ok = false;
    GafOut go{J.o32[0].p + J.R, J.o32[1].p + J.T, J.o32[2].p + J.R, J.o32[3].p + J.R, J.o32[4].p + J.R, J.o32[5].p + J.R, J.o32[6].p + J.R,
              J.o8[0].p + J.R, J.o8[1].p + J.R, J.o_hash.p + J.R};
    {
        KTimer t(ctx, "gaf_fill_kernel");
        hipLaunchKernelGGL(gaf_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_txt, n_raw, raw, W.ridx.p, W.soff.p, (uint32_t)R, (uint32_t)T, (uint32_t)J.T, go);
    }
    if (J.set_ok && R)
        hipLaunchKernelGGL(id_set_insert_kernel, dim3((uint32_t)((R + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)R, J.o_hash.p + J.R,
                           reinterpret_cast<unsigned long long *>(J.id_set.p), J.set_slots - 1, J.dup_cnt.p);
    PTX_HIP(ctx, hipGetLastError());
    J.piece_r0.push_back(J.R);
    J.R += R; J.T += T;
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
                        bool want_id_spans, bool want_host_columns) {
    out = HostReads();
    if (resident) resident->grouped = group;
    // PANTAX_HIP_TRACE=1: where the load spends its time (stderr)
    const bool trace = ctx->cfg.trace;
    const auto t_enter = std::chrono::steady_clock::now();
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
    // Pieces of the text (cut at line ends; positions inside a piece are 32-bit) travel on an upload stream of their own, fed by
    // an uploader thread through the pinned ring, while this thread tokenises the piece before: PCIe and the tokenizer kernels
    // work side by side.  Three text buffers rotate; the pieces' columns are written straight into the joined arrays.
    // Piece size: a sixth of the text, 64 MB .. 1 GiB (PANTAX_GAF_PIECE_BYTES caps it; a line longer than that cap is refused).
    uint64_t piece_max = std::min<uint64_t>(1ull << 30, std::max<uint64_t>(64ull << 20, size / 6));
    bool capped = false;
    if (ctx->cfg.gaf_piece_bytes && ctx->cfg.gaf_piece_bytes < 0xE0000000ull) { piece_max = ctx->cfg.gaf_piece_bytes; capped = true; }
    std::vector<uint64_t> piece_off, piece_end;
    uint64_t longest = 0;
    for (uint64_t off = 0; off < size;) {
        // the end of the text travels in pieces that shrink geometrically (a third of what is left, down to 64 MB): the tokenizer's work on the pieces
        // that are still unparsed when the LAST byte arrives is what the load waits for (14 ms per GiB against 18.5 ms per GiB of transfer: it keeps
        // up, so that is the last piece alone -- round 4's quarter pieces left a full piece + a remainder: 14 ms at 15 GB)
        const uint64_t pm = capped ? piece_max : std::min<uint64_t>(piece_max, std::max<uint64_t>(64ull << 20, (size - off) / 3));
        uint64_t end = std::min<uint64_t>(size, off + pm);
        if (end < size) {   // back to the last line end inside the piece
            const void *nl = memrchr(text + off, '\n', (size_t)(end - off));
            if (!nl && !capped) {   // a line longer than the default piece: forward to its end (pieces stay below 3.5 GiB)
                const uint64_t far = std::min<uint64_t>(size, off + 0xE0000000ull);
                const void *fw = memchr(text + end, '\n', (size_t)(far - end));
                if (fw) nl = fw;
                else if (far == size) { end = size; nl = text; }
            }
            if (!nl) return fail(ctx, PANTAX_HIP_E_LIMIT, "gaf_tokenize: a line of more than %llu bytes at offset %llu", (unsigned long long)(capped ? piece_max : 0xE0000000ull), (unsigned long long)off);
            if (end != size) end = (uint64_t)(static_cast<const char *>(nl) - text) + 1;
        }
        piece_off.push_back(off); piece_end.push_back(end);
        longest = std::max(longest, end - off);
        off = end;
    }
    const size_t NP = piece_off.size();
    GafJoined J;
    {
        constexpr size_t RING = 3;
        struct Shared {
            std::mutex mu;
            std::condition_variable cv;
            size_t uploaded = 0, tokenised = 0;   // pieces whose text is in HBM / whose text buffer is free again
            int rc = 0;
            bool stop = false;
        } sh;
        DevBuf<uint8_t> txt[RING];
        for (size_t k = 0; k < std::min(RING, NP); ++k) PTX_HIP(ctx, txt[k].alloc(longest + 16));
        GafWork W;
        lap("setup: piece cuts, text buffers");
        hipStream_t up_stream = nullptr;
        PTX_HIP(ctx, hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking));
        std::vector<hipEvent_t> ev_piece(NP, nullptr);
        for (auto &e : ev_piece) PTX_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        std::vector<void *> dsts(NP);
        for (size_t k = 0; k < NP; ++k) dsts[k] = txt[k % RING].p;
        std::vector<uint64_t> dev_size(NP);
        for (size_t k = 0; k < NP; ++k) dev_size[k] = piece_end[k] - piece_off[k];
        std::thread uploader([&] {
            (void)hipSetDevice(ctx->device);
            auto gate = [&](size_t k) {   // the text buffer of piece k is free once piece k - RING has been tokenised
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv.wait(lk, [&] { return sh.stop || k < sh.tokenised + RING; });
                return !sh.stop;
            };
            auto arrived = [&](size_t k, uint64_t bytes) {   // all of piece k is on the upload stream: the tokenizer's stream waits for this event
                if (hipEventRecord(ev_piece[k], up_stream) != hipSuccess) return fail(ctx, PANTAX_HIP_E_HIP, "gaf_tokenize: hipEventRecord failed");
                std::lock_guard<std::mutex> g(sh.mu);
                dev_size[k] = bytes;
                sh.uploaded = k + 1;
                sh.cv.notify_all();
                return 0;
            };
            const int rc = upload_text_pieces(ctx, NP, dsts.data(), text, fd, file_base, piece_off.data(), piece_end.data(), up_stream, gate,
                                              [&](size_t k) { return arrived(k, piece_end[k] - piece_off[k]); });
            std::lock_guard<std::mutex> g(sh.mu);
            if (rc != 0 && !sh.stop) { sh.rc = rc; sh.stop = true; }
            sh.cv.notify_all();
        });
        int rc = 0;
        std::thread prefault;   // the host columns (14 bytes per read) are allocated and page-faulted beside the upload, not after it
        for (size_t k = 0; k < NP && rc == 0; ++k) {
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv.wait(lk, [&] { return sh.stop || sh.uploaded > k; });
                if (sh.uploaded <= k) { rc = sh.rc ? sh.rc : PANTAX_HIP_E_HIP; break; }
            }
            if (hipStreamWaitEvent(ctx->stream, ev_piece[k], 0) != hipSuccess) { rc = fail(ctx, PANTAX_HIP_E_HIP, "gaf_tokenize: hipStreamWaitEvent failed"); }
            if (rc == 0) {
                const uint64_t in_size = piece_end[k] - piece_off[k], ds = dev_size[k];   // what follows, at this piece's ratio of device bytes to text bytes
                const uint64_t rest = in_size ? (uint64_t)((double)(size - piece_end[k]) * ((double)ds / (double)in_size)) : 0;
                rc = tokenize_piece(ctx, txt[k % RING].p, ds, text[piece_end[k] - 1] == '\n', rest, W, J);
            }
            if (rc == 0 && k == 0 && NP > 1 && want_host_columns) {
                const uint64_t est = J.cap_r;
                prefault = std::thread([&out, est] { out.qlen.resize(est); out.mapq.resize(est); out.flags.resize(est); out.id_hash.resize(est); });
            }
            if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, PANTAX_HIP_E_HIP, "gaf_tokenize: piece %zu failed", k);   // the text buffer goes back to the uploader
            std::lock_guard<std::mutex> g(sh.mu);
            sh.tokenised = k + 1;
            if (rc != 0) sh.stop = true;
            sh.cv.notify_all();
        }
        { std::lock_guard<std::mutex> g(sh.mu); if (rc != 0) sh.stop = true; sh.cv.notify_all(); }
        uploader.join();
        if (prefault.joinable()) prefault.join();
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(up_stream);   // on every path: copies from the pinned ring into the text buffers may still be in flight (round-3 advisor finding)
        if (rc != 0 && sh.rc != 0) {             // the uploader thread failed: its message sits in ITS thread's slot -- this thread reports it
            std::string msg;
            { std::lock_guard<std::mutex> g(ctx->err_mu); msg = ctx->err; }
            (void)fail(ctx, rc, "%s", msg.c_str());
        }
        (void)hipStreamDestroy(up_stream);
        for (auto &e : ev_piece) if (e) (void)hipEventDestroy(e);
        if (rc != 0) return rc;
    }
    lap("pieces: upload | scan + parse + fill");
    const uint64_t R = J.R, T = J.T;
    // id spans: piece-local 32-bit positions -> positions in the whole text
    if (want_id_spans) {
        out.id_span.resize(R);
        std::vector<uint32_t> id_off(R), id_len(R);
        PTX_TRY(download(ctx, id_off.data(), J.o32[5].p, R)); PTX_TRY(download(ctx, id_len.data(), J.o32[6].p, R));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t k = 0; k < NP; ++k) {
            const uint64_t base = piece_off[k], ra = J.piece_r0[k], rb = k + 1 < NP ? J.piece_r0[k + 1] : R;
            parallel_for(rb - ra, 8, [&](uint64_t i0, uint64_t i1) { for (uint64_t i = ra + i0; i < ra + i1; ++i) out.id_span[i] = {base + (uint64_t)id_off[i], id_len[i]}; });
        }
    }
    DevBuf<uint32_t> o32[5];
    DevBuf<uint8_t> o8[2];
    DevBuf<uint64_t> o_hash;
    for (int k = 0; k < 5; ++k) o32[k].take(J.o32[k]);
    o8[0].take(J.o8[0]); o8[1].take(J.o8[1]); o_hash.take(J.o_hash);
    if (!o32[0].p) {   // no piece at all
        static const uint32_t zero = 0;
        PTX_TRY(upload(ctx, o32[0], &zero, 1));
        for (int k = 1; k < 5; ++k) PTX_HIP(ctx, o32[k].alloc(1));
        for (auto &b8 : o8) PTX_HIP(ctx, b8.alloc(1));
        PTX_HIP(ctx, o_hash.alloc(1));
    }
    DevBuf<uint32_t> tot, scan_tmp;
    PTX_HIP(ctx, tot.alloc(4)); PTX_HIP(ctx, scan_tmp.alloc(16));
    if (want_host_columns) { out.qlen.resize(R); out.mapq.resize(R); out.flags.resize(R); out.id_hash.resize(R); }
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
    if (J.set_ok) {                                     // decided piece by piece, beside the upload
        if (R > 1) PTX_TRY(download(ctx, &n_dup, J.dup_cnt.p, 1));
    } else if (R > 1) {
        J.id_set.release();                             // the set overflowed (16 bytes per read of HBM): gone before the sort's buffers come

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
    if (want_host_columns) {
        PTX_TRY(download(ctx, out.qlen.data(), o32[4].p, R));
        PTX_TRY(download(ctx, out.mapq.data(), o8[0].p, R)); PTX_TRY(download(ctx, out.flags.data(), o8[1].p, R));
        PTX_TRY(download(ctx, out.id_hash.data(), o_hash.p, R));
    }
    if (resident) PTX_TRY(download(ctx, &max_id, tot.p + 3, 1));
    PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lap("host columns download");
    if (resident) {
        resident->R = R; resident->T = T;
        resident->d_step_off.take(o32[0]); resident->d_node_id.take(o32[1]); resident->d_pstart.take(o32[2]); resident->d_pend.take(o32[3]);
        resident->d_qlen.take(o32[4]); resident->d_mapq.take(o8[0]); resident->d_flags.take(o8[1]); resident->d_id_hash.take(o_hash);
        resident->has_flags = true;
        resident->max_node_id = max_id;
        if (group) PTX_TRY(build_step_read(ctx, resident, max_id));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        lap("locus-grouped copy");
    }
    out.n_lines = R;
    out.ids_distinct = n_dup == 0 ? 1 : 0;
    if (trace) std::fprintf(stderr, "[gaf_tokenize] total inside gaf_tokenize_device      %9.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count());
    return 0;
}

int reads_host_columns(Ctx *ctx, const Reads *rd, HostReads &out) {
    const uint64_t R = rd->R;
    if (R && (!rd->d_qlen.p || !rd->d_mapq.p || !rd->d_flags.p || !rd->d_id_hash.p)) return fail(ctx, PANTAX_HIP_E_STATE, "reads_host_columns: these reads were not tokenised on the device");
    out.qlen.resize(R); out.mapq.resize(R); out.flags.resize(R); out.id_hash.resize(R);
    if (R) {
        PTX_TRY(download(ctx, out.qlen.data(), rd->d_qlen.p, R));
        PTX_TRY(download(ctx, out.mapq.data(), rd->d_mapq.p, R)); PTX_TRY(download(ctx, out.flags.data(), rd->d_flags.p, R));
        PTX_TRY(download(ctx, out.id_hash.data(), rd->d_id_hash.p, R));
        PTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
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
    // without a gaf handle nobody can read the host-side columns (read_len, mapq, flags, id hashes): they are not downloaded
    PTX_TRY(gaf_tokenize_device(ctx, g->mf.data, g->mf.size, g->reads, rd.get(), g->mf.fd, 0, true, false, gaf_out != nullptr));
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
