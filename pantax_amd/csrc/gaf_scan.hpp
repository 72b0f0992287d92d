// gaf_scan.hpp -- shared front end of the device GAF readers (stage_gaf.hip, stage_gaf_filter.hip): the text goes to
// HBM once through pinned chunks and every newline position is listed (two launches + one chained scan).
#pragma once
#include "common.hpp"

namespace ptx {

// d_txt: size + 16 bytes; nl_pos[n_nl] = byte offsets of the '\n's in order.  Line i = (nl_pos[i-1], nl_pos[i]).
int gaf_upload_and_scan(Ctx *ctx, const char *text, uint64_t size, DevBuf<uint8_t> &d_txt, DevBuf<uint32_t> &nl_pos, uint32_t *n_nl_out,
                        int fd = -1 /* the open file behind `text`: uploaded with pread instead of through the mapping */,
                        uint64_t file_off = 0 /* where `text` starts in that file */);

// the scan alone, for a text that is already in HBM (d_txt: size + 16 bytes)
int gaf_scan_newlines(Ctx *ctx, uint64_t size, DevBuf<uint8_t> &d_txt, DevBuf<uint32_t> &nl_pos, uint32_t *n_nl_out);

// ... with the caller's grow-only work buffers (nothing is released, nothing waited for at the end); d_txt: device pointer
int gaf_scan_newlines_ws(Ctx *ctx, uint64_t size, const uint8_t *d_txt, DevBuf<uint32_t> &nl_pos, uint32_t *n_nl_out, DevBuf<uint32_t> &tile_cnt,
                         DevBuf<uint32_t> &tile_base, DevBuf<uint32_t> &tot, DevBuf<uint32_t> &scan_tmp);

#ifdef __HIPCC__
// A line's bytes through an 8-byte window in registers (round 6): the text is read by aligned 8-byte loads, one per eight bytes a thread walks, instead of a
// 1-byte load -- and its L1 round trip -- per byte; rounds 2-5 walked every line byte by byte (13.8 ms per 693-MB piece, as long as the piece's PCIe transfer).
// The loads reach at most 7 bytes in front of / behind the range asked for: inside the allocation (the text buffers are 256-byte aligned and 16 bytes longer
// than the text).
struct TxtWin {
    const uint8_t *txt;
    uintptr_t cur = ~(uintptr_t)0;
    uint64_t w = 0;
    __device__ __forceinline__ explicit TxtWin(const uint8_t *t) : txt(t) {}
    __device__ __forceinline__ void seek(uintptr_t a8) { if (a8 != cur) { w = *reinterpret_cast<const uint64_t *>(a8); cur = a8; } }
    __device__ __forceinline__ uint32_t at(uint32_t pos) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(txt) + pos;
        seek(a & ~(uintptr_t)7);
        return (uint32_t)(w >> ((a & 7u) * 8u)) & 0xFFu;
    }
    // first position in [from, end) that holds byte `ch` (end: none), eight bytes per step: zero-byte test of w ^ ch-in-every-byte
    __device__ __forceinline__ uint32_t find(uint32_t from, uint32_t end, uint32_t ch) {
        const uint64_t pat = 0x0101010101010101ull * (uint64_t)ch;
        uint32_t t = from;
        while (t < end) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(txt) + t;
            seek(a & ~(uintptr_t)7);
            const uint32_t skip = (uint32_t)(a & 7u);
            const uint64_t x = (w ^ pat) | ((1ull << (8u * skip)) - 1ull);          // the bytes in front of `t`: never a match
            const uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;   // (the LOWEST set flag is exact: borrows only travel upwards)
            if (z) { const uint32_t hit = t - skip + ((uint32_t)__builtin_ctzll(z) >> 3); return hit < end ? hit : end; }
            t += 8u - skip;
        }
        return end;
    }
};

#endif

}  // namespace ptx
