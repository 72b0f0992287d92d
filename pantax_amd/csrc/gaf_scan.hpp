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

}  // namespace ptx
