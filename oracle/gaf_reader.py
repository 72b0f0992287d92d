"""oracle/gaf_reader.py -- TEST INFRASTRUCTURE (checker only; nothing under pantax_amd/ imports it).

An independent restatement of how the reference reads a GAF file, written from the reference's call and the documented
behaviour of the reader it configures -- NOT from this repo's tokenizers (pantax_amd/csrc/host_io.cpp, stage_gaf.hip,
gaf_prune.cc), which are what it checks.

Reference: `load_gaf_file_lazy`, rcls.rs:119-137:

    LazyCsvReader::new(path).with_has_header(false).with_separator(b'\\t')
        .with_comment_prefix(Some("@")).with_null_values(Some(AllColumnsSingle("*"))).with_quote_char(None)
    select column_1 read_id, column_2 read_len, column_6 path, column_7/8/9 cast Int64 (read_path_len, read_start,
    read_end), column_12 mapq

and the walk of a read = every maximal run of ASCII digits of its path string (`Regex \\d+` + `find_iter`,
rcls.rs:237-258, profile.rs:790-796).

Rules of the configured reader (polars 0.46 `CsvReadOptions`; each is one line of the call above):
  L1  records end at '\\n'; a '\\r' directly in front of it belongs to the line end (the reader's eol handling of CR LF)
  L2  a line whose first byte is '@' is a comment and yields no row (`with_comment_prefix`)
  L3  an empty line yields no row
  F1  fields are separated by '\\t' and by nothing else; quotes have no meaning (`with_quote_char(None)`)
  F2  a field that is exactly "*" is null, in every column (`AllColumnsSingle("*")`); so is an EMPTY field (the reader's default
      `missing_is_null`: an empty string field is a missing value, also in string columns)
  F3  a row with fewer fields than the frame has columns has nulls in the missing ones; fields beyond column 12 are not selected
  C1  column_7..9 are cast to Int64: a value that is not an integer becomes null (non-strict cast); column_2 and column_12 are
      integer columns by inference, a null stays null
What the reference does NOT define is a file whose integer columns hold non-integers (the reader then errors out or infers
a string column): such values are read as null here, which is what a non-strict cast of them gives.

Output = the packed layout the tests compare (the same dict pantax_amd.io.load_gaf returns): 32-bit columns, so an integer
above 2^32-1 is clamped to 2^32-1 there (a limit of the packed layout, stated in include/pantax_hip.h), null read_len = 0, null
mapq = 255, flags bit 0 = "path, read_path_len, read_start or read_end is null" (the rows get_node_abundances cannot use,
profile.rs:361-437 drops them from the strain level)."""
import re

import numpy as np

_DIGITS = re.compile(rb"[0-9]+")
_U32 = 0xFFFFFFFF


def _int_or_none(field):
    """Int64 view of a field: None for null ("*", missing) and for anything that is not a plain run of digits (C1)"""
    if field is None or field == b"*" or not field or not field.isdigit():
        return None
    v = int(field)
    return v if v < (1 << 63) else None


def rows(text: bytes):
    """-> list of (read_id, read_len, path, read_path_len, read_start, read_end, mapq); None = null"""
    out = []
    for line in text.split(b"\n"):                       # L1 (a text without a final '\n' ends with its last line)
        if line.endswith(b"\r"):
            line = line[:-1]
        if not line or line.startswith(b"@"):            # L3, L2
            continue
        f = line.split(b"\t")                            # F1
        col = lambda k: (None if f[k] in (b"*", b"") else f[k]) if k < len(f) else None   # F2, F3 (k = 0-based column)
        out.append((col(0), _int_or_none(col(1)), col(5), _int_or_none(col(6)), _int_or_none(col(7)), _int_or_none(col(8)), _int_or_none(col(11))))
    return out


def packed(text: bytes):
    """the rows as the packed arrays of pantax_amd.io.load_gaf"""
    rs = rows(text)
    step_off, node_id, pstart, pend, qlen, mapq, flags = [0], [], [], [], [], [], []
    clamp = lambda v: 0 if v is None else min(v, _U32)
    for (_, ln, path, plen, ps, pe, mq) in rs:
        if path is not None:
            node_id.extend(min(int(m), _U32) for m in _DIGITS.findall(path))
        step_off.append(len(node_id))
        pstart.append(clamp(ps)); pend.append(clamp(pe)); qlen.append(clamp(ln))
        mapq.append(255 if mq is None else min(mq, 255))
        flags.append(1 if (path is None or plen is None or ps is None or pe is None) else 0)
    return dict(step_off=np.array(step_off, dtype=np.uint32), node_id=np.array(node_id, dtype=np.uint32), pstart=np.array(pstart, dtype=np.uint32),
                pend=np.array(pend, dtype=np.uint32), qlen=np.array(qlen, dtype=np.uint32), mapq=np.array(mapq, dtype=np.uint8),
                flags=np.array(flags, dtype=np.uint8))


def read_ids(text: bytes):
    return [r[0] for r in rows(text)]
