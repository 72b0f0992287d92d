"""A SECOND, independently written restatement of the long-read best-alignment filter -- TEST INFRASTRUCTURE ONLY.

gaf_filter.rs:21-97 (parse_line, filter_max_alignment_mt) in plain Python, mirroring the reference's statements; it shares
no code with oracle/pantax_oracle.c.  Rust's `str::parse::<i32>` / `::<f64>` grammars are spelled out as regular
expressions (Python's int() / float() accept more: surrounding blanks, underscores).

The reference writes from a rayon loop: which of several equal-best passing lines of a read is written, and in which order
the lines appear, is a scheduling accident.  What IS defined: the set of reads that get a line, and for each of them the set
of lines that may be it.  `candidates(text)` returns exactly that; the library's convention (first candidate in file order,
output in file order) is one admissible outcome.
"""
import math
import re

_I32 = re.compile(r"^[+-]?[0-9]+$")
_F64 = re.compile(r"^[+-]?(?:inf|infinity|nan|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)$", re.IGNORECASE)


def parse_i32(s):
    if not _I32.match(s):
        return None
    v = int(s)
    return v if -(1 << 31) <= v < (1 << 31) else None


def parse_f64(s):
    if not _F64.match(s):
        return None
    return float(s)


def parse_line(line):
    """gaf_filter.rs:21-42"""
    fields = line.strip().split("\t")                      # line.trim().split('\t') (:22)
    if len(fields) < 16:                                   # :23
        return None
    read_id = fields[0]
    align_10 = parse_i32(fields[9])                        # :28
    if align_10 is None:
        return None
    align_16 = parse_f64(fields[15].rsplit(":", 1)[-1])    # rsplit(':').next() = the part after the last ':' (:29)
    if align_16 is None:
        return None
    qual_12 = parse_i32(fields[11])                        # :30
    if qual_12 is None:
        return None
    e, b = parse_i32(fields[3]), parse_i32(fields[2])      # :31
    if e is None or b is None:
        return None
    span = e - b
    if not -(1 << 31) <= span < (1 << 31):                 # i32 subtraction overflow panics in debug, wraps in release: such lines do not occur
        return None
    return dict(line=line, read_id=read_id, align_10=align_10, align_16=align_16, qual_12=qual_12, span=span)


def candidates(text):
    """text: str.  -> (n_records, dict read_id -> list of line indices that filter_max_alignment_mt may write for it)"""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                                        # BufRead::lines: no empty line after the final newline
    lines = [l[:-1] if l.endswith("\r") else l for l in lines]   # lines() strips "\r\n" as well
    records = []
    for i, l in enumerate(lines):                          # :57-60
        r = parse_line(l)
        if r is not None:
            r["idx"] = i
            records.append(r)
    best = {}                                              # DashMap<String, (i32, f64)> (:63)
    for r in records:                                      # :67-76; the comparison is order-independent except for NaN identities
        e = best.get(r["read_id"])
        if e is None:
            best[r["read_id"]] = (r["align_10"], r["align_16"])
        elif r["align_10"] > e[0] or (r["align_10"] == e[0] and r["align_16"] > e[1]):
            best[r["read_id"]] = (r["align_10"], r["align_16"])
    out = {}
    for r in records:                                      # :82-95
        if not (r["qual_12"] > 20 and r["span"] > 1000):
            continue
        b10, b16 = best[r["read_id"]]
        if r["align_10"] == b10 and r["align_16"] == b16:  # NaN == NaN is false, as in Rust
            out.setdefault(r["read_id"], []).append(r["idx"])
    return len(records), out


def has_nan_identity(text):
    """a read whose best-candidate choice depends on the visiting order (NaN identity among equal align_10): left out of strict comparisons"""
    for l in text.split("\n"):
        r = parse_line(l[:-1] if l.endswith("\r") else l)
        if r is not None and math.isnan(r["align_16"]):
            return True
    return False
