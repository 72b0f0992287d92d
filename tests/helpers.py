"""Shared helpers for the parity tests (inputs from golden JSON / synthetic sets)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_micro_cov():
    with open(os.path.join(ROOT, "tests", "golden", "micro_cov.json")) as f:
        j = json.load(f)
    names = sorted(j["paths"].keys())  # BTreeMap byte order
    node_len = np.array(j["node_len"], dtype=np.int64)
    path_off = np.zeros(len(names) + 1, dtype=np.uint64)
    path_off[1:] = np.cumsum([len(j["paths"][n]) for n in names])
    path_nodes = np.concatenate([np.array(j["paths"][n], dtype=np.uint32) for n in names])
    rs = j["range_start"]
    step_off = np.zeros(len(j["reads"]) + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum([len(r["walk"]) for r in j["reads"]])
    node_id = np.array([w + rs for r in j["reads"] for w in r["walk"]], dtype=np.uint32)
    pstart = np.array([r["pstart"] for r in j["reads"]], dtype=np.int64)
    pend = np.array([r["pend"] for r in j["reads"]], dtype=np.int64)
    return j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend


def load_micro_bin():
    with open(os.path.join(ROOT, "tests", "golden", "micro_bin.json")) as f:
        j = json.load(f)
    step_off = np.zeros(len(j["reads"]) + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum([len(r["walk"]) for r in j["reads"]])
    node_id = np.array([w for r in j["reads"] for w in r["walk"]], dtype=np.uint32)
    qlen = np.array([r["qlen"] for r in j["reads"]], dtype=np.int64)
    mapq = np.array([r["mapq"] for r in j["reads"]], dtype=np.int64)
    species = np.array([r["species"] for r in j["reads"]], dtype=np.int32)
    rs = np.array([r[1] for r in j["ranges"]], dtype=np.int64)
    re = np.array([r[2] for r in j["ranges"]], dtype=np.int64)
    return j, step_off, node_id, qlen, mapq, species, rs, re


def select_reads(reads, sel):
    """Sub-select reads `sel` (indices) from a PackedReads -> (step_off, node_id, pstart, pend)."""
    so = reads.step_off.astype(np.int64)
    ns = (so[1:] - so[:-1])[sel]
    new_off = np.zeros(len(sel) + 1, dtype=np.uint64)
    new_off[1:] = np.cumsum(ns)
    starts = so[:-1][sel]
    tot = int(ns.sum())
    idx = np.repeat(starts, ns) + (np.arange(tot) - np.repeat(new_off[:-1].astype(np.int64), ns))
    return new_off, reads.node_id[idx], reads.pstart[sel], reads.pend[sel]


def make_longread_gaf(seed, n_reads, path_ids=40):
    """GAF text in GraphAligner's column layout (12 columns + NM, AS, dv, id tags) with several alignments per read, ties,
    low-mapq / short-span lines and every malformed spelling parse_line (gaf_filter.rs:21-42) has to reject."""
    rng = np.random.default_rng(seed)
    lines = []
    ident_forms = ["%.6f", "%.3f", "%.17g", "%e", "%.2e"]
    for r in range(n_reads):
        rid = "m64_%d/%d/ccs" % (seed, r)
        n_aln = int(rng.choice([1, 1, 1, 2, 2, 3, 5]))
        base_m = int(rng.integers(800, 20000))
        for a in range(n_aln):
            qlen = int(rng.integers(1200, 25000))
            qs = int(rng.integers(0, 200))
            qe = qs + int(rng.choice([rng.integers(50, 1001), rng.integers(1001, qlen + 1)], p=[0.2, 0.8]))
            matches = base_m if rng.random() < 0.4 else int(rng.integers(100, 20000))
            ident = float(rng.choice([0.99, 0.987654, rng.random(), 1.0]))
            mapq = int(rng.choice([0, 20, 21, 60], p=[0.1, 0.1, 0.2, 0.6]))
            path = "".join(rng.choice([">", "<"]) + str(int(x)) for x in rng.integers(1, 10**7, size=int(rng.integers(1, path_ids))))
            idtxt = ident_forms[int(rng.integers(0, len(ident_forms)))] % ident
            f = [rid, str(qlen), str(qs), str(qe), "+", path, "12345", "7", "12000", str(matches), str(qe - qs), str(mapq),
                 "NM:i:%d" % int(rng.integers(0, 50)), "AS:f:%.1f" % (matches * 0.9), "dv:f:%.4f" % (1 - ident), "id:f:" + idtxt]
            u = rng.random()
            if u < 0.02: f = f[:15]                                   # too few fields
            elif u < 0.03: f[9] = "*"                                  # matches not a number
            elif u < 0.04: f[9] = "+" + f[9]                           # explicit sign is fine for i32
            elif u < 0.05: f[15] = "id:f:nan"
            elif u < 0.06: f[15] = "id:f:" + rng.choice(["inf", "-inf", "Infinity", "1e400", "1e-400", ".5", "5.", ".", "0x10", "1_0", "", "1e", "+.5e+1"])
            elif u < 0.07: f[15] = "id:f:0.98765432109876543210987"   # more digits than the fast path holds
            elif u < 0.08: f[11] = "99999999999"                       # i32 overflow
            elif u < 0.09: f[3] = "-5"                                 # negative end: span <= 1000
            elif u < 0.10: f.append("cg:Z:150=")                       # extra columns are ignored
            elif u < 0.11: f[15] = "0.97"                              # no ':' at all: the whole field is the number
            line = "\t".join(f)
            v = rng.random()
            if v < 0.02: line = "  " + line + " "                     # trimmed before splitting
            elif v < 0.04: line = line + "\r"
            elif v < 0.05: line = line + "\t"                          # trailing tab is trimmed away
            lines.append(line)
        if rng.random() < 0.01: lines.append("")
        if rng.random() < 0.01: lines.append("@comment line")
    order = rng.permutation(len(lines))                               # alignments of a read are not adjacent in general
    txt = "\n".join(lines[i] for i in order)
    return (txt + ("\n" if seed % 2 else "")).encode()
