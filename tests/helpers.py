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


def gaf_quirks_text():
    """Every quirk of the GAF contract (rcls.rs:119-137 through the reader it configures, oracle/gaf_reader.py): comments, '*' nulls in
    every selected column, CR LF, ragged lines of every length, empty lines, empty fields, 13+ fields and long tags, non-integers in
    integer columns, numbers beyond 32 bits, digits inside non-numeric path text, no trailing newline."""
    long_tag = b"cs:Z:" + b":150" * 400
    return (b"@HD\tVN:1.0\n"
            b"r1\t150\t0\t150\t+\t>12<7>300\t400\t3\t153\t150\t150\t60\tNM:i:0\n"
            b"r2\t150\t0\t150\t+\t*\t*\t*\t*\t*\t*\t255\n"
            b"r3\t100\t0\t100\t+\t<5\t30\t20\t10\t100\t100\t*\r\n"
            b"\n"
            b"\r\n"
            b"r5\t99999999999\t0\t1\t+\t>4294967296>7\t1\t99999999999\t5\t1\t1\t300\ta\tb\tc\n"
            b"r6\tx12\t0\t1\t+\t>1>2\t12x\t3\t4\t1\t1\t7\n"
            b"r7\t10\t0\t10\t+\tabc>>9<<10zz11\t5\t\t6\n"
            b"@ comment in the middle\n"
            b"r8\n"
            b"*\t*\t*\t*\t*\t>3>4\t9\t1\t8\t*\t*\t17\n"                       # null id and length, the rest usable
            b"r10\t150\t0\t150\t+\t>21>22>23\t*\t0\t150\t150\t150\t60\n"           # only read_path_len null
            b"r11\t150\t0\t150\t+\t>21>22>23\t300\t*\t150\t150\t150\t60\n"         # only read_start null
            b"r12\t150\t0\t150\t+\t>21>22>23\t300\t0\t*\t150\t150\t60\n"           # only read_end null
            b"r13\t150\t0\t150\t+\t>21>22>23\t300\t0\t150\t150\t150\t60\t" + long_tag + b"\tNM:i:3\r\n"   # a long tag, CR LF behind it
            b"r14\t150\t0\t150\t+\t>5\t300\t7\t9\t150\t150\t60\r\n"                # exactly twelve fields + CR LF
            b"r15\t150\t0\t150\t+\t>5\t300\t7\t9\t150\t150\n"                       # eleven fields: no mapq column
            b"r16\t150\t0\t150\t+\t>5\t300\t7\t9\t150\n"                            # ten
            b"r17\t150\t0\t150\t+\t>5\n"                                             # six: the path is the last field
            b"r18\t150\t0\t150\t+\n"                                                 # five: no path column
            b"\t\t\t\t\t\t\t\t\t\t\t\n"                                             # twelve empty fields
            b"r19 with spaces\t+150\t0\t150\t+\t>7 >8\t 300\t7 \t9\t150\t150\t 60\n"  # spaces are not separators; '+150', ' 300' are not integers
            b"r20\t150\t0\t150\t+\t>007>08\t0300\t007\t0009\t150\t150\t060\n"       # leading zeros
            b"r21\t150\t0\t150\t+\t>1>2\t300\t-1\t150\t150\t150\t60\n"             # a negative read_start is an integer in the reference; the packed layout has no sign
            b"r22\t150\t0\t150\t+\t\t300\t0\t150\t150\t150\t60\n"                   # an EMPTY path field is a missing value: null
            b"@tail comment\n"
            b"r9\t90\t0\t90\t+\t>8>9\t200\t0\t90")


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


def slice_reads(reads, a, b):
    """Reads [a, b) of a PackedReads as a PackedReads of their own (step offsets rebased, arrays are views)."""
    from synthdata import PackedReads
    so = reads.step_off
    t0, t1 = int(so[a]), int(so[b])
    return PackedReads(so[a:b + 1] - so[a], reads.node_id[t0:t1], reads.strand[t0:t1], reads.pstart[a:b], reads.pend[a:b], reads.qlen[a:b],
                       reads.mapq[a:b], reads.plen[a:b], reads.read_id[a:b] if reads.read_id else [])


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


def expected_total_bases(sset, sp):
    """Sum of bases_per_node over all nodes from the read records alone (numpy, no kernel logic), sized for 10^7..10^8
    steps: one-node read -> target (dropped if < 0); otherwise seen = (len0 - ps) + interior lengths and the read
    contributes seen + max(target - seen, 0), minus the aligned length of repeated node occurrences (profile.rs:879-882).
    Repeats can only occur in walks that are not strictly monotone in node id; those few are handled read by read."""
    rd = sset.reads
    so = rd.step_off.astype(np.int64)
    k = so[1:] - so[:-1]
    gl = np.concatenate([g.node_len for g in sset.species])
    base = np.cumsum([0] + [g.n_nodes for g in sset.species])[:-1]
    first = np.array([g.range_start for g in sset.species])
    nn = np.array([g.n_nodes for g in sset.species])
    live = (sp >= 0) & (k > 0)
    spc = np.where(sp >= 0, sp, 0).astype(np.int64)
    sp_step = np.repeat(spc, k)
    live_step = np.repeat(live, k)
    loc = rd.node_id.astype(np.int64) - first[sp_step]
    v = np.where(live_step & (loc >= 0) & (loc < nn[sp_step]), base[sp_step] + loc, 0)
    del sp_step, loc
    ln = gl[v]
    c = np.concatenate([[0], np.cumsum(ln)])
    tot_len = c[so[1:]] - c[so[:-1]]
    idx0 = np.minimum(so[:-1], len(ln) - 1)
    idxl = np.maximum(so[1:] - 1, 0)
    len0 = np.where(k > 0, ln[idx0], 0)
    lenl = np.where(k > 0, ln[idxl], 0)
    target = rd.pend - rd.pstart
    seen = tot_len - lenl - rd.pstart
    multi = np.where(target > seen, target, seen)
    single = np.where(target >= 0, target, 0)
    abort = (k >= 2) & (rd.pstart > len0)
    per_read = np.where(k == 1, single, multi)
    per_read = np.where(live & ~abort, per_read, 0)
    # walks that are not strictly monotone: candidates for repeated nodes
    d = np.diff(v)
    inner = np.ones(len(v), dtype=bool)
    inner[so[:-1][k > 0]] = False                 # d[t-1] = v[t]-v[t-1] is a within-read difference iff t is not a read start
    dd = np.zeros(len(v), dtype=np.int64)
    dd[1:] = d
    up = np.concatenate([[0], np.cumsum(inner & (dd > 0))])
    dn = np.concatenate([[0], np.cumsum(inner & (dd < 0))])
    n_up = up[so[1:]] - up[so[:-1]]
    n_dn = dn[so[1:]] - dn[so[:-1]]
    mono = (n_up == np.maximum(k - 1, 0)) | (n_dn == np.maximum(k - 1, 0))
    rep_fix = 0
    for r in np.nonzero(~mono & live & ~abort & (k >= 2))[0]:
        ids = v[so[r]:so[r + 1]]
        lens = ln[so[r]:so[r + 1]].copy()
        lens[0] -= rd.pstart[r]
        s_ = int(lens[:-1].sum())
        lens[-1] = max(int(target[r]), s_) - s_
        seen_nodes = set()
        for j, n in enumerate(ids):
            if n in seen_nodes:
                rep_fix += int(lens[j])
            seen_nodes.add(n)
    return int(per_read.sum()) - rep_fix


def oracle_species_checks(sset, sp, keep, absolute, bases, cov, tb, hto, gmet, info, species_idx, threads=8, strain_kw=None, lp=True):
    """Species by species against the oracle (TEST checker), on `threads` host threads (the C calls release the GIL):
    bit-exact bases / cov / trio bases, and -- when `lp` -- candidate count, LP objective (1e-9) and every reported metric.
    -> list of (species, what) mismatches."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    rd = sset.reads
    S = len(sset.species)
    nb = np.cumsum([0] + [g.n_nodes for g in sset.species])
    hb = np.cumsum([0] + [g.n_paths for g in sset.species])
    first_, order = orc.group_reads(sp, S)         # reads grouped by species, file order inside (stable counting sort)
    first = first_[:-1].astype(np.int64)
    cnt = np.diff(first_.astype(np.int64))
    order = order.astype(np.int64)
    strain_kw = strain_kw or {}

    def one(s):
        bad = []
        g = sset.species[s]
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        sel = np.sort(order[first[s]:first[s] + cnt[s]])
        so, nid, ps, pe = select_reads(rd, sel)
        b, c, t, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        u0, u1 = int(hto[hb[s]]), int(hto[hb[s + 1]])
        if not np.array_equal(bases[nb[s]:nb[s + 1]], b): bad.append((s, "bases"))
        if not np.array_equal(cov[nb[s]:nb[s + 1]], c): bad.append((s, "cov"))
        if u1 - u0 != T.n_unique or not np.array_equal(tb[u0:u1], t): bad.append((s, "trio_bases"))
        if lp and keep[s]:
            rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t, **strain_kw)
            orc.abundance_constraint(absolute[s], omet)
            if info[s].n_candidates != nc or info[s].status1 != 0: bad.append((s, "candidates %d vs %d status %d" % (info[s].n_candidates, nc, info[s].status1)))
            elif nc and abs(info[s].obj1 - o1) > 1e-9 * max(1.0, abs(o1)): bad.append((s, "objective %r vs %r" % (info[s].obj1, o1)))
            else:
                for h, (gm, em) in enumerate(zip(gmet[hb[s]:hb[s + 1]], orc.metrics_to_dicts(omet))):
                    for key, ev in em.items():
                        gv = gm[key]
                        if ev is None or isinstance(ev, bool):
                            if gv != ev: bad.append((s, "%s[%d] %r vs %r" % (key, h, gv, ev)))
                        elif gv is None or abs(gv - ev) > 1e-7 * abs(ev) + 1e-9: bad.append((s, "%s[%d] %r vs %r" % (key, h, gv, ev)))
        return bad

    with ThreadPoolExecutor(threads) as ex:
        out = list(ex.map(one, species_idx))
    return [b for lst in out for b in lst]


def load_literal_case(k):
    """tests/golden/literal_cov_<k>.json (generated by oracle/gen_golden_literal.py from the literal Python reading of
    profile.rs:658-1026) -> graph arrays in the packed layouts + the expected outputs."""
    with open(os.path.join(ROOT, "tests", "golden", "literal_cov_%d.json" % k)) as f:
        j = json.load(f)
    names = sorted(j["paths"].keys())
    node_len = np.array(j["node_len"], dtype=np.int64)
    path_off = np.zeros(len(names) + 1, dtype=np.uint64)
    path_off[1:] = np.cumsum([len(j["paths"][n]) for n in names])
    path_nodes = np.concatenate([np.array(j["paths"][n], dtype=np.uint32) for n in names])
    step_off = np.zeros(len(j["reads"]) + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum([len(r["walk"]) for r in j["reads"]])
    node_id = np.array([w for r in j["reads"] for w in r["walk"]], dtype=np.uint32)
    pstart = np.array([r["read_start"] for r in j["reads"]], dtype=np.int64)
    pend = np.array([r["read_end"] for r in j["reads"]], dtype=np.int64)
    return j, names, node_len, path_off, path_nodes, j["range_start"], step_off, node_id, pstart, pend


def check_against_literal(j, names, abc, hap, ln, tb, bases, cov, n_abort):
    """abc/hap/ln/tb: a unique-trio table in any row order + its trio bases; compared as keyed sets with the fixture."""
    ex = j["expect"]
    got = {tuple(int(x) for x in abc[i]): (int(ln[i]), names[int(hap[i])], int(tb[i])) for i in range(len(abc))}
    exp = {tuple(t["key"]): (t["len"], t["hap"], t["bases"]) for t in ex["unique_trios"]}
    assert all(t["n_haps"] == 1 for t in ex["unique_trios"])          # a unique trio has exactly one owner (presence row is one-hot)
    assert got == exp
    assert [int(x) for x in bases] == ex["bases_per_node"]
    assert [int(x) for x in cov] == ex["node_base_cov"]
    assert int(n_abort) == ex["n_abort"]


def load_literal_strain_case(k):
    """tests/golden/literal_strain_<k>.json (oracle/gen_golden_literal_strain.py: literal Python reading of the species /
    strain level, LP by SciPy-HiGHS) -> (json, SyntheticSet built from it)."""
    import synthdata as synth
    with open(os.path.join(ROOT, "tests", "golden", "literal_strain_%d.json" % k)) as f:
        j = json.load(f)
    species = []
    for sp in j["species"]:
        names = list(sp["hap_names"])
        assert names == sorted(names)
        path_off = np.zeros(len(names) + 1, dtype=np.uint64)
        path_off[1:] = np.cumsum([len(sp["paths"][n]) for n in names])
        path_nodes = np.concatenate([np.array(sp["paths"][n], dtype=np.uint32) for n in names])
        species.append(synth.SpeciesGraph(sp["name"], np.array(sp["node_len"], dtype=np.int64), path_off, path_nodes, names, sp["range_start"],
                                          sp["range_end"], np.full(len(names), sp["genome_len"]), np.zeros(len(names))))
    rd = j["reads"]
    step_off = np.zeros(len(rd) + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum([len(r["walk"]) for r in rd])
    reads = synth.PackedReads(step_off, np.array([v for r in rd for v in r["walk"]], dtype=np.uint32),
                              np.array([s for r in rd for s in r["strand"]], dtype=np.uint8),
                              np.array([0 if r["read_start"] is None else r["read_start"] for r in rd], dtype=np.int64),   # null read_start: case 3, file seam only
                              np.array([r["read_end"] for r in rd], dtype=np.int64), np.array([r["read_len"] for r in rd], dtype=np.int64),
                              np.array([255 if r["mapq"] is None else r["mapq"] for r in rd], dtype=np.int64),
                              np.array([r.get("read_path_len", 0) for r in rd], dtype=np.int64), [r.get("read_id", "S0R%d/1" % i) for i, r in enumerate(rd)])
    return j, synth.SyntheticSet(species, reads)


def write_literal_gaf(j, path, tags="NM:i:0\tAS:i:150\tdv:f:0\tid:f:1"):
    """the reads of a literal fixture as GAF text: 12 mandatory columns + 4 tags, `*` where the fixture holds a null"""
    with open(path, "w") as f:
        for r in j["reads"]:
            walk = "".join(("<" if s else ">") + str(v) for v, s in zip(r["walk"], r["strand"])) or "*"
            st = "*" if r["read_start"] is None else str(r["read_start"])
            mq = "*" if r["mapq"] is None else str(r["mapq"])
            ql = r["read_len"]
            f.write("%s\t%d\t0\t%d\t+\t%s\t%d\t%s\t%d\t%d\t%d\t%s\t%s\n" % (r["read_id"], ql, ql, walk, r["read_path_len"], st, r["read_end"], ql, ql, mq, tags))


_LIT_FIELDS = [("unique_trio_nodes_fraction", "unique_trio_fraction", 0.0), ("frequencies_mean", "uniq_trio_cov_mean", 1e-9),
               ("path_cov_ratio", "path_base_cov", 2e-6), ("first_sol", "first_sol", 1e-7), ("divergence", "strain_cov_diff", 0.0),
               ("second_sol", "predicted_coverage", 1e-7), ("total_cov_diff", "total_cov_diff", 1e-7)]


def check_metrics_against_literal(exp_metrics, got_dicts, where=""):
    """exp_metrics: the fixture's HapMetrics of one species (hap order); got_dicts: metrics_to_dicts(...) of the same haps.
    Option-ness must agree field by field; rounded fields exactly, LP-derived ones to 1e-7 (unique optima, checked when the
    fixture was generated), the f32 ratio to f32 precision."""
    assert len(exp_metrics) == len(got_dicts)
    for h, (e, g) in enumerate(zip(exp_metrics, got_dicts)):
        for ek, gk, tol in _LIT_FIELDS:
            ev, gv = e[ek], g[gk]
            assert (ev is None) == (gv is None), (where, h, ek, ev, gv)
            if ev is None:
                continue
            if tol == 0.0:
                assert ev == gv, (where, h, ek, ev, gv)
            else:
                assert abs(ev - gv) <= tol * max(1.0, abs(ev)), (where, h, ek, ev, gv)
        assert bool(e["is_rescue"]) == bool(g["is_rescue"]), (where, h, e["is_rescue"], g["is_rescue"])


def oracle_strain_level(sset, sp, keep, absolute, species_idx, threads=8, **strain_kw):
    """The oracle's strain level (trio index, coverage, filters, both LPs, abundace_constraint) for the species `species_idx`, on
    `threads` host threads -> {species index: (metrics dicts per haplotype, candidates, objective 1)}.  TEST checker."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    rd = sset.reads
    first_, order = orc.group_reads(sp, len(sset.species))
    first_ = first_.astype(np.int64)
    order = order.astype(np.int64)

    def one(s):
        g = sset.species[s]
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        sel = np.sort(order[first_[s]:first_[s + 1]])
        so, nid, ps, pe = select_reads(rd, sel)
        b, c, t, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t, **strain_kw)
        assert rc_ == 0
        orc.abundance_constraint(absolute[s], omet)
        return s, (orc.metrics_to_dicts(omet), nc, o1)
    todo = [int(s) for s in species_idx if keep[s]]
    with ThreadPoolExecutor(threads) as ex:
        return dict(ex.map(one, todo))


def oracle_passing_rows(sset, level, sd=0.2, min_cov=0):
    """abundance_est's row filter (profile.rs:3219-3245) over the oracle's strain level -> {species name: {hap name: metrics}}"""
    out = {}
    for s, (d, nc, o1) in level.items():
        g = sset.species[s]
        rows = {}
        for h, m in enumerate(d):
            cov = m["predicted_coverage"]
            if cov is None:
                continue
            if (len(d) > 1 or (m["total_cov_diff"] is not None and m["total_cov_diff"] <= sd)) and cov >= min_cov and cov != 0.0:
                rows[g.hap_names[h]] = m
        out[g.name] = rows
    return out


_STEP_ROW_FIELDS = [(2, "predicted_coverage", 1e-7), (4, "path_base_cov", 2e-6), (5, "unique_trio_fraction", 0.0), (6, "uniq_trio_cov_mean", 1e-9),
                    (7, "first_sol", 1e-7), (8, "strain_cov_diff", 0.0), (9, "total_cov_diff", 1e-7)]


def check_step_rows_against_oracle(strain_rows, expected, l1_tol=1e-4):
    """strain_rows: the step's table (pipeline.finalize_end); expected: oracle_passing_rows(...) of SOME species.  For each of them the
    step must report exactly the oracle's strains, every metric within its tolerance (rounded ones exactly), and the species' relative L1
    of the LP solution and of the predicted coverage within l1_tol (north_star: 1e-4).  -> worst relative L1."""
    got = {}
    for r in strain_rows:
        if r[0] in expected:
            got.setdefault(r[0], {})[r[1]] = r
    worst = 0.0
    for name, rows in expected.items():
        g = got.get(name, {})
        assert set(g) == set(rows), (name, sorted(set(g) ^ set(rows)))
        d1 = n1 = d2 = n2 = 0.0
        for hap, m in rows.items():
            r = g[hap]
            for col, key, tol in _STEP_ROW_FIELDS:
                ev, gv = m[key], r[col]
                assert (ev is None) == (gv is None), (name, hap, key, ev, gv)
                if ev is None:
                    continue
                if tol == 0.0:
                    assert ev == gv, (name, hap, key, ev, gv)
                else:
                    assert abs(ev - gv) <= tol * max(1.0, abs(ev)), (name, hap, key, ev, gv)
            d1 += abs(r[7] - (m["first_sol"] or 0.0)); n1 += abs(m["first_sol"] or 0.0)
            d2 += abs(r[2] - m["predicted_coverage"]); n2 += abs(m["predicted_coverage"])
        worst = max(worst, d1 / n1 if n1 else 0.0, d2 / n2 if n2 else 0.0)
    assert worst <= l1_tol, worst
    return worst


def oracle_tables_parallel(sset, threads=8, fr=0.3, fc=0.46, sr=0.85, sd=0.2, min_ab=1e-4, min_cov=0):
    """Both tables of profile::profile from the oracle, species on `threads` host threads (the full-size configurations) ->
    (species rows [(name, abundance, coverage)], strain rows [(species, hap, coverage, abundance, metrics)]) in table order."""
    from oracle import oracle as orc
    rd = sset.reads
    S = len(sset.species)
    sp = orc.par_bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species], threads)
    counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, counts, sset.avg_len())
    species_rows = sorted([(sset.species[s].name, abundance[s], absolute[s]) for s in range(S) if keep[s]], key=lambda r: -r[1])
    sel = [s for s in range(S) if keep[s] and abundance[s] > min_ab]
    level = oracle_strain_level(sset, sp, keep, absolute, sel, threads=threads, fr=fr, fc=fc, sr=sr)
    passing = oracle_passing_rows(sset, level, sd=sd, min_cov=min_cov)
    # rows enter the final sort in species-table order (load_species_range walks species_abundance.txt, profile.rs:553-656), haplotypes in
    # BTreeMap order; the sort by abundance is stable here and in the library (equal LAD solutions -- ratios of small integers -- do tie;
    # polars leaves the order of ties open)
    idx_of = {g.name: i for i, g in enumerate(sset.species)}
    in_sel = set(sel)
    rows = [(name, hap, passing[name][hap]) for name, _, _ in species_rows if idx_of[name] in in_sel
            for hap in sset.species[idx_of[name]].hap_names if hap in passing[name]]
    tot = sum(r[2]["predicted_coverage"] for r in rows)
    out = [(sp_, hap, m["predicted_coverage"], m["predicted_coverage"] / tot, m) for sp_, hap, m in rows]
    out.sort(key=lambda r: -r[3])
    return species_rows, out, sp
