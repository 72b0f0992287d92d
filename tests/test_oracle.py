"""CPU tests: pin the oracle against the golden vectors (hand-computed micro graphs,
SciPy-HiGHS LP solutions).  These run with -m "not gpu"."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import load_micro_bin, load_micro_cov


def test_micro_trio_index():
    """trio_nodes_info (profile.rs:658-740): windows canonicalised by the w0>w2 swap,
    unique <=> exactly one (hap, position) occurrence."""
    j, names, node_len, path_off, path_nodes, *_ = load_micro_cov()
    g = orc.Graph(node_len, path_off, path_nodes)
    t = orc.TrioTable(g)
    exp = j["expected_trios"]
    assert t.n_unique == len(exp["hap"])
    assert t.abc.tolist() == exp["abc"]
    assert t.hap.tolist() == exp["hap"]
    assert t.len.tolist() == exp["len"]
    assert t.hap_off.tolist() == exp["hap_off"]


def test_micro_coverage():
    """get_node_abundances (profile.rs:743-1026). Per-read derivation (node: +bases [bitmap range]):
    r0 0,1,2 ps2 pe11: n0 +3 [2,5) n1 +3 [0,3) n2 +(9-6)=3 [0,3); trio u0 += 9
    r1 3,2,1 ps1 pe7 : n3 +1 [1,2) n2 +4 [0,4) n1 +(6-5)=1 [0,1); (3,2,1)->(1,2,3) u1 += 6
    r2 4 ps1 pe4     : n4 +3 [1,4)
    r3 4 ps5 pe2     : dropped
    r4 5 ps0 pe3     : n5 +3, bitmap untouched (3 > len 1)
    r5 0,1,0 ps4 pe10: n0 +1 [4,5) n1 +3 [0,3) n0(repeat) aln 6-4=2 [0,2) no bases
    r6 2,3,4 ps0 pe3 : n2 +4 [0,4) n3 +2 [0,2) n4 +max(3-6,0)=0
    r7 4,7 ps3 pe8   : n4 +3 [3,6) n7 +(5-3)=2 [0,2)
    r8 5,3,4 ps0 pe6 : n5 +1 [0,1) n3 +2 [0,2) n4 +3 [0,3); (5,3,4)->(4,3,5) u4 += 6
    r9 0,6,2,3 ps0 pe11: n0 +5 [0,5) n6 +1 n2 +4 n3 +(11-10)=1 [0,1); u5 += 10, (6,2,3)->(3,2,6) u6 += 6
    r10, r11         : reference would abort -> counted, skipped."""
    j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend = load_micro_cov()
    g = orc.Graph(node_len, path_off, path_nodes)
    t = orc.TrioTable(g)
    bases, cov, tb, nab = orc.node_coverage(g, t, rs, step_off, node_id, pstart, pend)
    e = j["expected"]
    assert bases.tolist() == e["bases_per_node"]
    assert cov.tolist() == e["node_base_cov"]
    assert tb.tolist() == e["trio_bases"]
    assert nab == e["n_abort"]


def test_micro_binning_and_counts():
    j, step_off, node_id, qlen, mapq, species, rs, re = load_micro_bin()
    sp = orc.bin_reads(step_off, node_id, rs, re)
    assert sp.tolist() == species.tolist()
    rc, bs, lm, uq = orc.species_counts(sp, qlen, mapq, len(rs))
    e = j["expected_counts"]
    assert rc.tolist() == e["read_count"] and bs.tolist() == e["base_sum"]
    assert lm.tolist() == e["less_multi"] and uq.tolist() == e["uniq_count"]
    # profile.rs:239-245: keep iff uniq_count>0 && less_multi > read_count/10 ; equal-length branch
    keep, absolute, abundance = orc.species_profile(sp, qlen, (rc, bs, lm, uq), np.array([1000.0, 2000.0, 500.0]))
    assert keep.tolist() == [1, 0, 0]
    assert absolute[0] == pytest.approx(2 * 150 / 1000.0) and abundance[0] == 1.0
    keep, absolute, abundance = orc.species_profile(sp, qlen, (rc, bs, lm, uq), np.array([1000.0, 2000.0, 500.0]),
                                                    filtered=False)
    assert keep.tolist() == [1, 1, 1]
    assert abundance.sum() == pytest.approx(1.0)


def test_lad_vs_highs_golden(golden_dir):
    """The PAO LP (profile.rs:1312-1460): objective must equal HiGHS' to 1e-9 relative;
    x must match to L1 <= 1e-6*sum on these (non-degenerate) instances."""
    z = np.load(os.path.join(golden_dir, "lp_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub, xh, objh = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i], z["x_%d" % i], float(z["obj_%d" % i])
        x, obj, it, st = orc.lad_solve(mask, a, len(ub), ub)
        assert st == 0
        assert obj == pytest.approx(objh, rel=1e-9, abs=1e-12), i
        assert orc.lad_objective(mask, a, xh) == pytest.approx(objh, rel=1e-9, abs=1e-12)
        assert np.all(x >= 0) and np.all(x <= ub + 1e-12)
        if bool(z["unique_%d" % i]):  # optimal face is a single point: x itself is comparable
            assert np.abs(x - xh).sum() <= 1e-6 * max(1.0, np.abs(xh).sum()), (i, x, xh)


def test_lad_wide_vs_highs_golden(golden_dir):
    """More than 64 columns (mask rows of several words): the oracle's LAD solver against SciPy-HiGHS on the committed wide
    cases (oracle/gen_golden_wide.py): 65 .. 256 columns, pinned columns, integer ties."""
    z = np.load(os.path.join(golden_dir, "lp_wide_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i]
        p = len(ub)
        x, obj, it, st = orc.lad_solve(mask, a, p, ub)
        assert st == 0
        assert obj == pytest.approx(float(z["obj_%d" % i]), rel=1e-9, abs=1e-12), (i, p)
        assert orc.lad_objective(mask, a, z["x_%d" % i]) == pytest.approx(float(z["obj_%d" % i]), rel=1e-9, abs=1e-12)
        assert np.all(x >= 0) and np.all(x <= ub + 1e-12)


def test_lad_huge_vs_highs_golden(golden_dir):
    """More than 256 columns (the reference's matrix has no column cap, profile.rs:1333-1342): the oracle's LAD solver against
    SciPy-HiGHS on the committed cases of oracle/gen_golden_huge.py, 257 .. 700 columns here (the 1100-column case takes the
    oracle two minutes; the GPU test runs it against the committed optimum)."""
    z = np.load(os.path.join(golden_dir, "lp_huge_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i]
        p = len(ub)
        assert orc.lad_objective(mask, a, z["x_%d" % i]) == pytest.approx(float(z["obj_%d" % i]), rel=1e-9, abs=1e-12)
        if p > 700:
            continue
        x, obj, it, st = orc.lad_solve(mask, a, p, ub)
        assert st == 0
        assert obj == pytest.approx(float(z["obj_%d" % i]), rel=1e-9, abs=1e-12), (i, p)
        assert np.all(x >= 0) and np.all(x <= ub + 1e-12)


def test_lad_degenerate_and_edge_cases():
    # no valid rows -> x = 0
    x, obj, it, st = orc.lad_solve(np.array([1, 3], dtype=np.uint64), np.zeros(2), 2, np.array([1.0, 1.0]))
    assert st == 0 and x.tolist() == [0.0, 0.0] and obj == 0.0
    # all rows identical pattern: x0 + x1 = median(a); objective value is what matters
    a = np.array([1.0, 2.0, 3.0, 4.0, 100.0])
    mask = np.full(5, 3, dtype=np.uint64)
    x, obj, it, st = orc.lad_solve(mask, a, 2, np.array([105.0, 105.0]))
    assert st == 0 and x.sum() == pytest.approx(3.0) and obj == pytest.approx((2 + 1 + 0 + 1 + 97) / 5)
    # upper bound active
    x, obj, it, st = orc.lad_solve(np.array([1, 1, 1], dtype=np.uint64), np.array([5.0, 6.0, 7.0]), 1, np.array([4.0]))
    assert st == 0 and x[0] == pytest.approx(4.0)
    # massive integer ties (depth-like data)
    rng = np.random.default_rng(3)
    n, p = 4000, 5
    mask = rng.integers(1, 1 << p, size=n).astype(np.uint64)
    truth = np.array([4.0, 0.0, 9.0, 0.0, 2.0])
    A = np.stack([((mask >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1)
    a = A @ truth  # noise-free: optimum is exactly truth with objective 0
    x, obj, it, st = orc.lad_solve(mask, a, p, np.full(p, 1.05 * a.max()))
    assert st == 0 and obj == pytest.approx(0.0, abs=1e-9) and np.allclose(x, truth, atol=1e-8)


def test_optimize_species_synthetic():
    """optimize_otu (profile.rs:2884-3026) on a small synthetic species: present strains are
    recovered, absent strains get no predicted coverage."""
    import synthdata as synth
    from tests.helpers import select_reads
    rng = np.random.default_rng(11)
    g = synth.make_species(rng, "1000", 6, 40000, 1, "GCF_000001", present_frac=0.5)
    reads = synth.make_reads(rng, [g], 8000)
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
    T = orc.TrioTable(G)
    bases, cov, tb, nab = orc.node_coverage(G, T, g.range_start, reads.step_off, reads.node_id, reads.pstart, reads.pend)
    assert nab == 0
    rc, met, nc, o1, o2 = orc.optimize_species(G, T, bases, cov, tb)
    assert rc == 0 and nc >= 1
    d = orc.metrics_to_dicts(met)
    present = set(np.nonzero(g.truth_depth > 0)[0].tolist())
    called = {h for h, m in enumerate(d) if m["predicted_coverage"] not in (None, 0.0)}
    assert called == present


def test_gaf_filter_hand_case():
    """filter_max_alignment_mt (gaf_filter.rs:44-97) on lines small enough to decide by eye."""
    from oracle import oracle as orc

    def ln(rid, qs, qe, matches, mapq, ident, extra=""):
        return "\t".join([rid, "20000", str(qs), str(qe), "+", ">1>2", "500", "0", "400", str(matches), "400", str(mapq),
                          "NM:i:0", "AS:f:1", "dv:f:0", "id:f:" + ident]) + extra
    lines = [
        ln("a", 0, 5000, 900, 60, "0.99"),        # 0: best of a -> written
        ln("a", 0, 5000, 800, 60, "1.0"),         # 1: fewer matches
        ln("b", 0, 5000, 900, 60, "0.90"),        # 2: same matches, lower identity
        ln("b", 0, 5000, 900, 20, "0.95"),        # 3: best of b but mapq 20 is not > 20 -> b writes nothing
        ln("c", 0, 1000, 900, 60, "0.99"),        # 4: span 1000 is not > 1000
        ln("d", 0, 1001, 900, 21, "0.99"),        # 5: written
        ln("d", 0, 1001, 900, 21, "0.99"),        # 6: equal best: one line per id, the first
        "e\t1\t2",                                # 7: not a record
        ln("f", 0, 5000, 900, 60, "nan"),         # 8: NaN never equals the best
        ln("g", 0, 5000, 900, 60, "9.9e-1") + "\r",   # 9: written, exponent spelling, CR dropped
        "",                                       # 10
        ln("h", 0, 5000, 900, 60, "0.5", "\tzz:Z:x"),   # 11: extra column, written
    ]
    keep, nrec = orc.gaf_filter("\n".join(lines).encode())
    assert nrec == 10
    assert np.nonzero(keep)[0].tolist() == [0, 5, 9, 11]
    keep2, _ = orc.gaf_filter(("\n".join(lines) + "\n").encode())
    assert keep2.tolist() == keep.tolist()


def test_gaf_filter_generated_properties():
    from oracle import oracle as orc
    from tests.helpers import make_longread_gaf
    txt = make_longread_gaf(3, 3000)
    keep, nrec = orc.gaf_filter(txt)
    lines = txt.split(b"\n")
    if lines and lines[-1] == b"" and txt.endswith(b"\n"):
        lines = lines[:-1]
    assert len(keep) == len(lines) and 0 < keep.sum() < nrec
    ids = [lines[i].strip().split(b"\t")[0] for i in np.nonzero(keep)[0]]
    assert len(ids) == len(set(ids))              # one line per read id
    for i in np.nonzero(keep)[0][:200]:
        f = lines[i].strip().split(b"\t")
        assert int(f[11]) > 20 and int(f[3]) - int(f[2]) > 1000


def test_sampler_and_gaf_filter_fixtures(golden_dir):
    """The committed fixtures of the two load-time operators (generated by oracle/gen_golden_aux.py)."""
    import json
    import os
    from oracle import oracle as orc
    z = json.load(open(os.path.join(golden_dir, "sampler_positions.json")))
    for c in z["cases"]:
        pos = orc.sample_sorted_positions(c["n"], c["amount"], seed=c["seed"])
        assert pos[:8].tolist() == c["first"] and pos[-4:].tolist() == c["last"]
        assert int(pos.astype(np.uint64).sum()) == c["sum"] and int(np.bitwise_xor.reduce(pos.astype(np.uint32))) == c["xor"]
    g = json.load(open(os.path.join(golden_dir, "gaf_filter.json")))
    keep, nrec = orc.gaf_filter(g["text"].encode("latin-1"))
    assert nrec == g["n_records"] and np.nonzero(keep)[0].tolist() == g["kept_lines"]


@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_equals_literal_python_restatement(k):
    """Two independent readings of profile.rs:658-1026 agree: the flat-array C oracle vs fixtures generated from
    oracle/ref_literal.py, which mirrors the reference's maps, sets and per-base byte vectors statement by statement."""
    from oracle import oracle as orc
    from tests.helpers import check_against_literal, load_literal_case
    j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend = load_literal_case(k)
    G = orc.Graph(node_len, path_off, path_nodes)
    T = orc.TrioTable(G)
    # reads with a node outside the species are "U" for the binning and never reach get_node_abundances in the pipeline;
    # the fixture keeps one to pin the panic convention, so the oracle is fed every read here
    b, c, t, na = orc.node_coverage(G, T, rs, step_off, node_id, pstart, pend)
    check_against_literal(j, names, T.abc, T.hap, T.len, t, b, c, na)
    ab = b / node_len
    assert np.allclose(ab, j["expect"]["node_abundance"], rtol=0, atol=0)


def test_lad_solver_vs_reference_milp_model(golden_dir):
    """The reference's actual model -- binary indicators z_j >= (x_j - min_cov) / (2 max a), sum z <= npaths, second solve
    with x_j == 0 rows (profile.rs:1363-1377, 1484-1488) -- solved by scipy.optimize.milp at the BASELINE.md section 2 shape
    (20 000 x 10, 60 % dense) and on integer-tied coverages (oracle/gen_golden_milp.py): the oracle's exact LAD reaches the
    same objective to 1e-9, i.e. the indicators are inert and the LP relaxation is the answer."""
    z = np.load(os.path.join(golden_dir, "lp_milp_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, fixed = z["mask_%d" % i], z["a_%d" % i], z["fixed_%d" % i]
        p = len(fixed)
        assert float(z["obj_milp_%d" % i]) == pytest.approx(float(z["obj_lp_%d" % i]), rel=1e-9)
        ub = np.where(fixed == 1, 0.0, 1.05 * a.max())
        x, obj, it, st = orc.lad_solve(mask, a, p, ub)
        assert st == 0
        assert obj == pytest.approx(float(z["obj_milp_%d" % i]), rel=1e-9, abs=1e-12), str(z["name_%d" % i])
        assert np.all(x[fixed == 1] == 0.0)


@pytest.mark.parametrize("k", [0, 1, 2, 4])
def test_oracle_equals_literal_python_restatement_of_the_strain_level(k):
    """The C oracle against fixtures from oracle/ref_literal_strain.py, the literal Python reading of rcls.rs:237-258 and
    profile.rs:208-349, 1028-1285, 1297-1511 (LP by SciPy-HiGHS), 2884-3070, 3167-3248 that shares no code with it: species of
    every read, the species table, every HapMetrics field of every haplotype (--shift, --filtered off, --min_depth, single-strain
    species, null MAPQ among the cases), the final strain rows."""
    from oracle import oracle as orc
    from tests.helpers import check_metrics_against_literal, load_literal_strain_case, select_reads
    j, sset = load_literal_strain_case(k)
    a, ex, rd = j["args"], j["expect"], sset.reads
    S = len(sset.species)
    names = [g.name for g in sset.species]
    sp = orc.bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species])
    assert [names[i] if i >= 0 else "U" for i in sp] == ex["read_species"]
    counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, counts, sset.avg_len(), filtered=a["filtered"])
    got_tab = sorted([(names[s], abundance[s], absolute[s]) for s in range(S) if keep[s]], key=lambda r: -r[1])
    assert [r[0] for r in got_tab] == [r["species_taxid"] for r in ex["species_profile"]]
    for g_, e_ in zip(got_tab, ex["species_profile"]):
        assert g_[1] == pytest.approx(e_["predicted_abundance"], rel=1e-12) and g_[2] == pytest.approx(e_["predicted_coverage"], rel=1e-12)
    rows = []
    # species in the order of the species table (the strain level iterates the joined frame, profile.rs:600-640): ties of the
    # final sort keep that order
    for s in sorted(range(S), key=lambda i: (-(abundance[i] if keep[i] else -1.0), i)):
        g = sset.species[s]
        if not keep[s] or not abundance[s] > a["min_species_abundance"]:
            assert g.name not in ex["per_species"]
            continue
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(rd, np.nonzero(sp == s)[0])
        b, c, tb, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        e = ex["per_species"][g.name]
        assert (T.n_unique, na, int(b.sum()), int(tb.sum())) == (e["n_unique_trios"], e["n_abort"], e["bases_total"], e["trio_bases_total"])
        # (case 4: --solver highs -- the second solve's solution handed out through highs_opt's slice, profile.rs:2865-2879)
        rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=a["fr"], fc=a["fc"], sr=a["sr"], shift=a["shift"], min_depth=a["min_depth"],
                                                   solver_semantics=1 if a.get("solver") == "highs" else 0)
        assert rc == 0 and nc == e["n_candidates"]
        if e["obj1"] is not None:
            assert o1 == pytest.approx(e["obj1"], rel=1e-9, abs=1e-12)
        if e["obj2"] is not None:
            assert o2 == pytest.approx(e["obj2"], rel=1e-9, abs=1e-12)
        orc.abundance_constraint(absolute[s], met)
        d = orc.metrics_to_dicts(met)
        check_metrics_against_literal(e["metrics"], d, g.name)
        for h, m in enumerate(d):
            cov = m["predicted_coverage"]
            if cov is None:
                continue
            if (len(d) > 1 or (m["total_cov_diff"] is not None and m["total_cov_diff"] <= a["sd"])) and cov >= a["min_cov"] and cov != 0.0:
                rows.append((g.name, g.hap_names[h], cov))
    tot = sum(r[2] for r in rows)
    rows = sorted([(r[0], r[1], r[2], r[2] / tot) for r in rows], key=lambda r: -r[3])
    assert [(r[0], r[1]) for r in rows] == [(e["species_taxid"], e["hap_id"]) for e in ex["final_rows"]]
    for r, e in zip(rows, ex["final_rows"]):
        assert r[2] == pytest.approx(e["predicted_coverage"], rel=1e-7) and r[3] == pytest.approx(e["predicted_abundance"], rel=1e-7)


@pytest.mark.parametrize("seed,n", [(9, 120), (21, 2000), (22, 3000)])
def test_gaf_filter_oracle_vs_literal_python_restatement(seed, n, golden_dir):
    """filter_max_alignment_mt (gaf_filter.rs:21-97): the C oracle's choice against oracle/ref_literal_gaf_filter.py, an
    independent literal reading with Rust's integer / float grammars spelled out.  The reference defines WHICH reads get a line
    and which lines may be it (the rest is rayon scheduling): the oracle keeps exactly one admissible line per such read -- the
    first in file order -- and nothing else; the committed fixture says the same."""
    import json
    import sys
    from oracle import oracle as orc
    from tests.helpers import make_longread_gaf
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir), "..", "oracle"))
    import ref_literal_gaf_filter as lg
    txt = make_longread_gaf(seed, n, path_ids=6)
    keep, nrec = orc.gaf_filter(txt)
    n_rec_lit, cand = lg.candidates(txt.decode("latin-1"))
    assert nrec == n_rec_lit
    kept = np.nonzero(keep)[0].tolist()
    # reads with a NaN identity among their records: whether the NaN becomes the read's "best" (and then nothing equals it)
    # depends on the order the reference's parallel loop visits the records -- undefined by the reference, left out here
    lines = txt.decode("latin-1").split("\n")
    nan_ids = set()
    for l in lines:
        r = lg.parse_line(l[:-1] if l.endswith("\r") else l)
        if r is not None and r["align_16"] != r["align_16"]:
            nan_ids.add(r["read_id"])
    rid_of = lambda i: lg.parse_line(lines[i][:-1] if lines[i].endswith("\r") else lines[i])["read_id"]
    assert [i for i in kept if rid_of(i) not in nan_ids] == sorted(v[0] for k_, v in cand.items() if k_ not in nan_ids)   # one line per read with candidates: its first candidate; file order
    assert len(nan_ids) < len(cand) // 10
    assert len(cand) > n // 4 and (n < 1000 or any(len(v) > 1 for v in cand.values()))   # the generator does produce rejected reads and (at size) equal-best ties
    if (seed, n) == (9, 120):
        g = json.load(open(os.path.join(golden_dir, "gaf_filter.json")))
        assert g["text"].encode("latin-1") == txt and g["kept_lines"] == kept and g["n_records"] == nrec
