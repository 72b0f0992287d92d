"""The two readings of the reference against each other on SEEDED RANDOM small cases (CPU only, no fixtures): the literal Python
restatements (oracle/ref_literal.py, oracle/ref_literal_strain.py: the reference's maps, sets, per-base byte vectors and row
frames statement by statement, LP by SciPy-HiGHS) and the flat-array C oracle (oracle/pantax_oracle.c) that every HIP parity test is
checked with.  The committed fixtures (tests/golden/literal_*.json) are eight such cases; here a few hundred are generated on the spot
-- both levels, --shift, --filtered off, --min_depth, single-strain species, null MAPQ / read_start rows, duplicate read ids -- so that
the pin "two independent readings of profile.rs agree" does not rest on a handful of seeds.  Where an LP optimum is a face rather than a
point the solution vector is a free choice of the solver: such species are compared up to the objective and counted, not skipped
silently."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as _orc_first   # noqa: F401  (the package, before its directory joins the path for the generator modules)
from tests.conftest import ROOT

sys.path.append(os.path.join(ROOT, "oracle"))


def _case_arrays(j):
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
    return names, node_len, path_off, path_nodes, step_off, node_id, pstart, pend


N_COV_CASES = 320


@pytest.mark.parametrize("block", range(8))
def test_coverage_level_literal_reading_equals_c_oracle_on_random_cases(block):
    """profile.rs:658-1026 (trio_nodes_info + get_node_abundances), N_COV_CASES random species: 2-9 haplotypes, 60-220 reads with
    reverse-strand walks, plus the hand-made records for the branches error-free reads never reach (one-node reads, end < start, ranges
    beyond the node, repeated nodes, the last-node clamp, the two panics).  Unique-trio table (keys, owner, length, bases), bases per node,
    covered bases per node, abort count: bit for bit."""
    import gen_golden_literal as ggl
    from oracle import oracle as orc
    from tests.helpers import check_against_literal
    per = N_COV_CASES // 8
    for i in range(block * per, (block + 1) * per):
        rng = np.random.default_rng(90000 + i)
        H = int(rng.integers(2, 10))
        j = ggl.make_case(5000 + i, H, int(rng.integers(1200, 3500)), int(rng.integers(60, 220)), int(rng.integers(1, 5000)))
        names, node_len, path_off, path_nodes, step_off, node_id, pstart, pend = _case_arrays(j)
        G = orc.Graph(node_len, path_off, path_nodes)
        T = orc.TrioTable(G)
        b, c, t, na = orc.node_coverage(G, T, j["range_start"], step_off, node_id, pstart, pend)
        check_against_literal(j, names, T.abc, T.hap, T.len, t, b, c, na)
        assert np.array_equal(b / node_len, np.array(j["expect"]["node_abundance"]))


_BASE = dict(fr=0.3, fc=0.46, sr=0.85, sd=0.2, min_cov=0, min_depth=0, shift=False, filtered=True, min_species_abundance=1e-4)
_PLANS = [   # (S, H, genome_len, reads, args, single_every, present_frac, mapq_null_every, dup_ids, null_start_every)
    (2, 3, 2500, 700, dict(_BASE), 0, 0.6, 0, 0, 0),
    (3, 4, 2500, 1100, dict(_BASE, shift=True, fr=0.4), 3, 0.6, 0, 0, 0),
    (2, 3, 2500, 800, dict(_BASE, filtered=False, min_depth=2, fc=0.3), 2, 0.7, 17, 0, 0),
    (3, 3, 2500, 1000, dict(_BASE), 0, 0.6, 0, 40, 40),
    (2, 5, 3000, 900, dict(_BASE, fr=0.2, sr=0.6), 0, 0.4, 0, 0, 0),
    # --solver highs: highs_opt hands the second solve's solution out through a slice of its first K columns (profile.rs:2865-2879)
    # (every strain present and a tight --fc: most species lose candidates in the second filter, so the slice does cut survivors off)
    (2, 5, 3000, 1500, dict(_BASE, solver="highs", fc=0.05, sr=0.99), 0, 1.0, 0, 0, 0),
    (2, 6, 3000, 2500, dict(_BASE, solver="highs", fc=0.02, sr=0.99, fr=0.1), 0, 1.0, 0, 0, 0),
]
N_STRAIN_CASES = 120


@pytest.mark.parametrize("block", range(6))
def test_strain_level_literal_reading_equals_c_oracle_on_random_cases(block):
    """rcls.rs:237-258, profile.rs:208-349 (species level), :361-463 (duplicate ids, null rows), :1028-1285 (filters), :1297-1511 (model and
    flow of gurobi_opt), :2884-3070, :3167-3248 on N_STRAIN_CASES random small databases under five option sets.  Integers, the species table,
    candidate counts, first-filter metrics and LP objectives always; the LP-derived metrics of a species where both readings land on the same
    optimum (an LP whose optimum is a face leaves the point to the solver: compared up to the objective and counted)."""
    import gen_golden_literal_strain as ggs
    from oracle import oracle as orc
    from tests.helpers import check_metrics_against_literal, select_reads
    import synthdata as synth
    per = N_STRAIN_CASES // 6
    n_species_full = n_species_face = n_highs_differs = n_highs = 0
    for i in range(block * per, (block + 1) * per):
        S, H, gl, nr, args, single_every, pf, mqn, dup_ids, null_start = _PLANS[i % len(_PLANS)]
        j = ggs.make_case(7000 + i, S, H, gl, nr, args, single_every, pf, mqn, dup_ids, null_start)
        ex = j["expect"]
        # the same inputs in the packed layouts (what tests/helpers.load_literal_strain_case does with a fixture file)
        species = []
        for sp_ in j["species"]:
            names = list(sp_["hap_names"])
            path_off = np.zeros(len(names) + 1, dtype=np.uint64)
            path_off[1:] = np.cumsum([len(sp_["paths"][n]) for n in names])
            path_nodes = np.concatenate([np.array(sp_["paths"][n], dtype=np.uint32) for n in names])
            species.append(synth.SpeciesGraph(sp_["name"], np.array(sp_["node_len"], dtype=np.int64), path_off, path_nodes, names, sp_["range_start"], sp_["range_end"],
                                              np.full(len(names), sp_["genome_len"]), np.zeros(len(names))))
        rd = j["reads"]
        step_off = np.zeros(len(rd) + 1, dtype=np.uint64)
        step_off[1:] = np.cumsum([len(r["walk"]) for r in rd])
        node_id = np.array([v for r in rd for v in r["walk"]], dtype=np.uint32)
        pstart = np.array([0 if r["read_start"] is None else r["read_start"] for r in rd], dtype=np.int64)
        pend = np.array([r["read_end"] for r in rd], dtype=np.int64)
        qlen = np.array([r["read_len"] for r in rd], dtype=np.int64)
        mapq = np.array([255 if r["mapq"] is None else r["mapq"] for r in rd], dtype=np.int64)
        Sn = len(species)
        names_sp = [g.name for g in species]
        sp = orc.bin_reads(step_off, node_id, [g.range_start for g in species], [g.range_end for g in species])
        assert [names_sp[k] if k >= 0 else "U" for k in sp] == ex["read_species"], i
        counts = orc.species_counts(sp, qlen, mapq, Sn)
        avg = np.array([g.genome_len.mean() for g in species], dtype=np.float64)
        keep, absolute, abundance = orc.species_profile(sp, qlen, counts, avg, filtered=args["filtered"])
        got_tab = sorted([(names_sp[s], abundance[s], absolute[s]) for s in range(Sn) if keep[s]], key=lambda r: -r[1])
        assert [r[0] for r in got_tab] == [r["species_taxid"] for r in ex["species_profile"]], i
        for g_, e_ in zip(got_tab, ex["species_profile"]):
            assert g_[1] == pytest.approx(e_["predicted_abundance"], rel=1e-12) and g_[2] == pytest.approx(e_["predicted_coverage"], rel=1e-12)
        # strain-level drops (profile.rs:361-437): rows with a null field; ids whose alignments span species
        drop = np.array([r["read_start"] is None for r in rd])
        ids = {}
        for k, r in enumerate(rd):
            if sp[k] >= 0 and not drop[k]:
                ids.setdefault(r["read_id"], set()).add(int(sp[k]))
        mixed = {rid for rid, ss in ids.items() if len(ss) > 1}
        drop |= np.array([r["read_id"] in mixed for r in rd])
        packed = synth.PackedReads(step_off, node_id, np.zeros(len(node_id), dtype=np.uint8), pstart, pend, qlen, mapq, qlen, [])
        for s in range(Sn):
            g = species[s]
            if not keep[s] or not abundance[s] > args["min_species_abundance"]:
                assert g.name not in ex["per_species"], i
                continue
            e = ex["per_species"].get(g.name)
            sel = np.nonzero((sp == s) & ~drop)[0]
            if e is None:                                  # no record of the species reaches the strain level (:3301-3303)
                assert len(sel) == 0, i
                continue
            G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
            T = orc.TrioTable(G)
            so, nid, ps, pe = select_reads(packed, sel)
            b, c, tb, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
            assert (T.n_unique, na, int(b.sum()), int(tb.sum())) == (e["n_unique_trios"], e["n_abort"], e["bases_total"], e["trio_bases_total"]), (i, g.name)
            sem = 1 if args.get("solver") == "highs" else 0
            rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=args["fr"], fc=args["fc"], sr=args["sr"], shift=args["shift"], min_depth=args["min_depth"],
                                                       solver_semantics=sem)
            assert rc == 0 and nc == e["n_candidates"], (i, g.name)
            if sem:   # does the slice matter here?  (a survivor of the second filter behind a candidate that did not survive)
                _, met_g, *_ = orc.optimize_species(G, T, b, c, tb, fr=args["fr"], fc=args["fc"], sr=args["sr"], shift=args["shift"], min_depth=args["min_depth"])
                n_highs += 1
                n_highs_differs += [m_["predicted_coverage"] is None for m_ in orc.metrics_to_dicts(met)] != [m_["predicted_coverage"] is None for m_ in orc.metrics_to_dicts(met_g)]
            if e["obj1"] is not None:
                assert o1 == pytest.approx(e["obj1"], rel=1e-9, abs=1e-12), (i, g.name)
            orc.abundance_constraint(absolute[s], met)
            d = orc.metrics_to_dicts(met)
            em = e["metrics"]
            # the first-filter metrics never depend on the LP's point
            for h, (a_, b_) in enumerate(zip(em, d)):
                assert a_["unique_trio_nodes_fraction"] == b_["unique_trio_fraction"], (i, g.name, h)
                for ek, gk, tol in (("frequencies_mean", "uniq_trio_cov_mean", 1e-9), ("path_cov_ratio", "path_base_cov", 2e-6)):
                    assert (a_[ek] is None) == (b_[gk] is None), (i, g.name, h, ek)
                    if a_[ek] is not None:
                        assert abs(a_[ek] - b_[gk]) <= tol * max(1.0, abs(a_[ek])), (i, g.name, h, ek)
            same_point = all((a_["first_sol"] is None) == (b_["first_sol"] is None) and
                             (a_["first_sol"] is None or abs(a_["first_sol"] - b_["first_sol"]) <= 1e-6 * max(1.0, abs(a_["first_sol"]))) for a_, b_ in zip(em, d))
            # both readings solved a second LP: same optimum VALUE, always -- in the highs plans (every strain present, a tight --fc) only where the first
            # solves landed on the same point: a first LP whose optimum is a face hands the second filter different solutions, hence different second LPs
            if e["obj2"] is not None and o2 is not None and (same_point or not sem):
                assert abs(o2 - e["obj2"]) <= 1e-9 * max(1.0, abs(e["obj2"])), (i, g.name, o2, e["obj2"])
            # (abundace_constraint may have scaled both second solutions onto the species coverage: the sum BEFORE the scaling survives in
            # total_cov_diff, so a second LP solved to another point of its face shows there)
            if same_point:
                same_point = all((a_["total_cov_diff"] is None) == (b_["total_cov_diff"] is None) and
                                 (a_["total_cov_diff"] is None or abs(a_["total_cov_diff"] - b_["total_cov_diff"]) <= 1e-6) for a_, b_ in zip(em, d))
            if same_point and e["obj2"] is not None and o2 is not None:
                same_point = all(
                    (a_["second_sol"] is None) == (b_["predicted_coverage"] is None) and
                    (a_["second_sol"] is None or abs(a_["second_sol"] - b_["predicted_coverage"]) <= 1e-6 * max(1.0, abs(a_["second_sol"]))) for a_, b_ in zip(em, d))
            if same_point:
                check_metrics_against_literal(em, d, "%d %s" % (i, g.name))
                n_species_full += 1
            else:
                n_species_face += 1
    assert n_species_full >= 5 and n_species_full >= n_species_face, (n_species_full, n_species_face)   # most LPs of these sizes have a point optimum
    assert n_highs == 0 or n_highs_differs >= 1, (n_highs, n_highs_differs)   # the highs plans do reach cases where its slice changes the outcome
