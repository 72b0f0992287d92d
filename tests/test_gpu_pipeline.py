"""GPU test of the pipeline seam (pantax_hip_profile == profile::profile, profile.rs:3325): a small
synthetic DB + GAF on disk -> species_abundance.txt / strain_abundance.txt, checked against the
oracle run on the same inputs (numeric comparison; polars' float text is not byte-pinned)."""
import os

import numpy as np
import pytest

from tests.helpers import select_reads

pytestmark = pytest.mark.gpu


def _read_tsv(path):
    with open(path) as f:
        hdr = f.readline().rstrip("\n").split("\t")
        return hdr, [line.rstrip("\n").split("\t") for line in f]


def _oracle_tables(sset, fr=0.3, fc=0.46, sr=0.85, sd=0.2, min_ab=1e-4):
    from oracle import oracle as orc
    rd = sset.reads
    S = len(sset.species)
    sp = orc.bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species])
    counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, counts, sset.avg_len())
    species_rows = sorted([(sset.species[s].name, abundance[s], absolute[s]) for s in range(S) if keep[s]], key=lambda r: -r[1])
    rows = []
    for s, g in enumerate(sset.species):
        if not keep[s] or not abundance[s] > min_ab:
            continue
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(rd, np.nonzero(sp == s)[0])
        b, c, tb, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=fr, fc=fc, sr=sr)
        assert rc == 0
        orc.abundance_constraint(absolute[s], met)
        d = orc.metrics_to_dicts(met)
        for h, m in enumerate(d):
            cov = m["predicted_coverage"]
            if cov is None:
                continue
            if (len(d) > 1 or (m["total_cov_diff"] is not None and m["total_cov_diff"] <= sd)) and cov >= 0 and cov != 0.0:
                rows.append((g.name, g.hap_names[h], m))
    tot = sum(r[2]["predicted_coverage"] for r in rows)
    out = [(sp_, hap, m["predicted_coverage"], m["predicted_coverage"] / tot, m) for sp_, hap, m in rows]
    out.sort(key=lambda r: -r[3])
    return species_rows, out, sp


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from pantax_amd import synth
    from pantax_amd.engine import Engine
    sset = synth.make_set(31, 4, 5, 30000, 30000, present_frac=0.4, single_strain_every=4, with_ids=False)
    root = tmp_path_factory.mktemp("pantax")
    db = root / "db"
    db.mkdir()
    synth.write_db(sset, str(db))
    gaf = root / "gfa_mapped.gaf"
    synth.write_gaf(sset.reads, str(gaf))
    eng = Engine(0)
    yield sset, root, db, gaf, eng
    eng.close()


def _check_outputs(wd, sset, exp_species, exp_strain):
    hdr, rows = _read_tsv(os.path.join(wd, "species_abundance.txt"))
    assert hdr == ["species_taxid", "predicted_abundance", "predicted_coverage"]
    assert [r[0] for r in rows] == [r[0] for r in exp_species]
    for r, e in zip(rows, exp_species):
        assert float(r[1]) == pytest.approx(e[1], rel=1e-12) and float(r[2]) == pytest.approx(e[2], rel=1e-12)
    hdr, rows = _read_tsv(os.path.join(wd, "strain_abundance.txt"))
    assert hdr == ["species_taxid", "strain_taxid", "genome_ID", "predicted_coverage", "predicted_abundance", "path_base_cov",
                   "unique_trio_fraction", "uniq_trio_cov_mean", "first_sol", "strain_cov_diff", "total_cov_diff"]
    assert len(rows) == len(exp_strain) and len(rows) > 0
    for r, (sp_, hap, cov, ab, m) in zip(rows, exp_strain):
        assert r[0] == sp_ and r[2].startswith(hap)
        assert float(r[3]) == pytest.approx(cov, rel=1e-7) and float(r[4]) == pytest.approx(ab, rel=1e-7)
        for col, key in [(5, "path_base_cov"), (6, "unique_trio_fraction"), (7, "uniq_trio_cov_mean"), (8, "first_sol"),
                         (9, "strain_cov_diff"), (10, "total_cov_diff")]:
            if m[key] is None:
                assert r[col] == ""
            else:
                assert float(r[col]) == pytest.approx(m[key], rel=1e-7, abs=1e-9), (key, r)
    assert abs(sum(float(r[4]) for r in rows) - 1.0) < 1e-9


def test_profile_seam_bin_and_gfa(world):
    sset, root, db, gaf, eng = world
    exp_species, exp_strain, sp = _oracle_tables(sset)
    cwd = os.getcwd()
    for name, zip_ in [("wd_bin", "serialize"), ("wd_gfa", None)]:
        wd = root / name
        wd.mkdir()
        os.chdir(str(wd))   # ori_strain_abundance.txt goes to the current directory (profile.rs:3217)
        try:
            eng.profile(str(db), str(wd), str(gaf), zip=zip_, out_binning_file=str(wd / "reads_classification.tsv"))
        finally:
            os.chdir(cwd)
        _check_outputs(str(wd), sset, exp_species, exp_strain)
        assert os.path.exists(wd / "ori_strain_abundance.txt")
        # binning report: read_id, mapq, species, read_len without header, one row per GAF line
        with open(wd / "reads_classification.tsv") as f:
            rep = [l.rstrip("\n").split("\t") for l in f]
        assert len(rep) == sset.reads.n_reads
        names = [g.name for g in sset.species]
        assert [r[2] for r in rep[:200]] == [names[i] if i >= 0 else "U" for i in sp[:200]]


def test_profile_seam_resume_strain_only(world):
    """--strain after an earlier --species run (profile.rs:3365-3417): same strain table."""
    sset, root, db, gaf, eng = world
    exp_species, exp_strain, sp = _oracle_tables(sset)
    wd = root / "wd_resume"
    wd.mkdir()
    cwd = os.getcwd()
    os.chdir(str(wd))
    try:
        eng.profile(str(db), str(wd), str(gaf), species=True, strain=False, out_binning_file=str(wd / "reads_classification.tsv"))
        assert os.path.exists(wd / "species_abundance.txt") and not os.path.exists(wd / "strain_abundance.txt")
        eng.profile(str(db), str(wd), str(gaf), species=False, strain=True)
        # existing outputs short-circuit the run unless --force (profile.rs:136-156)
        before = os.path.getmtime(wd / "strain_abundance.txt")
        eng.profile(str(db), str(wd), str(gaf), species=True, strain=True)
        assert os.path.getmtime(wd / "strain_abundance.txt") == before
    finally:
        os.chdir(cwd)
    _check_outputs(str(wd), sset, exp_species, exp_strain)


def test_profile_seam_errors(world):
    from pantax_amd.engine import PantaxHipError
    sset, root, db, gaf, eng = world
    wd = root / "wd_err"
    wd.mkdir()
    with pytest.raises(PantaxHipError):
        eng.profile(str(db), str(wd), str(root / "missing.gaf"))
    with pytest.raises(PantaxHipError):
        eng.profile(str(db), str(wd), str(gaf), sample_nodes=500000)
    with pytest.raises(PantaxHipError):
        eng.profile(str(db), str(wd), str(gaf), species=False, strain=False)
