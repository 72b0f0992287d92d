"""Every BASELINE.json GPU configuration at FULL size through the HIP path (cfg2 lives in test_gpu_fullsize.py):
  cfg3        100 species x 10 strains, 10 M short reads                -- every species against the oracle
  cfg4        1 000 species x 10 strains, 100 M short reads on ONE GPU (the bench line's set) -- binning of every read against
                                                                           the oracle, properties, oracle on a species sample
  cfg4 share  125 species x 10 strains, 12.5 M short reads (1/8 of cfg4) -- size-independent properties + oracle on a sample
  cfg5 share  125 species x 50 strains, 125 k HiFi-shaped reads (1/8 of cfg5: 6 250 of 50 k strains, total bases = cfg4's)
                                                                        -- properties + oracle on a sample
Properties: conservation of aligned bases (from the read records alone), counters partition the reads, order invariance
under a permutation of the reads, idempotence, cov <= node length, LP local optimality, table normalisation.
The sets come from synth.make_set_mp (species generated on all host cores)."""
import os

import numpy as np
import pytest

from tests.helpers import (check_step_rows_against_oracle, expected_total_bases, oracle_passing_rows, oracle_species_checks, oracle_strain_level,
                           oracle_tables_parallel, select_reads, slice_reads)

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 1, 32)


@pytest.fixture(scope="module")
def eng():
    from pantax_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _bin_oracle(sset, threads):
    """rcls.rs:237-258 on the host for all reads, in `threads` slices"""
    from oracle import oracle as orc
    rd = sset.reads
    return orc.par_bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species], threads)


def _run_stages(eng, sset, fr=0.3):
    from pantax_amd.engine import metrics_to_dicts
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    keep, absolute, abundance = eng.species_profiling((rc, bs, lm, uq), sset.avg_len())
    eng.db_reset()
    abc, hap, ln, hto = eng.trio_nodes_info()
    bases, cov, tb, nab = eng.get_node_abundances()
    met, info = eng.strain_profiling(absolute, species_active=keep, fr=fr)
    return dict(sp=sp, counts=(rc, bs, lm, uq), keep=keep, absolute=absolute, hto=hto.astype(np.int64), bases=bases, cov=cov, tb=tb, nab=nab,
                gmet=metrics_to_dicts(met, eng.H), info=info, trio=(abc, hap, ln))


def _common_properties(sset, out):
    rd = sset.reads
    sp = out["sp"]
    rc, bs, lm, uq = out["counts"]
    assert rc.sum() + (sp < 0).sum() == rd.n_reads
    assert bs.sum() == rd.qlen[sp >= 0].sum()
    assert np.array_equal(rc, np.bincount(sp[sp >= 0], minlength=len(sset.species)))
    assert np.array_equal(uq, np.bincount(sp[(sp >= 0) & (rd.mapq == 60)], minlength=len(sset.species)))
    gl = np.concatenate([g.node_len for g in sset.species])
    assert np.all(out["cov"] <= gl.astype(np.uint64))
    assert int(out["bases"].sum()) == expected_total_bases(sset, sp)


def _lp_local_optimality(eng, sset, out, s, n_probe=40):
    """x of the solver seam minimises the LAD objective over the box: no probe direction lowers it (convexity)."""
    from oracle import oracle as orc
    g = sset.species[s]
    nb = int(np.cumsum([0] + [x.n_nodes for x in sset.species])[s])
    bases = out["bases"][nb:nb + g.n_nodes]
    cov = out["cov"][nb:nb + g.n_nodes]
    ab = bases / g.node_len
    cand = np.arange(g.n_paths)
    x, ratio, obj, st = eng.pao_solve(g.node_len, ab, cov, g.path_off, g.path_nodes, cand)
    assert st == 0
    mask = np.zeros(g.n_nodes, dtype=np.uint64)
    for kk in range(g.n_paths):
        mask[g.path_nodes[int(g.path_off[kk]):int(g.path_off[kk + 1])]] |= np.uint64(1 << kk)
    f0 = orc.lad_objective(mask, ab, x)
    assert f0 == pytest.approx(obj, rel=1e-9)
    ub = 1.05 * ab.max()
    rng = np.random.default_rng(s)
    for trial in range(n_probe):
        d = np.zeros(len(x))
        if trial < 2 * min(len(x), 10):
            d[trial // 2] = 1.0 if trial % 2 == 0 else -1.0
        else:
            d = rng.normal(size=len(x))
        for step in (1e-6, 1e-3, 0.1):
            y = np.clip(x + step * d, 0.0, ub)
            assert orc.lad_objective(mask, ab, y) >= f0 - 1e-9 * max(1.0, f0)


def _tables_normalised(eng, sset):
    from pantax_amd.pipeline import StepConfig, profile_step
    names = [g.name for g in sset.species]
    haps = [h for g in sset.species for h in g.hap_names]
    sp_rows, st_rows, stats = profile_step(eng, names, haps, sset.avg_len(), StepConfig())
    assert sum(r[1] for r in sp_rows) == pytest.approx(1.0, rel=1e-12)
    assert sum(r[3] for r in st_rows) == pytest.approx(1.0, rel=1e-12)
    return sp_rows, st_rows, stats


@pytest.fixture(scope="module")
def cfg3_set():
    import synthdata as synth
    return synth.make_set_mp(20260504, 100, 10, 10_000_000, 5_000_000)


def test_cfg3_full_size_every_species_against_oracle(eng, cfg3_set):
    sset = cfg3_set
    out = _run_stages(eng, sset)
    assert np.array_equal(out["sp"], _bin_oracle(sset, THREADS))
    _common_properties(sset, out)
    bad = oracle_species_checks(sset, out["sp"], out["keep"], out["absolute"], out["bases"], out["cov"], out["tb"], out["hto"], out["gmet"],
                                out["info"], range(len(sset.species)), threads=THREADS)
    assert not bad, bad[:10]
    # the single-call resident step gives normalised tables ...
    sp_rows, st_rows, stats = _tables_normalised(eng, sset)
    assert len(sp_rows) == int(out["keep"].sum())
    # ... whose strain rows are the oracle's for EVERY species: the step path (node_cov_stats_kernel, masks formed in the row sort,
    # objective over the rows, the rows kernel without export copies) against the checker, not only the stage calls above
    level = oracle_strain_level(sset, out["sp"], out["keep"], out["absolute"], range(len(sset.species)), threads=THREADS)
    active = {r[0] for r in sp_rows if r[1] > 1e-4}                    # load_species_range's -a cut (profile.rs:602)
    check_step_rows_against_oracle(st_rows, {k: v for k, v in oracle_passing_rows(sset, level).items() if k in active})


def test_cfg3_file_seam_every_species_against_oracle(eng, cfg3_set, tmp_path_factory):
    """BASELINE configs[2] through the DROP-IN seam (pantax_hip_profile == profile::profile, profile.rs:3325: files in, files out;
    rows a6 + a16 + (b) + f2 at size): 100 `.bin` graphs + 1.4 GB of GAF text on disk -> the two tables, compared with the oracle's
    tables for every species; cold from the bincode files (images written on the way), then from the device-ready images -- same bytes."""
    import synthdata as synth
    from tests.test_gpu_pipeline import _check_outputs
    sset = cfg3_set
    root = tmp_path_factory.mktemp("cfg3_seam")
    db = root / "db"
    db.mkdir()
    synth.write_db(sset, str(db), write_gfa=False, threads=THREADS)
    gaf = root / "gfa_mapped.gaf"
    synth.write_gaf_parallel(sset.reads, str(gaf), threads=THREADS)
    exp_species, exp_strain, _ = oracle_tables_parallel(sset, threads=THREADS)
    cwd = os.getcwd()
    outs = []
    for name, ic in (("wd_cold", 2), ("wd_warm", 1)):
        wd = root / name
        wd.mkdir()
        os.chdir(str(wd))
        try:
            eng.profile(str(db), str(wd), str(gaf), zip="serialize", sample_nodes=0, image_cache=ic)
        finally:
            os.chdir(cwd)
        _check_outputs(str(wd), sset, exp_species, exp_strain)
        outs.append(wd)
    assert len(list((db / "species_graph_info").glob("*.hipdb"))) == len([r for r in exp_species if r[1] > 1e-4])   # one image per selected species
    for f in ("species_abundance.txt", "strain_abundance.txt"):
        assert open(outs[0] / f).read() == open(outs[1] / f).read()
    import shutil
    shutil.rmtree(str(root), ignore_errors=True)


def test_reference_db_shape_every_species_against_oracle(eng, tmp_path_factory):
    """The shape of the database PanTax ships (synthdata.RefDbSet: the strains-per-species histogram of the reference's genomes_info.txt -- 85 % of the
    species hold ONE genome, a chain of 1024-bp chunks, build_eq1.rs:26-36; the others 2 .. 10 strains) at a twentieth of its species count: ~440
    species, 1.2 M reads.  H = 1 species (no trio table, the single-path branches of the filters, profile.rs:1191-1224, :1269-1278), thousands-of-nodes
    graphs beside 3e5-node ones, many tiny LPs.  Through the FILE seam (graphs in groups, image cache on the second run) and through the resident
    step: both tables against the oracle for every species."""
    import synthdata as synth
    from pantax_amd.pipeline import StepConfig, profile_step
    from tests.test_gpu_pipeline import _check_outputs
    ns = synth.RefDbSet(20260507, 1_200_000, scale=0.05, threads=THREADS)
    sset = ns.make()
    assert sum(1 for g in sset.species if g.n_paths == 1) > 300 and max(g.n_paths for g in sset.species) == 10
    exp_species, exp_strain, _ = oracle_tables_parallel(sset, threads=THREADS)
    singles = {g.name for g in sset.species if g.n_paths == 1}
    assert sum(1 for r in exp_strain if r[0] in singles) >= 20 and sum(1 for r in exp_strain if r[0] not in singles) >= 10   # both kinds reach the table
    # resident step
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    names = [g.name for g in sset.species]
    haps = [h for g in sset.species for h in g.hap_names]
    sp_rows, st_rows, _ = profile_step(eng, names, haps, sset.avg_len(), StepConfig())
    # (long nodes on average: the node statistics take their covered-base counts from a per-stretch prefix in LDS; round 5's per-lane word loop gives the same tables)
    eng.set_option("ncs_no_prefix", "1")
    try:
        sp_rows_b, st_rows_b, _ = profile_step(eng, names, haps, sset.avg_len(), StepConfig())
    finally:
        eng.set_option("ncs_no_prefix", None)
    assert sp_rows_b == sp_rows and st_rows_b == st_rows
    # (four fifths of the species are absent from the sample: the statistics and histogram passes of the step do not read them -- same tables when they do)
    eng.set_option("no_absent_skip", "1")
    try:
        sp_rows_c, st_rows_c, _ = profile_step(eng, names, haps, sset.avg_len(), StepConfig())
    finally:
        eng.set_option("no_absent_skip", None)
    assert sp_rows_c == sp_rows and st_rows_c == st_rows
    # the STAGE call's covered bases (popcount pass -- what the file seam runs; long nodes: popcount_long_kernel's per-stretch prefix) against the plain
    # per-thread loop, and against the oracle for a sample of the species (every twentieth: single-genome chains and pangenome graphs)
    from oracle import oracle as orc
    from tests.helpers import select_reads
    sp_of_read = eng.rcls_profile()[0]
    eng.trio_nodes_info(fetch=False)
    bases_a, cov_a, _, _ = eng.get_node_abundances()
    eng.set_option("ncs_no_prefix", "1")
    try:
        bases_b, cov_b, _, _ = eng.get_node_abundances()
    finally:
        eng.set_option("ncs_no_prefix", None)
    assert np.array_equal(cov_a, cov_b) and np.array_equal(bases_a, bases_b) and int(cov_a.sum()) > 0
    checked = 0
    for si in range(0, len(sset.species), 20):
        g = sset.species[si]
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        so, nid, ps, pe = select_reads(sset.reads, np.nonzero(sp_of_read == si)[0])
        b_ref, c_ref = orc.node_coverage(G, orc.TrioTable(G), g.range_start, so, nid, ps, pe)[:2]
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        assert np.array_equal(cov_a[lo:hi], c_ref) and np.array_equal(bases_a[lo:hi], b_ref)
        checked += int(c_ref.sum() > 0)
    assert checked >= 5
    assert [r[0] for r in sp_rows] == [r[0] for r in exp_species]
    for r, e in zip(sp_rows, exp_species):
        assert r[1] == pytest.approx(e[1], rel=1e-12) and r[2] == pytest.approx(e[2], rel=1e-12)
    assert sorted((r[0], r[1]) for r in st_rows) == sorted((r[0], r[1]) for r in exp_strain)
    exp = {(r[0], r[1]): r for r in exp_strain}
    for r in st_rows:
        e = exp[(r[0], r[1])]
        assert r[2] == pytest.approx(e[2], rel=1e-7) and r[3] == pytest.approx(e[3], rel=1e-7)
    eng.release()
    # file seam, graphs in several groups
    root = tmp_path_factory.mktemp("refdb_seam")
    db = root / "db"
    db.mkdir()
    synth.write_db(sset, str(db), write_gfa=False, threads=THREADS)
    gaf = root / "gfa_mapped.gaf"
    synth.write_gaf_parallel(sset.reads, str(gaf), threads=THREADS)
    cwd = os.getcwd()
    eng.set_option("db_groups", "3")
    try:
        for name, ic in (("wd_cold", 2), ("wd_warm", 1)):
            wd = root / name
            wd.mkdir()
            os.chdir(str(wd))
            try:
                eng.profile(str(db), str(wd), str(gaf), zip="serialize", sample_nodes=0, image_cache=ic)
            finally:
                os.chdir(cwd)
            _check_outputs(str(wd), sset, exp_species, exp_strain)
    finally:
        eng.set_option("db_groups", None)
    import shutil
    shutil.rmtree(str(root), ignore_errors=True)


def test_cfg4_share_full_size_properties_and_oracle_sample(eng):
    import synthdata as synth
    sset = synth.make_set_mp(20260505, 125, 10, 12_500_000, 5_000_000)
    rd = sset.reads
    out = _run_stages(eng, sset)
    assert np.array_equal(out["sp"], _bin_oracle(sset, THREADS))
    _common_properties(sset, out)
    # idempotence
    b2, c2, t2, n2 = eng.get_node_abundances()
    assert np.array_equal(b2, out["bases"]) and np.array_equal(c2, out["cov"]) and np.array_equal(t2, out["tb"]) and n2 == out["nab"]
    # oracle on a species sample (incl. the one with the most and the fewest reads)
    cnt = out["counts"][0]
    sample = sorted({int(np.argmax(cnt)), int(np.argmin(cnt)), 0, 31, 62, 93, 124})
    bad = oracle_species_checks(sset, out["sp"], out["keep"], out["absolute"], out["bases"], out["cov"], out["tb"], out["hto"], out["gmet"],
                                out["info"], sample, threads=THREADS)
    assert not bad, bad[:10]
    _lp_local_optimality(eng, sset, out, sample[1])
    # order invariance: the same reads in another order give the same integers
    perm = np.random.default_rng(4).permutation(rd.n_reads)
    so, nid, ps, pe = select_reads(rd, perm)
    eng.upload_reads(so, nid, ps, pe, rd.qlen[perm], rd.mapq[perm])
    sp_p, rc, bs, lm, uq = eng.rcls_profile()
    assert np.array_equal(sp_p, out["sp"][perm])
    for a, b in zip(out["counts"], (rc, bs, lm, uq)):
        assert np.array_equal(a, b)
    b_p, c_p, t_p, n_p = eng.get_node_abundances()
    assert np.array_equal(b_p, out["bases"]) and np.array_equal(c_p, out["cov"]) and np.array_equal(t_p, out["tb"]) and n_p == out["nab"]
    eng.upload_packed(rd)
    _tables_normalised(eng, sset)


def test_cfg5_share_full_size_long_reads_properties_and_oracle_sample(eng):
    import synthdata as synth
    sset = synth.make_set_mp(20260506, 125, 50, 125_000, 5_000_000, long_reads=True)
    rd = sset.reads
    assert (np.diff(rd.step_off.astype(np.int64)) > 64).mean() > 0.9          # walks longer than a wave: the long-read path
    out = _run_stages(eng, sset, fr=0.5)                                        # long reads: --fr 0.5 (main.rs:108-114)
    assert np.array_equal(out["sp"], _bin_oracle(sset, THREADS))
    _common_properties(sset, out)
    cnt = out["counts"][0]
    sample = sorted({int(np.argmax(cnt)), int(np.argmin(cnt)), 0, 41, 83, 124})
    bad = oracle_species_checks(sset, out["sp"], out["keep"], out["absolute"], out["bases"], out["cov"], out["tb"], out["hto"], out["gmet"],
                                out["info"], sample, threads=THREADS, strain_kw=dict(fr=0.5))
    assert not bad, bad[:10]
    _lp_local_optimality(eng, sset, out, sample[0], n_probe=24)                 # a 50-column LP through the solver seam
    perm = np.random.default_rng(5).permutation(rd.n_reads)
    so, nid, ps, pe = select_reads(rd, perm)
    eng.upload_reads(so, nid, ps, pe, rd.qlen[perm], rd.mapq[perm])
    sp_p, *_ = eng.rcls_profile()
    assert np.array_equal(sp_p, out["sp"][perm])
    b_p, c_p, t_p, n_p = eng.get_node_abundances()
    assert np.array_equal(b_p, out["bases"]) and np.array_equal(c_p, out["cov"]) and np.array_equal(t_p, out["tb"]) and n_p == out["nab"]


def test_cfg4_full_size_one_gpu_properties_and_oracle_sample(eng):
    """BASELINE.json configs[3] -- the configuration the metric is quoted on, and bench.py's default set -- on ONE GPU at full size:
    1 000 species / 10 000 strains / 100 M short reads (V = 3.2e8, P = 2.2e9, T = 7.6e8; the loop of profile.rs:3297-3319 at
    that size)."""
    from concurrent.futures import ThreadPoolExecutor
    from bench import native_set, workload_spec
    spec = workload_spec("cfg4")
    sset = native_set(spec, threads=THREADS).make()
    rd = sset.reads
    S = len(sset.species)
    assert (S, rd.n_reads) == (1000, 100_000_000)
    out = _run_stages(eng, sset)
    sp = out["sp"]
    # binning of every read against the oracle
    assert np.array_equal(sp, _bin_oracle(sset, THREADS))
    # counters partition the reads; coverage never exceeds a node; conservation of aligned bases from the read records alone
    rc, bs, lm, uq = out["counts"]
    assert rc.sum() + (sp < 0).sum() == rd.n_reads
    assert np.array_equal(rc, np.bincount(sp[sp >= 0], minlength=S))
    assert bs.sum() == rd.qlen[sp >= 0].sum()
    assert np.array_equal(uq, np.bincount(sp[(sp >= 0) & (rd.mapq == 60)], minlength=S))
    gl = np.concatenate([g.node_len for g in sset.species])
    assert np.all(out["cov"] <= gl.astype(np.uint64))
    del gl
    cuts = np.linspace(0, rd.n_reads, 33).astype(np.int64)

    def part(i):
        import synthdata as synth
        a, b = int(cuts[i]), int(cuts[i + 1])
        return expected_total_bases(synth.SyntheticSet(sset.species, slice_reads(rd, a, b)), sp[a:b])
    with ThreadPoolExecutor(8) as ex:
        assert int(out["bases"].sum()) == sum(ex.map(part, range(32)))
    # idempotence
    b2, c2, t2, n2 = eng.get_node_abundances()
    assert np.array_equal(b2, out["bases"]) and np.array_equal(c2, out["cov"]) and np.array_equal(t2, out["tb"]) and n2 == out["nab"]
    del b2, c2, t2
    # oracle on a species sample (incl. the ones with the most and the fewest reads): integers bit for bit, LP objective 1e-9, metrics
    cnt = out["counts"][0]
    sample = sorted({int(np.argmax(cnt)), int(np.argmin(cnt)), 999} | set(range(0, 1000, 21)))      # fifty species (nine until round 6)
    bad = oracle_species_checks(sset, sp, out["keep"], out["absolute"], out["bases"], out["cov"], out["tb"], out["hto"], out["gmet"],
                                out["info"], sample, threads=THREADS)
    assert not bad, bad[:10]
    # order invariance: the same reads in another order (blocks of 4096 reads shuffled, every block reversed) give the same integers
    nb = (rd.n_reads + 4095) // 4096
    blocks = np.random.default_rng(44).permutation(nb)
    perm = (blocks[:, None] * 4096 + np.arange(4095, -1, -1)[None, :]).ravel()
    perm = perm[perm < rd.n_reads]
    so, nid, ps, pe = select_reads(rd, perm)
    eng.upload_reads(so, nid, ps, pe, rd.qlen[perm], rd.mapq[perm])
    del so, nid
    sp_p, rc_p, bs_p, lm_p, uq_p = eng.rcls_profile()
    assert np.array_equal(sp_p, sp[perm])
    for a, b in zip(out["counts"], (rc_p, bs_p, lm_p, uq_p)):
        assert np.array_equal(a, b)
    b_p, c_p, t_p, n_p = eng.get_node_abundances()
    assert np.array_equal(b_p, out["bases"]) and np.array_equal(c_p, out["cov"]) and np.array_equal(t_p, out["tb"]) and n_p == out["nab"]
    del b_p, c_p, t_p
    # the single-call resident step: normalised tables over the kept species
    eng.upload_packed(rd)
    sp_rows, st_rows, stats = _tables_normalised(eng, sset)
    assert len(sp_rows) == int(out["keep"].sum())
    # ... and the STEP's strain rows at the bench's size against the oracle for the sampled species (what bench.py's abundance-L1 leg
    # checks after its timed region, here inside the suite): integers of the metrics exactly, LP-derived ones to 1e-7, L1 <= 1e-4
    level = oracle_strain_level(sset, sp, out["keep"], out["absolute"], sample, threads=THREADS)
    active = {r[0] for r in sp_rows if r[1] > 1e-4}                    # load_species_range's -a cut (profile.rs:602)
    expected = {k: v for k, v in oracle_passing_rows(sset, level).items() if k in active}
    assert len(expected) >= 5
    check_step_rows_against_oracle(st_rows, expected)


def test_cfg5_full_size_one_gpu_as_four_dbs(eng):
    """BASELINE.json configs[4] at ITS size on ONE GPU: 1 000 species x 50 strains (50 000 strains, 1.1e10 path steps), 1e6 HiFi-shaped reads of
    ~680 steps.  One resident db addresses 2^32 path steps, so the species are cut into four dbs that share the GPU, each with the reads of its
    species, stepped side by side and finalised like ranks (pipeline.profile_steps_many; species are independent from a4 on, profile.rs:3297-3319).
    Checked: the tables are normalised over ALL dbs; the strain rows of a species sample (the species with the most reads, one of every db)
    equal the oracle's -- metrics field by field, abundance L1 <= 1e-4; the coverage integers of two species bit for bit; the second sample
    through the same dbs gives the same tables."""
    from bench import native_set, workload_spec
    from oracle import oracle as orc
    from pantax_amd.engine import Engine
    from pantax_amd.pipeline import StepConfig, profile_steps_many, split_species_by_path_steps
    eng.release()                                    # the module's engine still holds the previous test's 100 GB
    spec = workload_spec("cfg5")
    ns = native_set(spec, threads=THREADS)
    rd = ns.reads()
    species = ns.graphs()
    import synthdata as synth
    sset = synth.SyntheticSet(species, rd)
    S = len(species)
    assert (S, rd.n_reads) == (1000, 1_000_000) and int(ns.P.sum()) > 2 ** 32
    groups = split_species_by_path_steps(ns.P)
    assert len(groups) >= 3
    so = rd.step_off.astype(np.int64)
    klen = np.diff(so)
    first = np.where(klen > 0, rd.node_id[np.minimum(so[:-1], len(rd.node_id) - 1)].astype(np.int64), 0)
    sp_of = np.clip(np.searchsorted(ns.range_start, first, side="right") - 1, 0, S - 1)
    mapq = np.where((rd.mapq < 0) | (rd.mapq > 254), 255, rd.mapq)
    cfg = StepConfig(fr=0.5)                         # long reads: --fr 0.5 (main.rs:108-114)
    engs, names_l, haps_l, avg_l, sels = [], [], [], [], []
    try:
        for gi, (a, b) in enumerate(groups):
            sel = np.nonzero((sp_of >= a) & (sp_of < b) & ((klen > 0) | (gi == 0)))[0]
            so_k, nid_k, ps_k, pe_k = select_reads(rd, sel)
            e = Engine(0)
            e.upload_db(species[a:b])
            e.upload_reads(so_k, nid_k, ps_k, pe_k, rd.qlen[sel], mapq[sel])
            engs.append(e); sels.append(sel)
            names_l.append([g.name for g in species[a:b]])
            haps_l.append([hn for g in species[a:b] for hn in g.hap_names])
            avg_l.append(ns.avg_len()[a:b])
        outs = profile_steps_many(engs, names_l, haps_l, avg_l, 2, cfg)
        sp_rows, st_rows, stats = outs[0]
        assert (outs[1][0], outs[1][1]) == (sp_rows, st_rows)
        assert sum(r[1] for r in sp_rows) == pytest.approx(1.0, rel=1e-12) and sum(r[3] for r in st_rows) == pytest.approx(1.0, rel=1e-12)
        assert {r[0] for r in st_rows} <= {r[0] for r in sp_rows} and len({r[0] for r in st_rows}) > 500
        # the oracle on a species sample: binning of every read, the species table, then the strain level of the sample
        sp = orc.par_bin_reads(rd.step_off, rd.node_id, ns.range_start, ns.range_end, THREADS)
        counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
        keep, absolute, abundance = orc.species_profile(sp, rd.qlen, counts, ns.avg_len())
        exp_sp = sorted([(species[s].name, abundance[s], absolute[s]) for s in range(S) if keep[s]], key=lambda r: -r[1])
        assert [r[0] for r in sp_rows] == [r[0] for r in exp_sp]
        for g_, e_ in zip(sp_rows, exp_sp):
            assert g_[1] == pytest.approx(e_[1], rel=1e-12) and g_[2] == pytest.approx(e_[2], rel=1e-12)
        sample = sorted({int(np.argmax(counts[0]))} | {a + (b - a) // 3 for a, b in groups})
        level = oracle_strain_level(sset, sp, keep, absolute, sample, threads=THREADS, fr=0.5)
        active = {r[0] for r in sp_rows if r[1] > 1e-4}
        expected = {k: v for k, v in oracle_passing_rows(sset, level).items() if k in active}
        assert len(expected) >= 3
        check_step_rows_against_oracle(st_rows, expected)
        # coverage integers of two species of the second db, through its stage calls, against the oracle
        a, b = groups[1]
        e1 = engs[1]
        e1.rcls_profile(want_species=False)
        e1.db_reset()
        abc, hap, ln, hto = e1.trio_nodes_info()
        bases, cov, tb, nab = e1.get_node_abundances()
        nb = np.cumsum([0] + [g.n_nodes for g in species[a:b]])
        hb = np.cumsum([0] + [g.n_paths for g in species[a:b]])
        for s in (a + 1, b - 2):
            g = species[s]
            G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
            T = orc.TrioTable(G)
            so_s, nid_s, ps_s, pe_s = select_reads(rd, np.nonzero(sp == s)[0])
            ob, oc, ot, _ = orc.node_coverage(G, T, g.range_start, so_s, nid_s, ps_s, pe_s)
            k = s - a
            assert np.array_equal(bases[nb[k]:nb[k + 1]], ob) and np.array_equal(cov[nb[k]:nb[k + 1]], oc)
            u0, u1 = int(hto[hb[k]]), int(hto[hb[k + 1]])
            assert u1 - u0 == T.n_unique and np.array_equal(abc[u0:u1], T.abc) and np.array_equal(tb[u0:u1], ot)
    finally:
        for e in engs:
            e.close()
