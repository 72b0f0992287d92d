"""GPU tests at BASELINE.json sizes (cfg2: 1 species x 10 strains, 1M reads, 5 Mbp) and a multi-species set:
direct oracle comparison where the oracle finishes in seconds, plus size-independent properties of the
path (conservation of aligned bases, order invariance, linearity over read sets, idempotence, local
optimality of the LP solution, normalisation)."""
import numpy as np
import pytest

from tests.helpers import select_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pantax_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def cfg2():
    import synthdata as synth
    return synth.make_set(20260503, 1, 10, 1_000_000, 5_000_000)


def _expected_total_bases(sset, sp):
    """Sum of bases_per_node over all nodes, from the read records alone (numpy, no kernel logic):
    one-node read -> target (dropped if < 0); otherwise seen = (len0 - ps) + interior lengths and the read
    contributes seen + max(target - seen, 0) -- minus the lengths of repeated node occurrences, which the
    generator only produces in its adversarial records (handled read by read)."""
    rd = sset.reads
    so = rd.step_off.astype(np.int64)
    k = so[1:] - so[:-1]
    gl = np.concatenate([g.node_len for g in sset.species])
    base = np.cumsum([0] + [g.n_nodes for g in sset.species])[:-1]
    first = np.array([g.range_start for g in sset.species])
    live = (sp >= 0) & (k > 0)
    spc = np.where(sp >= 0, sp, 0)
    step_read = np.repeat(np.arange(len(k)), k)
    v = base[spc[step_read]] + (rd.node_id.astype(np.int64) - first[spc[step_read]])
    v = np.where(live[step_read], v, 0)
    ln = gl[v]
    tot_len = np.zeros(len(k), dtype=np.int64)
    np.add.at(tot_len, step_read, ln)
    len0 = np.zeros(len(k), dtype=np.int64)
    len0[k > 0] = ln[so[:-1][k > 0]]
    lenl = np.zeros(len(k), dtype=np.int64)
    lenl[k > 0] = ln[so[1:][k > 0] - 1]
    target = rd.pend - rd.pstart
    seen = tot_len - lenl - rd.pstart           # (len0 - ps) + interior, for k >= 2
    multi = np.where(target > seen, target, seen)
    single = np.where(target >= 0, target, 0)
    abort = (k >= 2) & (rd.pstart > len0)
    per_read = np.where(k == 1, single, multi)
    per_read = np.where(live & ~abort, per_read, 0)
    # repeated nodes: subtract the aligned length of every non-first occurrence (rare; loop)
    rep_fix = 0
    order = np.lexsort((v, step_read))
    same = (step_read[order][1:] == step_read[order][:-1]) & (v[order][1:] == v[order][:-1])
    for r in np.unique(step_read[order][1:][same]):
        if not live[r] or abort[r] or k[r] < 2:
            continue
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


def test_cfg2_against_oracle_and_properties(eng, cfg2):
    from oracle import oracle as orc
    sset = cfg2
    rd = sset.reads
    g = sset.species[0]
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    # binning: every read is either "U" or counted once; base_sum is the qlen sum of the binned reads
    assert rc.sum() + (sp < 0).sum() == rd.n_reads
    assert bs.sum() == rd.qlen[sp >= 0].sum()
    ref_sp = orc.bin_reads(rd.step_off, rd.node_id, [g.range_start], [g.range_end])
    assert np.array_equal(sp, ref_sp)
    abc, hap, ln, hto = eng.trio_nodes_info()
    bases, cov, tb, nab = eng.get_node_abundances()
    # full-size oracle (finishes in < 2 s): bit-exact integers
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
    T = orc.TrioTable(G)
    so, nid, ps, pe = select_reads(rd, np.nonzero(sp == 0)[0])
    b, c, t, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
    assert np.array_equal(bases, b) and np.array_equal(cov, c) and np.array_equal(tb, t) and nab == na
    assert np.array_equal(abc, T.abc) and np.array_equal(ln, T.len)
    # conservation: total aligned bases from the read records alone
    assert int(bases.sum()) == _expected_total_bases(sset, sp)
    assert np.all(cov <= g.node_len.astype(np.uint64))
    # idempotence
    bases2, cov2, tb2, _ = eng.get_node_abundances()
    assert np.array_equal(bases, bases2) and np.array_equal(cov, cov2) and np.array_equal(tb, tb2)
    # strain level vs oracle at full size
    keep, absolute, abundance = eng.species_profiling((rc, bs, lm, uq), sset.avg_len())
    met, info = eng.strain_profiling(absolute, species_active=keep)
    rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
    orc.abundance_constraint(absolute[0], omet)
    assert info[0].n_candidates == nc and info[0].status1 == 0
    assert info[0].obj1 == pytest.approx(o1, rel=1e-9)
    from pantax_amd.engine import metrics_to_dicts
    for gm, em in zip(metrics_to_dicts(met, eng.H), orc.metrics_to_dicts(omet)):
        for key, ev in em.items():
            if ev is None or isinstance(ev, bool):
                assert gm[key] == ev
            else:
                assert gm[key] == pytest.approx(ev, rel=1e-7, abs=1e-9), key


def test_cfg2_lp_local_optimality(eng, cfg2):
    """The solver's x is a minimiser of the LAD objective: no coordinate move inside the box lowers it
    (convex objective => coordinate-wise + random-direction probes from the optimum cannot go down)."""
    from oracle import oracle as orc
    sset = cfg2
    g = sset.species[0]
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    eng.rcls_profile(want_species=False)
    eng.trio_nodes_info(fetch=False)
    bases, cov, tb, _ = eng.get_node_abundances()
    ab = bases / g.node_len
    cand = np.arange(g.n_paths)   # all ten strains as candidates: a 10-variable LP over ~2.4e5 rows
    x, ratio, obj, st = eng.pao_solve(g.node_len, ab, cov, g.path_off, g.path_nodes, cand)
    assert st == 0
    mask = np.zeros(g.n_nodes, dtype=np.uint64)
    for kk in range(g.n_paths):
        mask[g.path_nodes[int(g.path_off[kk]):int(g.path_off[kk + 1])]] |= np.uint64(1 << kk)
    f0 = orc.lad_objective(mask, ab, x)
    assert f0 == pytest.approx(obj, rel=1e-9)
    ub = 1.05 * ab.max()
    rng = np.random.default_rng(0)
    for trial in range(60):
        d = np.zeros(len(x))
        if trial < 2 * len(x):
            d[trial // 2] = 1.0 if trial % 2 == 0 else -1.0
        else:
            d = rng.normal(size=len(x))
        for step in (1e-6, 1e-3, 0.1):
            y = np.clip(x + step * d, 0.0, ub)
            assert orc.lad_objective(mask, ab, y) >= f0 - 1e-9 * max(1.0, f0)
    # the oracle's exact solver agrees on the objective
    xo, objo, it, sto = orc.lad_solve(mask, ab, len(x), np.full(len(x), ub))
    assert sto == 0 and obj == pytest.approx(objo, rel=1e-9)


def test_multispecies_order_invariance_and_linearity(eng):
    """20 species / 1M reads: the integer outputs do not depend on read order (atomics, locus grouping) and
    are additive over a split of the read set; the bitmap count is monotone and sub-additive."""
    import synthdata as synth
    sset = synth.make_set(77, 20, 8, 1_000_000, 400_000, single_strain_every=7)
    rd = sset.reads
    eng.upload_db(sset.species)

    def run(sel):
        so, nid, ps, pe = select_reads(rd, sel)
        eng.upload_reads(so, nid, ps, pe, rd.qlen[sel], rd.mapq[sel])
        sp, rc, bs, lm, uq = eng.rcls_profile()
        out = eng.get_node_abundances()
        return sp, (rc, bs, lm, uq), out

    eng.trio_nodes_info(fetch=False)
    allr = np.arange(rd.n_reads)
    sp, cnt, (b_all, c_all, t_all, n_all) = run(allr)
    perm = np.random.default_rng(1).permutation(rd.n_reads)
    sp_p, cnt_p, (b_p, c_p, t_p, n_p) = run(perm)
    assert np.array_equal(sp_p, sp[perm])
    for a, b in zip(cnt, cnt_p):
        assert np.array_equal(a, b)
    assert np.array_equal(b_all, b_p) and np.array_equal(c_all, c_p) and np.array_equal(t_all, t_p) and n_all == n_p
    half = rd.n_reads // 2
    _, cnt1, (b1, c1, t1, n1) = run(allr[:half])
    _, cnt2, (b2, c2, t2, n2) = run(allr[half:])
    assert np.array_equal(b1 + b2, b_all) and np.array_equal(t1 + t2, t_all) and n1 + n2 == n_all
    for a, x, y in zip(cnt, cnt1, cnt2):
        assert np.array_equal(a, x + y)
    assert np.all(c_all >= np.maximum(c1, c2)) and np.all(c_all <= c1 + c2)
    assert int(b_all.sum()) == _expected_total_bases(sset, sp)


def test_step_normalisation(eng, cfg2):
    from pantax_amd.pipeline import StepConfig, profile_step
    sset = cfg2
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    names = [g.name for g in sset.species]
    haps = [h for g in sset.species for h in g.hap_names]
    sp_rows, st_rows, stats = profile_step(eng, names, haps, sset.avg_len(), StepConfig())
    assert sum(r[1] for r in sp_rows) == pytest.approx(1.0, rel=1e-12)
    assert sum(r[3] for r in st_rows) == pytest.approx(1.0, rel=1e-12)
    present = {sset.species[0].hap_names[h] for h in np.nonzero(sset.species[0].truth_depth > 0)[0]}
    assert {r[1] for r in st_rows} == present


@pytest.mark.gpu
@pytest.mark.parametrize("seed,S,H,R,L,filtered", [(31, 4, 5, 40000, 120000, True), (32, 6, 3, 30000, 90000, False), (33, 1, 8, 20000, 150000, True)])
def test_single_call_step_equals_stage_by_stage(eng, seed, S, H, R, L, filtered):
    """pantax_hip_profile_step (device-side species decision, one host wait) == the same stages called one
    by one with the species decision taken on the host: identical tables, bit for bit."""
    import synthdata as synth
    from pantax_amd.pipeline import StepConfig, profile_step
    sset = synth.make_set(seed, S, H, R, L)
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    names = [g.name for g in sset.species]
    haps = [h for g in sset.species for h in g.hap_names]
    avg = sset.avg_len()
    if S > 2:
        avg = np.array(avg, dtype=np.float64); avg[1] = 0.0   # a species without a genome length is dropped (profile.rs:329)
    for rebuild in (True, False):
        cfg = StepConfig(filtered=filtered, rebuild_trio=rebuild)
        a = profile_step(eng, names, haps, avg, cfg, single_call=True)
        b = profile_step(eng, names, haps, avg, cfg, single_call=False)
        assert a[0] == b[0]
        assert a[1] == b[1]
        assert np.array_equal(np.array(a[2]["obj"], dtype=float), np.array(b[2]["obj"], dtype=float), equal_nan=True)
        assert a[2]["n_active"] == b[2]["n_active"]


def test_stream_of_steps_with_tables_built_behind_the_next_step(eng, monkeypatch):
    """profile_steps_pipelined (what bench.py times): the tables of step i are built on a helper thread while step i+1's
    kernels run; with a different input per step the results equal one profile_step call per input, in order."""
    import synthdata as synth
    from pantax_amd.pipeline import StepConfig, profile_step, profile_steps_pipelined
    sset = synth.make_set(77, 5, 4, 60000, 80000)
    rd = sset.reads
    eng.upload_db(sset.species)
    names = [g.name for g in sset.species]
    haps = [h for g in sset.species for h in g.hap_names]
    avg = sset.avg_len()
    rng = np.random.default_rng(3)
    flag_sets = [None] + [(rng.random(rd.n_reads) < f).astype(np.uint8) for f in (0.3, 0.6, 0.1)]   # four different samples of the same reads
    single = []
    for fl in flag_sets:
        eng.upload_packed(rd, fl)
        single.append(profile_step(eng, names, haps, avg, StepConfig()))
    from pantax_amd import pipeline
    for min_haps in (1, 10 ** 9):                                   # helper-thread path, inline path
        monkeypatch.setattr(pipeline, "PIPELINE_THREAD_MIN_HAPS", min_haps)
        got = profile_steps_pipelined(eng, names, haps, avg, len(flag_sets), StepConfig(), next_input=lambda i: eng.upload_packed(rd, flag_sets[i]))
        assert len(got) == len(single)
        for a, b in zip(got, single):
            assert a[0] == b[0] and a[1] == b[1] and a[2]["n_active"] == b[2]["n_active"]
    assert any(x[1] != single[0][1] for x in single[1:])          # the inputs really differ


def test_many_species_radix_path_against_oracle(eng):
    """8 species, 6.3e5 nodes in all: above the sample-sort limit, so the LP rows go through the LSD radix sort and
    eight workgroups solve their LPs side by side.  Every species against the oracle: bit-exact integers, equal
    objectives and metrics."""
    from oracle import oracle as orc
    import synthdata as synth
    from pantax_amd.engine import metrics_to_dicts
    sset = synth.make_set(4242, 8, 10, 800_000, 1_250_000)
    rd = sset.reads
    eng.upload_db(sset.species)
    assert eng.V > 600_000
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    ref_sp = orc.bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species])
    assert np.array_equal(sp, ref_sp)
    eng.db_reset()
    eng.trio_nodes_info(fetch=False)
    bases, cov, tb, nab = eng.get_node_abundances()
    keep, absolute, abundance = eng.species_profiling((rc, bs, lm, uq), sset.avg_len())
    met, info = eng.strain_profiling(absolute, species_active=keep)
    gm_all = metrics_to_dicts(met, eng.H)
    nb = np.cumsum([0] + [g.n_nodes for g in sset.species])
    hb = np.cumsum([0] + [g.n_paths for g in sset.species])
    hto = eng.trio_nodes_info()[3].astype(np.int64)
    for s, g in enumerate(sset.species):
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(rd, np.nonzero(sp == s)[0])
        b, c, t, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        assert np.array_equal(bases[nb[s]:nb[s + 1]], b) and np.array_equal(cov[nb[s]:nb[s + 1]], c)
        assert np.array_equal(tb[hto[hb[s]]:hto[hb[s + 1]]], t)
        if not keep[s]:
            continue
        rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
        orc.abundance_constraint(absolute[s], omet)
        assert info[s].n_candidates == nc and info[s].status1 == 0, s
        if nc:
            assert info[s].obj1 == pytest.approx(o1, rel=1e-9), s
        for gm, em in zip(gm_all[hb[s]:hb[s + 1]], orc.metrics_to_dicts(omet)):
            for key, ev in em.items():
                if ev is None or isinstance(ev, bool):
                    assert gm[key] == ev, (s, key)
                else:
                    assert gm[key] == pytest.approx(ev, rel=1e-7, abs=1e-9), (s, key)


def test_step_enqueue_collect_halves(eng):
    """pantax_hip_profile_step_enqueue / _collect: step i+1 enqueued before step i is collected gives, step for step, what the
    one-call form gives; a third enqueue without a collect, a collect without an enqueue and a one-call step while something is
    in flight are refused."""
    import synthdata as synth
    from pantax_amd._ffi import PantaxHipError
    from pantax_amd.engine import metrics_to_dicts
    sset = synth.make_set(78, 4, 5, 50000, 90000)
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    avg = sset.avg_len()

    def snap(t):
        keep, absolute, met, info, passed, s_all, s_pass = t
        return (keep.copy(), absolute.copy(), metrics_to_dicts(met, eng.H), [(info[s].status1, info[s].obj1, info[s].obj2) for s in range(eng.S)],
                np.array(passed).copy(), s_all.copy(), s_pass.copy())
    ref_a = snap(eng.profile_step(avg, fr=0.3))
    avg2 = np.asarray(avg, dtype=np.float64) * 2.0         # halves every predicted coverage: the two steps cannot be mistaken for one another
    ref_b = snap(eng.profile_step(avg2, fr=0.3, rebuild_trio=False))
    eng.profile_step_enqueue(avg, fr=0.3)
    eng.profile_step_enqueue(avg2, fr=0.3, rebuild_trio=False)
    with pytest.raises(PantaxHipError):
        eng.profile_step_enqueue(avg)                      # two in flight already
    with pytest.raises(PantaxHipError):
        eng.profile_step(avg)                              # the one-call form needs an empty queue
    got_a = snap(eng.profile_step_collect())
    eng.profile_step_enqueue(avg, fr=0.3)                  # slot of step a again, while b is still in flight
    got_b = snap(eng.profile_step_collect())
    got_c = snap(eng.profile_step_collect())
    with pytest.raises(PantaxHipError):
        eng.profile_step_collect()
    for got, ref in ((got_a, ref_a), (got_b, ref_b), (got_c, ref_a)):
        for x, y in zip(got, ref):
            if isinstance(x, np.ndarray):
                assert np.array_equal(x, y)
            else:
                assert x == y
    assert not np.array_equal(ref_a[1], ref_b[1])          # the two steps really differ
    eng._inflight = 0
