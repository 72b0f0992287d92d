"""CPU tests of the test-data plumbing of bench.py's default workload: the native synthetic generator
(tools/native/synth_set.c: every species and every chunk of reads is a pure function of the seed) and the pthread driver
around the oracle that bench.py's cpu_baseline uses (oracle/oracle_parallel.c)."""
import numpy as np

import synthdata as synth


def _set(threads=4):
    return synth.NativeSet(20260599, 5, 6, 40_000, 60_000, threads=threads)


def test_native_set_is_deterministic_and_chunks_are_slices():
    a, b = _set(1), _set(4)
    sa, sb = a.make(), b.make()
    for ga, gb in zip(sa.species, sb.species):
        assert np.array_equal(ga.node_len, gb.node_len) and np.array_equal(ga.path_nodes, gb.path_nodes) and np.array_equal(ga.path_off, gb.path_off)
        assert ga.hap_names == gb.hap_names and ga.range_start == gb.range_start
    ra, rb = sa.reads, sb.reads
    assert ra.n_reads == 40_000 and np.array_equal(ra.node_id, rb.node_id) and np.array_equal(ra.pstart, rb.pstart) and np.array_equal(ra.mapq, rb.mapq)
    # a chunk range generated alone == the same slice of the whole set (what a rank of a strong-scaling run generates)
    lo, hi = 64, 160
    part = _set(2).reads(lo, hi)
    r0 = sum(a.chunk_reads(c) for c in range(lo))
    so = ra.step_off.astype(np.int64)
    assert part.n_reads == sum(a.chunk_reads(c) for c in range(lo, hi))
    assert np.array_equal(part.node_id, ra.node_id[so[r0]:so[r0 + part.n_reads]])
    assert np.array_equal(part.pend, ra.pend[r0:r0 + part.n_reads]) and np.array_equal(part.strand, ra.strand[so[r0]:so[r0 + part.n_reads]])
    # graphs of a subset generated alone == the same graphs
    g2 = _set(2).graphs([3])[0]
    assert np.array_equal(g2.path_nodes, sa.species[3].path_nodes) and g2.range_start == sa.species[3].range_start


def test_native_set_model():
    ns = _set()
    sset = ns.make()
    rd = sset.reads
    so = rd.step_off.astype(np.int64)
    k = np.diff(so)
    assert k.min() >= 1 and np.all(rd.qlen == 150)
    starts = np.array([g.range_start for g in sset.species])
    ends = np.array([g.range_end for g in sset.species])
    assert np.all(starts[1:] == ends[:-1] + 1)
    for g in sset.species:
        assert g.n_paths == 6 and len(g.node_len) == g.range_end - g.range_start + 1
        assert int(g.path_off[-1]) == len(g.path_nodes) and g.path_nodes.max() < g.n_nodes
        for h in range(g.n_paths):
            w = g.path_nodes[int(g.path_off[h]):int(g.path_off[h + 1])]
            assert np.all(np.diff(w.astype(np.int64)) > 0)                       # walks go up the node order
            assert int(g.node_len[w].sum()) == int(g.genome_len[h])
        assert (g.truth_depth > 0).sum() == 1                                    # round(0.2 * 6) strains present
    # error-free reads: path span == read length unless the record is adversarial
    first = rd.node_id[so[:-1]].astype(np.int64)
    sp = np.searchsorted(starts, first, side="right") - 1
    gl = np.concatenate([g.node_len for g in sset.species])
    base = np.cumsum([0] + [g.n_nodes for g in sset.species])
    plain = (rd.pend - rd.pstart == 150)
    assert plain.mean() > 0.99
    r = int(np.nonzero(plain & (k >= 3))[0][0])
    ids = rd.node_id[so[r]:so[r + 1]].astype(np.int64)
    if np.all((ids >= starts[sp[r]]) & (ids <= ends[sp[r]])) and len(set(ids)) == len(ids):
        lens = gl[base[sp[r]] + ids - starts[sp[r]]]
        assert lens[0] - rd.pstart[r] + lens[1:-1].sum() < 150 <= lens.sum() - rd.pstart[r]
    assert 0.80 < (rd.mapq == 60).mean() < 0.90 and 0.45 < rd.strand.mean() < 0.55


def test_parallel_gaf_writer_equals_serial(tmp_path):
    rd = _set().reads(0, 8)
    a, b = tmp_path / "a.gaf", tmp_path / "b.gaf"
    synth.write_gaf(rd, str(a))
    n = synth.write_gaf_parallel(rd, str(b), threads=3)
    ta, tb = a.read_bytes(), b.read_bytes()
    assert n == len(tb) and ta == tb and ta.count(b"\n") == rd.n_reads


def test_parallel_oracle_driver_equals_serial_calls():
    from oracle import oracle as orc
    from tests.helpers import select_reads
    ns = synth.NativeSet(20260598, 4, 5, 30_000, 50_000, present_frac=0.6, threads=2)
    sset = ns.make()
    rd = sset.reads
    S = len(sset.species)
    sp = orc.par_bin_reads(rd.step_off, rd.node_id, ns.range_start, ns.range_end, 3)
    assert np.array_equal(sp, orc.bin_reads(rd.step_off, rd.node_id, ns.range_start, ns.range_end))
    first, order = orc.group_reads(sp, S)
    o2 = np.argsort(sp, kind="stable")
    assert np.array_equal(order, o2[(sp[o2] >= 0)].astype(np.uint64))
    counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
    keep, absolute, _ = orc.species_profile(sp, rd.qlen, counts, ns.avg_len())
    Gs = [orc.Graph(g.node_len, g.path_off, g.path_nodes) for g in sset.species]
    pr = orc.par_profile_species(Gs, ns.range_start, rd.step_off, rd.node_id, rd.pstart, rd.pend, first, order, keep, absolute, np.arange(S), 3,
                                 want_metrics=True)
    for s, g in enumerate(sset.species):
        T = orc.TrioTable(Gs[s])
        so, nid, ps, pe = select_reads(rd, np.nonzero(sp == s)[0])
        b, c, tb, _ = orc.node_coverage(Gs[s], T, g.range_start, so, nid, ps, pe)
        rc, met, nc, o1, o2_ = orc.optimize_species(Gs[s], T, b, c, tb)
        orc.abundance_constraint(absolute[s], met)
        assert pr["rc"][s] == rc and pr["n_cand"][s] == nc and pr["n_rows"][s] == int((b > 0).sum())
        assert pr["obj1"][s] == o1 and pr["obj2"][s] == o2_
        h0 = int(pr["hap_off"][s])
        for h in range(g.n_paths):
            m, e = pr["metrics"][h0 + h], met[h]
            assert m.has == e.has and m.second_sol == e.second_sol and m.first_sol == e.first_sol and m.total_cov_diff == e.total_cov_diff
