"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the
C ABI, against the CPU oracle on the same seeded inputs and against the golden vectors.
Integer outputs must be bit-exact."""
import numpy as np
import pytest

from tests.helpers import load_micro_bin, load_micro_cov, select_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pantax_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


class _G:
    def __init__(self, node_len, path_off, path_nodes, rs):
        self.node_len, self.path_off, self.path_nodes = node_len, path_off, path_nodes
        self.range_start, self.range_end = rs, rs + len(node_len) - 1


def _oracle_cov_per_species(sset, sp):
    from oracle import oracle as orc
    out = []
    for si, g in enumerate(sset.species):
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(sset.reads, np.nonzero(sp == si)[0])
        out.append((G, T) + orc.node_coverage(G, T, g.range_start, so, nid, ps, pe))
    return out


def test_micro_binning_golden(eng):
    j, step_off, node_id, qlen, mapq, species, rs, re = load_micro_bin()
    gs = [_G(np.ones(int(e - s + 1), dtype=np.int64), np.array([0, 1], dtype=np.uint64), np.array([0], dtype=np.uint32), int(s))
          for s, e in zip(rs, re)]
    eng.upload_db(gs)
    eng.upload_reads(step_off, node_id, np.zeros(len(qlen)), np.ones(len(qlen)), qlen, mapq)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    assert sp.tolist() == species.tolist()
    e = j["expected_counts"]
    assert rc.tolist() == e["read_count"] and bs.tolist() == e["base_sum"]
    assert lm.tolist() == e["less_multi"] and uq.tolist() == e["uniq_count"]


def test_micro_coverage_golden(eng):
    j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend = load_micro_cov()
    eng.upload_db([_G(node_len, path_off, path_nodes, rs)])
    R = len(pstart)
    eng.upload_reads(step_off, node_id, pstart, pend, np.full(R, 150), np.full(R, 60))
    sp, *_ = eng.rcls_profile()
    # the abort read that walks below the range is "U" for binning; force species 0 semantics by
    # checking only what the pipeline would feed: reads binned to species 0
    trio = None
    try:
        trio = eng.trio_nodes_info()
    except Exception:
        pass
    bases, cov, tb, nab = eng.get_node_abundances()
    e = j["expected"]
    assert bases.tolist() == e["bases_per_node"]
    assert cov.tolist() == e["node_base_cov"]
    # r11 (node below range) is binned "U" and never reaches the histogram; r10 is the counted abort
    assert nab == 1
    if trio is not None:
        abc, hap, ln, hto = trio
        ex = j["expected_trios"]
        assert abc.tolist() == ex["abc"] and hap.tolist() == ex["hap"] and ln.tolist() == ex["len"]
        assert hto.tolist() == ex["hap_off"]
        assert tb.tolist() == e["trio_bases"]


@pytest.mark.parametrize("seed,S,H,R,L", [(1, 1, 4, 3000, 15000), (2, 3, 6, 20000, 40000), (3, 5, 10, 50000, 30000)])
def test_binning_and_coverage_vs_oracle(eng, seed, S, H, R, L):
    from oracle import oracle as orc
    from pantax_amd import synth
    sset = synth.make_set(seed, S, H, R, L, adversarial_frac=0.01, single_strain_every=4 if S >= 5 else 0)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    ref_sp = orc.bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species])
    assert np.array_equal(sp, ref_sp)
    ref_counts = orc.species_counts(ref_sp, rd.qlen, rd.mapq, S)
    for a, b in zip((rc, bs, lm, uq), ref_counts):
        assert np.array_equal(a, b)
    have_trio = True
    try:
        abc, hap, ln, hto = eng.trio_nodes_info()
    except Exception:
        have_trio = False
    bases, cov, tb, nab = eng.get_node_abundances()
    ref = _oracle_cov_per_species(sset, ref_sp)
    tot_abort = 0
    for si, (G, T, b, c, t, na) in enumerate(ref):
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        assert np.array_equal(bases[lo:hi], b)
        assert np.array_equal(cov[lo:hi], c)
        tot_abort += na
        if have_trio:
            h0, h1 = int(eng.hap_off[si]), int(eng.hap_off[si + 1])
            u0, u1 = int(hto[h0]), int(hto[h1])
            assert u1 - u0 == T.n_unique
            assert np.array_equal(abc[u0:u1], T.abc) and np.array_equal(hap[u0:u1], T.hap)
            assert np.array_equal(ln[u0:u1], T.len)
            assert np.array_equal(hto[h0:h1 + 1] - hto[h0], T.hap_off)
            assert np.array_equal(tb[u0:u1], t)
    assert nab == tot_abort


def test_species_active_mask_and_flags(eng):
    from oracle import oracle as orc
    from pantax_amd import synth
    sset = synth.make_set(9, 3, 4, 6000, 15000)
    rd = sset.reads
    eng.upload_db(sset.species)
    flags = (np.arange(rd.n_reads) % 7 == 0).astype(np.uint8)
    eng.upload_packed(rd, flags=flags)
    sp, *_ = eng.rcls_profile()
    active = np.array([1, 0, 1], dtype=np.uint8)
    bases, cov, tb, nab = eng.get_node_abundances(species_active=active, with_trio=False)
    for si, g in enumerate(sset.species):
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        if not active[si]:
            assert not bases[lo:hi].any() and not cov[lo:hi].any()
            continue
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        so, nid, ps, pe = select_reads(rd, np.nonzero((sp == si) & (flags == 0))[0])
        b, c, _, _ = orc.node_coverage(G, None, g.range_start, so, nid, ps, pe)
        assert np.array_equal(bases[lo:hi], b) and np.array_equal(cov[lo:hi], c)


def test_long_reads_and_empty_inputs(eng):
    from oracle import oracle as orc
    from pantax_amd import synth
    sset = synth.make_set(5, 2, 5, 300, 60000, long_reads=True)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, *_ = eng.rcls_profile()
    bases, cov, tb, nab = eng.get_node_abundances(with_trio=False)
    for si, (G, T, b, c, t, na) in enumerate(_oracle_cov_per_species(sset, sp)):
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        assert np.array_equal(bases[lo:hi], b) and np.array_equal(cov[lo:hi], c)
    # zero reads
    eng.upload_reads(np.zeros(1), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0))
    sp, rc, *_ = eng.rcls_profile()
    assert len(sp) == 0 and not rc.any()
    bases, cov, tb, nab = eng.get_node_abundances(with_trio=False)
    assert not bases.any() and not cov.any() and nab == 0
