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


def test_binning_with_overlapping_ranges_is_first_match_in_file_order(eng):
    """species_range.txt rows that overlap or nest (never written by sort_range.rs, but legal input): the reference takes the
    FIRST row, in file order, that contains [min, max] of the read's node ids (rcls.rs:237-258); the sorted-range binary
    search must not be used then."""
    from oracle import oracle as orc
    rs = np.array([40, 1, 120, 300, 90], dtype=np.int64)       # deliberately not sorted, overlapping, one nested
    re = np.array([160, 100, 130, 400, 95], dtype=np.int64)
    gs = [_G(np.ones(int(e - s + 1), dtype=np.int64), np.array([0, 1], dtype=np.uint64), np.array([0], dtype=np.uint32), int(s)) for s, e in zip(rs, re)]
    rng = np.random.default_rng(3)
    walks = []
    for _ in range(4000):
        lo = int(rng.integers(1, 420))
        k = int(rng.integers(1, 6))
        walks.append(np.clip(lo + rng.integers(0, int(rng.choice([3, 15, 60])), size=k), 1, 450).astype(np.uint32))
    walks += [np.array([40], np.uint32), np.array([160], np.uint32), np.array([161], np.uint32), np.array([1, 100], np.uint32),
              np.array([90, 95], np.uint32), np.array([125], np.uint32), np.array([100, 101], np.uint32), np.zeros(0, np.uint32)]
    step_off = np.concatenate([[0], np.cumsum([len(w) for w in walks])]).astype(np.uint64)
    node_id = np.concatenate(walks)
    R = len(walks)
    qlen = rng.integers(50, 200, size=R)
    mapq = rng.choice([0, 3, 30, 60], size=R)
    eng.upload_db(gs)
    eng.upload_reads(step_off, node_id, np.zeros(R), np.ones(R), qlen, mapq)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    ref = orc.bin_reads(step_off, node_id, rs, re)
    assert np.array_equal(sp, ref)
    assert (ref == 0).sum() > 0 and (ref == 1).sum() > 0 and (ref == 2).sum() == 0     # row 2 is shadowed by row 0
    assert sp[R - 3] == 0 and sp[R - 4] == 0 and sp[R - 5] == 1                         # 125 -> row 0; [90,95] -> row 0; [1,100] -> row 1
    for a, b in zip((rc, bs, lm, uq), orc.species_counts(ref, qlen, mapq, len(rs))):
        assert np.array_equal(a, b)


def test_micro_coverage_golden(eng):
    j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend = load_micro_cov()
    eng.upload_db([_G(node_len, path_off, path_nodes, rs)])
    R = len(pstart)
    eng.upload_reads(step_off, node_id, pstart, pend, np.full(R, 150), np.full(R, 60))
    sp, *_ = eng.rcls_profile()
    # the abort read that walks below the range is "U" for binning; force species 0 semantics by
    # checking only what the pipeline would feed: reads binned to species 0
    trio = eng.trio_nodes_info()
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


@pytest.mark.parametrize("seed,S,H,R,L,uniq", [(1, 1, 4, 3000, 15000, None), (2, 3, 6, 20000, 40000, None), (3, 5, 10, 50000, 30000, None),
                                                  (4, 2, 40, 20000, 20000, None),   # > 16 windows per node: the hashed uniqueness test
                                                  (2, 3, 6, 20000, 40000, "1"), (4, 2, 40, 20000, 20000, "0"),   # and each form forced
                                                  (2, 3, 6, 20000, 40000, "block"), (3, 5, 10, 50000, 30000, "block"), (4, 2, 40, 20000, 20000, "block")])
def test_binning_and_coverage_vs_oracle(eng, seed, S, H, R, L, uniq, set_opt):
    from oracle import oracle as orc
    import synthdata as synth
    # default: uniqueness through the visit table (species with a node of more than 64 visits: by node block in LDS); forced: every
    # species by node block, or the global bucket path with either of its kernels
    if uniq == "block":
        set_opt(eng, "trio_path", "block")   # read by the library at db upload
    elif uniq is not None:
        set_opt(eng, "trio_path", "bucket")
        set_opt(eng, "uniq_hash", uniq)   # read by the library at every trio build
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
    abc, hap, ln, hto = eng.trio_nodes_info()
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


@pytest.mark.parametrize("general,long_kernel,shape", [("1", None, None), ("1", "step", None), ("1", None, "1222"), ("1", None, "2448"), (None, None, None)])
def test_coverage_kernels_agree_on_short_reads(eng, general, long_kernel, shape, set_opt):
    """Short reads take coverage_fast_kernel (one wave per 64-step group); cov_general=1 sends the same groups through
    the kernel that otherwise only sees the groups of longer walks -- round 6's select-only instantiation in several shapes, or round 5's
    coverage_step_kernel (cov_long=step).  All must equal the oracle bit for bit."""
    import synthdata as synth
    if general:
        set_opt(eng, "cov_general", general)
    if long_kernel:
        set_opt(eng, "cov_long", long_kernel)
    if shape:
        set_opt(eng, "covl_shape", shape)
    sset = synth.make_set(11, 4, 6, 60000, 50000, adversarial_frac=0.02, single_strain_every=4)
    from oracle import oracle as orc
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, *_ = eng.rcls_profile()
    eng.trio_nodes_info(fetch=False)
    bases, cov, tb, nab = eng.get_node_abundances()
    ref = _oracle_cov_per_species(sset, sp)
    u0 = 0
    for si, (G, T, b, c, t, na) in enumerate(ref):
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        assert np.array_equal(bases[lo:hi], b) and np.array_equal(cov[lo:hi], c)
        assert np.array_equal(tb[u0:u0 + T.n_unique], t)
        u0 += T.n_unique
    assert nab == sum(r[5] for r in ref)


def test_range_wider_than_the_graph_is_refused_at_upload(eng):
    """optimize_otu derives nvert from the species' range (profile.rs:2938), so a range that does not span exactly the graph's
    nodes is not a database the reference could run; db_upload refuses it -- which is what lets the coverage pass place a node id
    of a binned read (inside the range, rcls.rs:253-257) without a bounds test."""
    from pantax_amd.engine import PantaxHipError
    g = _G(np.array([10, 20, 30, 40, 50], dtype=np.int64), np.array([0, 5], dtype=np.uint64), np.arange(5, dtype=np.uint32), 101)
    g.range_end = 110
    with pytest.raises(PantaxHipError):
        eng.upload_db([g])


@pytest.mark.parametrize("k", [0, 1, 2])
def test_coverage_and_trio_index_vs_literal_python_restatement(eng, k):
    """The HIP path against fixtures from oracle/ref_literal.py -- the literal Python reading of profile.rs:658-1026 that
    shares no code with the C oracle: unique-trio table (keyed), trio bases, bases_per_node, node_base_cov bit for bit."""
    from tests.helpers import check_against_literal, load_literal_case
    j, names, node_len, path_off, path_nodes, rs, step_off, node_id, pstart, pend = load_literal_case(k)
    g = _G(node_len, path_off, path_nodes, rs)
    eng.upload_db([g])
    R = len(pstart)
    eng.upload_reads(step_off, node_id, pstart, pend, np.full(R, 60), np.full(R, 60))
    sp, *_ = eng.rcls_profile()
    # a walk that leaves the species is "U" for the binning (rcls.rs:253-257) and never reaches get_node_abundances; the
    # fixture holds one such read to pin the reference's index panic, which the HIP path therefore cannot count
    so = step_off.astype(np.int64)
    outside = [r for r in range(R) if any(not (rs <= int(x) < rs + len(node_len)) for x in node_id[so[r]:so[r + 1]])]
    assert all(sp[r] == -1 for r in outside) and (sp == -1).sum() == len(outside) + sum(1 for r in range(R) if so[r] == so[r + 1])
    abc, hap, ln, hto = eng.trio_nodes_info()
    bases, cov, tb, nab = eng.get_node_abundances()
    check_against_literal(j, names, abc, hap, ln, tb, bases, cov, nab + len(outside))


@pytest.mark.parametrize("path", ["block", None])
@pytest.mark.parametrize("V,H,K", [(200, 40, 300), (700, 30, 500), (300, 3, 9000), (3000, 12, 4000)])
def test_trio_index_block_path_overflowing_lds_table(eng, V, H, K, path, set_opt):
    """Random walks over a few hundred nodes: a node block meets thousands of DISTINCT windows, more than its LDS table
    holds, so it is redone in sub-passes over key classes; walks jump between blocks at every step (runs of length 1)
    and visit both orientations of the same window.  `path` None: whatever the upload chooses -- the visit table where no
    node has more than 64 visits (700 x 30 x 500, 3000 x 12 x 4000: stretches of every length up to the limit, walks that
    return to a node, windows with equal ends), the node-block kernel otherwise."""
    from oracle import oracle as orc
    if path:
        set_opt(eng, "trio_path", path)
    rng = np.random.default_rng(V + H)
    node_len = rng.integers(1, 40, size=V).astype(np.int64)
    walks = [rng.integers(0, V, size=K).astype(np.uint32) for _ in range(H)]
    walks[1] = np.concatenate([walks[0][::-1][:K // 2], walks[1]])            # reverse traversals of another walk's windows
    walks[2] = np.concatenate([walks[2], np.array([5, 5, 5, 5, 6, 5, 6], dtype=np.uint32)])   # a == c windows and repeats
    path_off = np.concatenate([[0], np.cumsum([len(w) for w in walks])]).astype(np.uint64)
    g = _G(node_len, path_off, np.concatenate(walks), 1)
    eng.upload_db([g])
    abc, hap, ln, hto = eng.trio_nodes_info()
    T = orc.TrioTable(orc.Graph(node_len, path_off, np.concatenate(walks)))
    assert len(abc) == T.n_unique
    assert np.array_equal(abc, T.abc) and np.array_equal(hap, T.hap) and np.array_equal(ln, T.len) and np.array_equal(hto, T.hap_off)


@pytest.mark.parametrize("uniq", ["0", "1", None])
def test_trio_index_with_huge_buckets(eng, uniq, set_opt):
    """Paths that keep coming back to a handful of nodes: thousands of windows share their smallest end node, far more
    than one workgroup's LDS table holds (the hashed form falls back to scanning the bucket), and most trios repeat."""
    from oracle import oracle as orc
    if uniq is not None:
        set_opt(eng, "trio_path", "bucket")
        set_opt(eng, "uniq_hash", uniq)
    rng = np.random.default_rng(17)
    V, H, K = 9, 40, 300
    node_len = rng.integers(1, 40, size=V).astype(np.int64)
    walks = [rng.integers(0, V, size=K).astype(np.uint32) for _ in range(H)]
    for h in range(0, H, 5):                       # and some stretches nobody else has
        walks[h] = np.concatenate([walks[h], np.array([0, 8, 0, 8, 7, 7, 7, h % V, 3], dtype=np.uint32)])
    path_off = np.concatenate([[0], np.cumsum([len(w) for w in walks])]).astype(np.uint64)
    g = _G(node_len, path_off, np.concatenate(walks), 1)
    eng.upload_db([g])
    abc, hap, ln, hto = eng.trio_nodes_info()
    T = orc.TrioTable(orc.Graph(node_len, path_off, np.concatenate(walks)))
    assert len(abc) == T.n_unique
    assert np.array_equal(abc, T.abc) and np.array_equal(hap, T.hap) and np.array_equal(ln, T.len) and np.array_equal(hto, T.hap_off)


def test_species_active_mask_and_flags(eng):
    from oracle import oracle as orc
    import synthdata as synth
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


@pytest.mark.parametrize("long_kernel", [None, "step"])
def test_long_reads_and_empty_inputs(eng, long_kernel, set_opt):
    from oracle import oracle as orc
    import synthdata as synth
    if long_kernel:
        set_opt(eng, "cov_long", long_kernel)
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


@pytest.mark.parametrize("long_kernel,shape", [(None, None), (None, "1120"), (None, "2242"), (None, "2848"), ("step", None), ("walk_sum_in_bin", None)])
def test_long_walks_with_revisits(eng, long_kernel, shape, set_opt):
    """(every shape of the long-walk kernel -- groups in flight, groups per workgroup, window size, window start -- and round 5's kernel)
    Walks of 65 .. 6000 steps (more than a wave, more than the upload-time hash holds) that come back to nodes they
    have already visited, the first node included, with arbitrary start/end offsets: first-occurrence rule
    (profile.rs:879-882), `seen` across waves (:857-859) and the trio windows at wave borders, bit for bit."""
    from oracle import oracle as orc
    import synthdata as synth
    if long_kernel == "walk_sum_in_bin":         # the walk sums inside the binning pass instead of by walk_sum_kernel
        set_opt(eng, "walk_sum_in_bin", "1")
    elif long_kernel:
        set_opt(eng, "cov_long", long_kernel)
    if shape:
        set_opt(eng, "covl_shape", shape)
    sset = synth.make_set(91, 2, 4, 200, 200000, long_reads=True)
    rng = np.random.default_rng(92)
    offs, ids, ps, pe = [0], [], [], []
    lens = [65, 66, 127, 128, 129, 200, 500, 1000, 2500, 4095, 4096, 4097, 6000] + list(rng.integers(65, 1500, size=120))
    for n, k in enumerate(lens):
        g = sset.species[n % 2]
        h = int(rng.integers(0, g.n_paths))
        path = g.path_nodes[int(g.path_off[h]):int(g.path_off[h + 1])]
        walk = []
        while len(walk) < k:                       # pieces of the path, each starting somewhere near what was already walked
            if walk and rng.random() < 0.7:
                at = int(np.clip(np.searchsorted(path, walk[int(rng.integers(0, len(walk)))]) + rng.integers(-3, 4), 0, len(path) - 1))
            else:
                at = int(rng.integers(0, len(path)))
            piece = path[at:at + int(rng.integers(2, 400))]
            if rng.random() < 0.5:
                piece = piece[::-1]
            walk.extend(int(x) for x in piece)
            if rng.random() < 0.3:
                walk.append(walk[0])              # back to the very first node
        walk = np.array(walk[:k], dtype=np.int64)
        total = int(g.node_len[walk].sum())
        a = int(rng.integers(0, g.node_len[walk[0]] + 1))
        mode = n % 4
        b = a + [total - a - int(rng.integers(0, g.node_len[walk[-1]] + 1)), int(rng.integers(0, 50)), total + 77, total // 2][mode]
        ids.append(walk + g.range_start)
        offs.append(offs[-1] + k)
        ps.append(a)
        pe.append(max(b, 0))
    R = len(lens)
    rd = synth.PackedReads(np.array(offs, dtype=np.uint64), np.concatenate(ids).astype(np.uint32), np.zeros(offs[-1], np.uint8),
                           np.array(ps, dtype=np.int64), np.array(pe, dtype=np.int64), np.full(R, 15000, np.int64), np.full(R, 60, np.int64),
                           np.zeros(R, np.int64))
    sset.reads = rd
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, *_ = eng.rcls_profile()
    assert (sp == np.arange(R) % 2).all()
    eng.trio_nodes_info(fetch=False)
    bases, cov, tb, nab = eng.get_node_abundances()
    assert nab == 0
    trio_off = 0
    for si, (G, T, b, c, t, na) in enumerate(_oracle_cov_per_species(sset, sp)):
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        assert np.array_equal(bases[lo:hi], b) and np.array_equal(cov[lo:hi], c)
        assert np.array_equal(tb[trio_off:trio_off + len(t)], t) and t.sum() > 0
        trio_off += len(t)
    assert trio_off == len(tb)


@pytest.mark.parametrize("long_kernel,shape", [(None, None), (None, "1232"), ("step", None)])
def test_short_and_long_reads_in_one_sample(eng, long_kernel, shape, set_opt):
    """A sample that mixes 150-bp reads with HiFi-shaped ones (both strands, adversarial records): the groups that hold a step of a longer walk
    go to the long-walk kernel, the others to the short-read one, and a group may hold walks of both kinds.  Bit for bit against the oracle."""
    import synthdata as synth
    if long_kernel:
        set_opt(eng, "cov_long", long_kernel)
    if shape:
        set_opt(eng, "covl_shape", shape)
    a = synth.make_set(23, 3, 5, 900, 120000, long_reads=True, adversarial_frac=0.02)
    b = synth.make_set(23, 3, 5, 40000, 120000, adversarial_frac=0.02)
    ra, rb = a.reads, b.reads
    assert all(np.array_equal(x.path_nodes, y.path_nodes) for x, y in zip(a.species, b.species))      # the same graphs
    order = np.random.default_rng(5).permutation(ra.n_reads + rb.n_reads)
    lens = np.concatenate([np.diff(ra.step_off.astype(np.int64)), np.diff(rb.step_off.astype(np.int64))])
    starts = np.concatenate([ra.step_off[:-1].astype(np.int64), rb.step_off[:-1].astype(np.int64) + int(ra.step_off[-1])])
    ids_all = np.concatenate([ra.node_id, rb.node_id])
    strand_all = np.concatenate([ra.strand, rb.strand])
    offs = np.concatenate([[0], np.cumsum(lens[order])]).astype(np.uint64)
    idx = np.repeat(starts[order] - offs[:-1].astype(np.int64), lens[order]) + np.arange(int(offs[-1]))
    cat = lambda f: np.concatenate([getattr(ra, f), getattr(rb, f)])[order]
    rd = synth.PackedReads(offs, ids_all[idx].astype(np.uint32), strand_all[idx], cat("pstart"), cat("pend"), cat("qlen"), cat("mapq"), cat("plen"))
    a.reads = rd
    assert (lens > 64).any() and (lens <= 64).any()
    eng.upload_db(a.species)
    eng.upload_packed(rd)
    sp, *_ = eng.rcls_profile()
    eng.trio_nodes_info(fetch=False)
    bases, cov, tb, nab = eng.get_node_abundances()
    ref = _oracle_cov_per_species(a, sp)
    u0 = 0
    for si, (G, T, b_, c, t, na) in enumerate(ref):
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        assert np.array_equal(bases[lo:hi], b_) and np.array_equal(cov[lo:hi], c)
        assert np.array_equal(tb[u0:u0 + T.n_unique], t)
        u0 += T.n_unique
    assert nab == sum(r[5] for r in ref)


def _paths_from_masks(mask, p):
    """mask: (n,) uint64, or (n, nw) words for more than 64 columns."""
    offs = [0]
    nodes = []
    m2 = mask.reshape(len(mask), -1)
    for k in range(p):
        sel = np.nonzero((m2[:, k >> 6] >> np.uint64(k & 63)) & np.uint64(1))[0]
        nodes.append(sel.astype(np.uint32))
        offs.append(offs[-1] + len(sel))
    return np.array(offs, dtype=np.uint64), (np.concatenate(nodes) if nodes else np.zeros(0, dtype=np.uint32))


def test_pao_solve_vs_highs_golden(eng, golden_dir):
    """Solver seam (X_opt, profile.rs:2690-2698) on the committed SciPy-HiGHS vectors: objective equal
    to 1e-9 relative; x equal where the optimal face is a single point."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(golden_dir, "lp_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub, xh, objh = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i], z["x_%d" % i], float(z["obj_%d" % i])
        p = len(ub)
        path_off, path_nodes = _paths_from_masks(mask, p)
        x, ratio, obj, st = eng.pao_solve(np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), path_off,
                                          path_nodes, np.arange(p), fixed_zero=(ub == 0).astype(np.uint8))
        assert st == 0
        assert obj == pytest.approx(objh, rel=1e-9, abs=1e-12), i
        assert orc.lad_objective(mask, a, x) == pytest.approx(objh, rel=1e-9, abs=1e-12), i
        assert np.all(x >= 0) and np.all(x <= ub + 1e-12)
        if bool(z["unique_%d" % i]):
            assert np.abs(x - xh).sum() <= 1e-6 * max(1.0, np.abs(xh).sum()), (i, x, xh)


def test_pao_solve_vs_reference_milp_model(eng, golden_dir):
    """Solver seam against the reference's ACTUAL mixed-integer model (binary indicators + sum z <= npaths, second solve with
    x_j == 0 rows; profile.rs:1363-1377, 1484-1488) solved by scipy.optimize.milp at the BASELINE.md section 2 shape (20 000 x
    10, 60 % dense) and on integer-tied coverages: objective equal to 1e-9, pinned columns exactly zero."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(golden_dir, "lp_milp_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, fixed = z["mask_%d" % i], z["a_%d" % i], z["fixed_%d" % i]
        p = len(fixed)
        po, pn = _paths_from_masks(mask, p)
        x, ratio, obj, st = eng.pao_solve(np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), po, pn, np.arange(p), fixed_zero=fixed)
        objm = float(z["obj_milp_%d" % i])
        assert st == 0
        assert obj == pytest.approx(objm, rel=1e-9, abs=1e-12), str(z["name_%d" % i])
        assert orc.lad_objective(mask, a, x) == pytest.approx(objm, rel=1e-9, abs=1e-12)
        assert np.all(x[fixed == 1] == 0.0) and np.all(x >= 0) and np.all(x <= 1.05 * a.max() + 1e-12)
        xm = z["x_milp_%d" % i]
        if np.abs(xm - z["x_lp_%d" % i]).sum() < 1e-9:      # both HiGHS runs landed on the same vertex: compare x too
            assert np.abs(x - xm).sum() <= 1e-6 * max(1.0, np.abs(xm).sum()) or orc.lad_objective(mask, a, xm) == pytest.approx(obj, rel=1e-12)


def test_pao_solve_batch_equals_per_species_calls(eng, golden_dir):
    """pantax_hip_pao_solve_batch (arrays of offsets in, solutions out, host buffers): all 11 golden LPs as ONE
    batch == the same LPs solved one call at a time, objectives == SciPy-HiGHS; a species without candidates, one with 65
    candidates (the wide kernel, four mask words) and one with 257 (more than four words: the kernel without a cap on the columns)
    sit in the same batch."""
    import os
    z = np.load(os.path.join(golden_dir, "lp_cases.npz"))
    species, fixed, objs = [], [], []
    for i in range(int(z["n_cases"])):
        mask, a, ub = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i]
        p = len(ub)
        po, pn = _paths_from_masks(mask, p)
        species.append((np.ones(len(a), dtype=np.int64), a, None, po, pn, np.arange(p)))
        fixed.append((ub == 0).astype(np.uint8))
        objs.append(float(z["obj_%d" % i]))
    single = [eng.pao_solve(sp[0], sp[1], np.zeros(len(sp[1]), dtype=np.uint64), sp[3], sp[4], sp[5], fixed_zero=f) for sp, f in zip(species, fixed)]
    # + a species with nothing to solve, + one beyond the 64-column word (wide path), + one beyond the wide path
    po2, pn2 = _paths_from_masks(np.array([1, 3, 2], dtype=np.uint64), 2)
    species.append((np.ones(3, dtype=np.int64), np.array([1.0, 2.0, 3.0]), None, po2, pn2, np.zeros(0, dtype=np.uint32)))
    fixed.append(np.zeros(0, dtype=np.uint8))
    n65 = 200
    m65 = (np.uint64(1) << (np.arange(n65, dtype=np.uint64) % np.uint64(64)))
    po65, pn65 = _paths_from_masks(m65, 64)
    po65 = np.concatenate([po65, [po65[-1] + 1]]).astype(np.uint64); pn65 = np.concatenate([pn65, [0]]).astype(np.uint32)
    species.append((np.ones(n65, dtype=np.int64), np.ones(n65), None, po65, pn65, np.arange(65)))
    fixed.append(np.zeros(65, dtype=np.uint8))
    po257 = np.arange(258, dtype=np.uint64); pn257 = (np.arange(257) % n65).astype(np.uint32)
    species.append((np.ones(n65, dtype=np.int64), np.ones(n65), None, po257, pn257, np.arange(257)))
    fixed.append(np.zeros(257, dtype=np.uint8))
    batch = eng.pao_solve_batch(species, fixed)
    for i, ((x1, r1, o1, st1), (xb, rb, ob, stb, itb)) in enumerate(zip(single, batch)):
        assert st1 == 0 and stb == 0
        assert np.array_equal(x1, xb), i                                   # the same solver kernel: bit for bit
        assert o1 == pytest.approx(ob, rel=1e-13), i                       # the objective's chunked sum depends on the batch shape: last ulp
        assert ob == pytest.approx(objs[i], rel=1e-9, abs=1e-12), i
    assert batch[-3][3] == 0 and len(batch[-3][0]) == 0
    # 65 columns: every node on exactly one of 64 paths with a = 1, path 64 visits node 0 as well -> objective 0 at x = 1
    x65, r65, o65, st65, it65 = batch[-2]
    assert st65 == 0 and o65 == pytest.approx(0.0, abs=1e-12) and np.allclose(x65[:64] + np.where(np.arange(64) == 0, x65[64], 0.0), 1.0)
    # 257 columns: path k visits node k % 200 only, a = 1 -> the paths of every node sum to 1, objective 0
    x257, r257, o257, st257, it257 = batch[-1]
    assert st257 == 0 and o257 == pytest.approx(0.0, abs=1e-12)
    assert np.allclose(np.bincount(np.arange(257) % n65, weights=x257, minlength=n65), 1.0)


def test_pao_solve_at_and_beyond_64_candidates(eng):
    """64 candidate paths is what one membership word holds: solved (objective == the oracle's exact LAD, which SciPy-HiGHS
    pins on the golden cases); 65 goes through the wide path (four words) and gives the same optimum with an unused extra
    column; 257 and 600 columns (more than four words) go through the kernel that sizes its state at run time."""
    from oracle import oracle as orc
    rng = np.random.default_rng(64)
    n, p = 6000, 64
    # haplotype-like membership: every node on a random subset of the paths, a few paths nearly everywhere
    mask = np.zeros(n, dtype=np.uint64)
    for k in range(p):
        on = rng.random(n) < (0.9 if k < 3 else rng.uniform(0.05, 0.5))
        mask |= (on.astype(np.uint64) << np.uint64(k))
    mask[mask == 0] = 1
    truth = np.where(rng.random(p) < 0.15, rng.uniform(1, 30, p), 0.0)
    A = np.stack([((mask >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1)
    a = np.maximum(A @ truth + rng.normal(0, 0.5, n), 0.01)
    po, pn = _paths_from_masks(mask, p)
    x, ratio, obj, st = eng.pao_solve(np.ones(n, dtype=np.int64), a, np.zeros(n), po, pn, np.arange(p))
    ub = np.full(p, 1.05 * a.max())
    xo, objo, it, sto = orc.lad_solve(mask, a, p, ub)
    assert st == 0 and sto == 0
    assert obj == pytest.approx(objo, rel=1e-9) and orc.lad_objective(mask, a, x) == pytest.approx(objo, rel=1e-9)
    assert np.all(x >= -1e-12) and np.all(x <= ub + 1e-9)
    # one more path than the word holds: the wide path, same LP plus a column that only touches node 0
    po65 = np.concatenate([po, [po[-1] + 1]]).astype(np.uint64)
    pn65 = np.concatenate([pn, [0]]).astype(np.uint32)
    x65, ratio65, obj65, st65 = eng.pao_solve(np.ones(n, dtype=np.int64), a, np.zeros(n), po65, pn65, np.arange(65))
    mask65 = np.zeros((n, 2), dtype=np.uint64); mask65[:, 0] = mask; mask65[0, 1] = 1
    xo65, objo65, it65, sto65 = orc.lad_solve(mask65, a, 65, np.full(65, 1.05 * a.max()))
    assert st65 == 0 and sto65 == 0 and obj65 == pytest.approx(objo65, rel=1e-9) and obj65 <= obj * (1 + 1e-12)
    assert orc.lad_objective(mask65, a, x65) == pytest.approx(objo65, rel=1e-9)
    # more than 256 columns: the same LP plus columns that each touch a few more nodes
    for pw in (257, 600):
        extra = [np.sort(rng.choice(n, size=int(rng.integers(1, 40)), replace=False)).astype(np.uint32) for _ in range(pw - p)]
        pow_ = np.concatenate([po, po[-1] + np.cumsum([len(e_) for e_ in extra])]).astype(np.uint64)
        pnw = np.concatenate([pn] + extra).astype(np.uint32)
        nw = (pw + 63) // 64
        maskw = np.zeros((n, nw), dtype=np.uint64); maskw[:, 0] = mask
        for k, e_ in enumerate(extra):
            maskw[e_, (p + k) >> 6] |= np.uint64(1) << np.uint64((p + k) & 63)
        xw, ratiow, objw, stw = eng.pao_solve(np.ones(n, dtype=np.int64), a, np.zeros(n), pow_, pnw, np.arange(pw))
        xow, objow, itw, stow = orc.lad_solve(maskw, a, pw, np.full(pw, 1.05 * a.max()))
        assert stw == 0 and stow == 0 and objw == pytest.approx(objow, rel=1e-9) and objw <= obj * (1 + 1e-12)
        assert orc.lad_objective(maskw, a, xw) == pytest.approx(objow, rel=1e-9)
        assert np.all(xw >= -1e-12) and np.all(xw <= 1.05 * a.max() + 1e-9)


def test_pao_solve_wide_vs_highs_golden(eng, golden_dir):
    """More than 64 candidate columns (65 .. 256: the wide path, four mask words per node, rows grouped by a hash of the
    words and the grouping verified, W / G of the solver in global memory) on committed SciPy-HiGHS vectors: objective
    equal to 1e-9 relative, bounds and pinned columns respected.  The reference's matrix has no column cap
    (profile.rs:1333-1342)."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(golden_dir, "lp_wide_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub, objh = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i], float(z["obj_%d" % i])
        p = len(ub)
        po, pn = _paths_from_masks(mask, p)
        x, ratio, obj, st = eng.pao_solve(np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), po, pn, np.arange(p),
                                          fixed_zero=(ub == 0).astype(np.uint8))
        assert st == 0, (i, p)
        assert obj == pytest.approx(objh, rel=1e-9, abs=1e-12), (i, p)
        assert orc.lad_objective(mask, a, x) == pytest.approx(objh, rel=1e-9, abs=1e-12), (i, p)
        assert np.all(x >= 0) and np.all(x <= ub + 1e-12)


def test_pao_solve_huge_vs_highs_golden(eng, golden_dir):
    """More than 256 candidate columns (257 .. 1100; the reference's dense matrix has no cap, profile.rs:1333-1342): mask words,
    basis inverse and column state of the solver sized at run time, on committed SciPy-HiGHS vectors -- objective equal to 1e-9
    relative, bounds and pinned columns respected."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(golden_dir, "lp_huge_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub, objh = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i], float(z["obj_%d" % i])
        p = len(ub)
        po, pn = _paths_from_masks(mask, p)
        x, ratio, obj, st = eng.pao_solve(np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), po, pn, np.arange(p),
                                          fixed_zero=(ub == 0).astype(np.uint8))
        assert st == 0, (i, p)
        assert obj == pytest.approx(objh, rel=1e-9, abs=1e-12), (i, p)
        assert orc.lad_objective(mask, a, x) == pytest.approx(objh, rel=1e-9, abs=1e-12), (i, p)
        assert np.all(x >= 0) and np.all(x <= ub + 1e-12)


def test_wide_lp_solution_vector_where_the_optimum_determines_it(eng, golden_dir):
    """65 .. 130 candidate columns (the wide path): not only the objective but x itself -- first_sol per strain -- against
    SciPy-HiGHS on every column the optimal face pins down (tests/golden/lp_wide_unique_cases.npz from
    oracle/gen_golden_wide_unique.py: per column min and max over the optimal face agree to 1e-8)."""
    import os
    from oracle import oracle as orc
    z = np.load(os.path.join(golden_dir, "lp_wide_unique_cases.npz"))
    for i in range(int(z["n_cases"])):
        mask, a, ub, objh, xh, det = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i], float(z["obj_%d" % i]), z["x_%d" % i], z["determined_%d" % i]
        p = len(ub)
        assert det.sum() >= p // 2
        po, pn = _paths_from_masks(mask, p)
        x, ratio, obj, st = eng.pao_solve(np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), po, pn, np.arange(p))
        assert st == 0, (i, p)
        assert obj == pytest.approx(objh, rel=1e-9, abs=1e-12), (i, p)
        assert np.allclose(x[det], xh[det], rtol=1e-6, atol=1e-7), (i, p, np.abs(x - xh)[det].max())
        xo, objo, _, sto = orc.lad_solve(mask, a, p, ub)                      # the checker itself, on the same columns
        assert sto == 0 and np.allclose(xo[det], xh[det], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n_walks", [35, 130, 350])
def test_species_with_more_than_64_candidates(eng, n_walks):
    """A species whose first filter leaves more than 64 columns (no unique trio: every walk exists twice, so every haplotype
    is a candidate, profile.rs:1208).  70 columns: solved on the wide path (four mask words), objective == the oracle's (the LP is
    degenerate by construction -- twin columns -- so x is compared through the objective and the twin sums).  260 and 700 columns:
    more than 256 haplotypes, the path without a cap (the reference has none, profile.rs:1333-1342): mask words, basis inverse and
    column state sized at run time, same checks.  The species next to it is unaffected either way."""
    from oracle import oracle as orc
    import synthdata as synth
    from pantax_amd.engine import metrics_to_dicts
    nh = 2 * n_walks
    sset = synth.make_set(70, 2, 4, 20000, 20000, present_frac=0.6)
    g = sset.species[0]
    rng = np.random.default_rng(71)
    base = [g.path_nodes[int(g.path_off[h]):int(g.path_off[h + 1])] for h in range(g.n_paths)]
    walks = []
    for k in range(n_walks):                  # different walks, each twice -> no trio occurs once
        w = base[k % len(base)].copy()
        cut = sorted(rng.integers(3, len(w) - 3, size=2))
        w = np.concatenate([w[:cut[0]], w[cut[1]:]]) if k >= len(base) else w
        walks += [w, w]
    g.path_nodes = np.concatenate(walks).astype(np.uint32)
    g.path_off = np.concatenate([[0], np.cumsum([len(w) for w in walks])]).astype(np.uint64)
    g.hap_names = ["GCF_9%05d.1" % i for i in range(nh)]
    g.genome_len = np.array([int(g.node_len[w].sum()) for w in walks], dtype=np.int64)
    g.truth_depth = np.zeros(nh)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, (rc, bs, lm, uq), sset.avg_len())
    assert keep.all()
    abc, hap, ln, hto = eng.trio_nodes_info()
    assert hto[nh] == 0                        # no unique trio in species 0
    eng.get_node_abundances(fetch=False)
    met, info = eng.strain_profiling(absolute, species_active=keep)
    got = metrics_to_dicts(met, eng.H)
    ref = _oracle_cov_per_species(sset, sp)
    G, T, b, c, t, na = ref[0]
    assert info[0].n_candidates == nh
    rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
    assert rc_ == 0 and nc == nh and info[0].status1 == 0 and info[0].status2 == 0
    assert info[0].obj1 == pytest.approx(o1, rel=1e-9)          # no second solve in this branch (profile.rs:1279-1283)
    orc.abundance_constraint(absolute[0], omet)
    od = orc.metrics_to_dicts(omet)
    for k in range(n_walks):               # twins share their walk: only the sum of the pair is determined ...
        for key in ("path_base_cov",):
            assert got[2 * k][key] == pytest.approx(od[2 * k][key], rel=1e-6) and got[2 * k + 1][key] == pytest.approx(od[2 * k + 1][key], rel=1e-6)
    # ... and not even that where different walks cover the same nodes; the objective above and the LP's own value agree
    mask, ratio = orc.path_masks(G, np.arange(nh), c)
    ab = np.asarray(b, dtype=np.float64) / np.asarray(sset.species[0].node_len, dtype=np.float64)   # profile.rs:980-990
    x1 = np.array([got[h]["first_sol"] for h in range(nh)])
    assert orc.lad_objective(mask, ab, x1) == pytest.approx(o1, rel=1e-9)
    G, T, b, c, t, na = ref[1]
    rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
    assert rc_ == 0 and info[1].status1 == 0 and info[1].n_candidates == nc
    orc.abundance_constraint(absolute[1], omet)
    for h, e in enumerate(orc.metrics_to_dicts(omet)):
        for key, ev in e.items():
            gv = got[nh + h][key]
            if ev is None or gv is None or isinstance(ev, bool):
                assert gv == ev, (h, key, gv, ev)
            else:
                assert gv == pytest.approx(ev, rel=1e-7, abs=1e-9), (h, key, gv, ev)


def test_pao_solve_edge_cases(eng):
    # no covered node: x = 0
    po, pn = _paths_from_masks(np.array([1, 3], dtype=np.uint64), 2)
    x, ratio, obj, st = eng.pao_solve([1, 1], [0.0, 0.0], [0, 0], po, pn, [0, 1])
    assert st == 0 and x.tolist() == [0.0, 0.0] and obj == 0.0
    # upper bound 1.05*max active: rows (x=5),(x=6),(x=7) plus a far row through path 1 only
    mask = np.array([1, 1, 1, 2], dtype=np.uint64)
    a = np.array([50.0, 60.0, 70.0, 1.0])
    po, pn = _paths_from_masks(mask, 2)
    x, ratio, obj, st = eng.pao_solve([1] * 4, a, [1, 0, 1, 1], po, pn, [0, 1])
    assert st == 0 and x[0] == pytest.approx(60.0) and x[1] == pytest.approx(1.0)
    assert ratio[0] == pytest.approx(2.0 / 3.0) and ratio[1] == pytest.approx(1.0)
    # noise-free integer data with massive ties: exact recovery, objective 0
    rng = np.random.default_rng(3)
    n, p = 4000, 5
    mask = rng.integers(1, 1 << p, size=n).astype(np.uint64)
    truth = np.array([4.0, 0.0, 9.0, 0.0, 2.0])
    A = np.stack([((mask >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1)
    po, pn = _paths_from_masks(mask, p)
    x, ratio, obj, st = eng.pao_solve(np.ones(n, dtype=np.int64), A @ truth, np.zeros(n), po, pn, np.arange(p))
    assert st == 0 and obj == pytest.approx(0.0, abs=1e-9) and np.allclose(x, truth, atol=1e-8)


def test_path_cov_ratio_beyond_f32_integer_range(eng):
    """path_cov_ratio (profile.rs:1344-1357) once a path holds more than 2^24 bases: the reference accumulates both sums in f32
    (in whatever order nalgebra's product walks the column), so from there on its last digits depend on the summation order.
    This build forms both sums exactly and divides once in f32: it equals the correctly rounded quotient of the f32-rounded
    exact sums, and the oracle's sequential f32 accumulation agrees to a few units of f32 precision -- the documented
    divergence of INTEGRATION.md section 5.  Below 2^24 the two are identical (every other test)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(17)
    V, p = 6000, 3
    node_len = rng.integers(2000, 9000, size=V).astype(np.int64)          # ~3.3e7 bases per path: beyond 2^24 = 1.68e7
    cov = (node_len * rng.random(V)).astype(np.uint64)
    mask = rng.integers(1, 1 << p, size=V).astype(np.uint64)
    mask[:10] = (1 << p) - 1
    po, pn = _paths_from_masks(mask, p)
    ab = cov / node_len
    x, ratio, obj, st = eng.pao_solve(node_len, ab, cov, po, pn, np.arange(p))
    assert st == 0
    G = orc.Graph(node_len, po, pn)
    _, ratio_orc = orc.path_masks(G, np.arange(p), cov)
    for k in range(p):
        on = ((mask >> np.uint64(k)) & np.uint64(1)).astype(bool)
        s_cov, s_len = int(cov[on].sum()), int(node_len[on].sum())
        assert s_len > (1 << 24)
        assert np.float32(ratio[k]) == np.float32(s_cov) / np.float32(s_len)                 # exact sums, one f32 divide
        assert abs(float(ratio[k]) - s_cov / s_len) <= 2.0 ** -23 * (s_cov / s_len)           # within one f32 ulp of the true ratio
        assert abs(float(ratio_orc[k]) - float(ratio[k])) <= 64 * 2.0 ** -24 * float(ratio[k])   # the f32 running sums drift, a little


@pytest.mark.parametrize("seed,S,H,R,L,pf,opts", [
    (21, 2, 6, 30000, 40000, 0.5, {}), (22, 4, 10, 80000, 30000, 0.4, {}), (23, 3, 5, 20000, 30000, 0.2, {}),
    # the option branches of first_filter_paths / second_filter_paths (profile.rs:1080-1285, main.rs:108-124)
    (22, 4, 10, 80000, 30000, 0.4, dict(shift=True)),                       # --shift: coverage-dependent fraction threshold (:1140-1165)
    (22, 4, 10, 80000, 30000, 0.4, dict(fr=0.5, fc=0.2, sr=0.4)),            # long-read fr, tighter divergence cut, looser rescue
    (24, 3, 6, 6000, 30000, 0.6, dict(shift=True, fr=0.5)),                  # low depth: the shifted threshold drops below fr
    (23, 3, 5, 20000, 30000, 0.2, dict(min_depth=3)),                        # --min_depth feeds the single-path statistics only (:2941-2944)
    (25, 4, 1, 20000, 30000, 1.0, dict(min_depth=2)),                        # every species a single strain
    (26, 2, 40, 120000, 30000, 0.9, dict(fr=0.05)),                          # 30-40 LP columns per species, thousands of membership patterns
    # more than 64 LP columns per species: the wide path (four mask words per node, W / G of the solver in global memory)
    (27, 2, 100, 300000, 30000, 0.9, dict(fr=0.05)),                         # 80-100 columns in both species
    (28, 3, 150, 400000, 20000, 0.8, dict(fr=0.05)),                         # wide species next to a single-strain one in one batch
    (29, 2, 100, 150000, 30000, 0.3, dict(fr=0.2)),                          # 100 haplotypes, fewer than 64 pass the first filter: one-word path
])
def test_strain_profiling_vs_oracle(eng, seed, S, H, R, L, pf, opts):
    """optimize_otu + abundace_constraint (profile.rs:2884-3070) for every species, against the oracle."""
    from oracle import oracle as orc
    import synthdata as synth
    from pantax_amd.engine import metrics_to_dicts
    sset = synth.make_set(seed, S, H, R, L, present_frac=pf, single_strain_every=3 if S >= 3 else 0)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, (rc, bs, lm, uq), sset.avg_len())
    eng.trio_nodes_info(fetch=False)
    eng.get_node_abundances(fetch=False)
    met, info = eng.strain_profiling(absolute, species_active=keep, **opts)
    got = metrics_to_dicts(met, eng.H)
    ref = _oracle_cov_per_species(sset, sp)
    n_cols = n_face = 0
    for si, (G, T, b, c, t, na) in enumerate(ref):
        if not keep[si]:
            continue
        rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t, **opts)
        assert rc_ == 0
        orc.abundance_constraint(absolute[si], omet)
        exp = orc.metrics_to_dicts(omet)
        h0 = int(eng.hap_off[si])
        assert info[si].n_candidates == nc and info[si].status1 == 0 and info[si].status2 == 0
        n_cols += nc
        if nc:
            assert info[si].obj1 == pytest.approx(o1, rel=1e-9, abs=1e-12)
            if not np.isnan(o2):
                assert info[si].obj2 == pytest.approx(o2, rel=1e-9, abs=1e-12)
        # an LP whose optimum is a face, not a point (likely with tens of columns): any x with the optimal objective is
        # right and what follows from first_sol follows it -- the objective was compared above, x is checked to attain it
        cand = [h for h, e in enumerate(exp) if e["first_sol"] is not None]
        if nc and any(got[h0 + h]["first_sol"] != pytest.approx(exp[h]["first_sol"], rel=1e-7, abs=1e-9) for h in cand):
            mask, _ = orc.path_masks(G, cand, c)
            xg = np.array([got[h0 + h]["first_sol"] for h in cand])
            g_ = sset.species[si]
            assert orc.lad_objective(mask, b / np.asarray(g_.node_len, dtype=np.float64), xg) == pytest.approx(o1, rel=1e-9), (si, "x is not optimal")
            n_face += 1
            continue
        for h, e in enumerate(exp):
            g = got[h0 + h]
            for key, ev in e.items():
                gv = g[key]
                if ev is None or gv is None or isinstance(ev, bool):
                    assert gv == ev, (si, h, key, gv, ev)
                else:
                    assert gv == pytest.approx(ev, rel=1e-7, abs=1e-9), (si, h, key, gv, ev)
    assert n_cols > 0 and (n_face == 0 or H >= 20)      # the small cases of this test have unique optima


@pytest.mark.gpu
@pytest.mark.parametrize("seed,S,H,R,L,pf", [(31, 5, 10, 90000, 30000, 0.4), (32, 3, 30, 90000, 20000, 0.6)])
def test_hap_trio_statistics_by_key_whatever_the_row_order(eng, seed, S, H, R, L, pf, set_opt):
    """a9's per-haplotype statistics of the unique-trio abundances (count of non-zero ones, z-score-filtered mean, profile.rs:1028-1147) are
    taken BY KEY over rows numbered in filing order (round 5).  The two routes that file rows -- from the visit kernel's records, and by the
    pass over the walks (option trio_rows=path) -- number them differently, so the sums run in different fixed orders: the metrics agree to
    rounding, the decisions exactly; the same route twice gives the same bits (every sum has a fixed order)."""
    from oracle import oracle as orc
    import synthdata as synth
    from pantax_amd.engine import metrics_to_dicts
    sset = synth.make_set(seed, S, H, R, L, present_frac=pf)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, (rc, bs, lm, uq), sset.avg_len())
    outs = []
    for mode in (None, "path", None):
        set_opt(eng, "trio_rows", mode)
        eng.db_reset()
        eng.trio_nodes_info(fetch=False)
        eng.get_node_abundances(fetch=False)
        met, info = eng.strain_profiling(absolute, species_active=keep)
        outs.append((metrics_to_dicts(met, eng.H), [(i.n_candidates, i.status1, i.status2, i.n_rows, i.n_patterns) for i in info]))
    assert outs[0] == outs[2]
    assert outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][0], outs[1][0]):
        for k in a:
            if a[k] is None or isinstance(a[k], bool):
                assert a[k] == b[k], k
            else:
                assert b[k] is not None and abs(a[k] - b[k]) <= 1e-12 * max(1.0, abs(a[k])), (k, a[k], b[k])


def _same_infos(a, b):
    """solve infos of two row pipelines: candidates, statuses, pivots, row and pattern counts equal; the objectives equal to the last bits
    of a double (the many-species step sums them over the sorted rows, the other pipelines over the nodes: another order of additions)"""
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[:7] == y[:7], (x, y)
        for u, v in zip(x[7:], y[7:]):
            assert (np.isnan(u) and np.isnan(v)) or abs(u - v) <= 1e-12 * max(1.0, abs(u), abs(v)), (x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,S,H,R,L,pf,opts", [
    (22, 4, 10, 80000, 30000, 0.4, {}),                        # a handful of patterns per species
    (26, 2, 40, 120000, 30000, 0.9, dict(fr=0.05)),            # thousands of patterns
    (28, 3, 150, 400000, 20000, 0.8, dict(fr=0.05)),           # a wide species in the batch
])
def test_row_pipelines_agree(eng, seed, S, H, R, L, pf, opts, set_opt):
    """The four row pipelines of lad_prepare -- whole-batch sample sort (small inputs), whole-batch radix sort, the node-order
    compaction + batched per-species sort (round 3's many-species step) and the batched sort straight from the node arrays (the
    many-species step now; both forced here at a small size) -- and the two ways
    of building the membership masks (by node from the node -> haplotypes table of the upload, or by walking the candidates'
    paths, PANTAX_MASK=walk) give the same metrics, iteration counts, row and pattern counts bit for bit, and the same objectives (to 1e-12: the sort straight
    from the nodes sums them over the sorted rows)."""
    from oracle import oracle as orc
    import synthdata as synth
    sset = synth.make_set(seed, S, H, R, L, present_frac=pf, single_strain_every=3 if S >= 3 else 0)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, (rc, bs, lm, uq), sset.avg_len())
    eng.trio_nodes_info(fetch=False)
    eng.get_node_abundances(fetch=False)
    outs = []
    for sort, maskmode in ((None, None), ("nodes", None), ("radix", None), (None, "walk"), ("nodes", "walk"), ("radix", "walk")):
        set_opt(eng, "row_sort", sort)
        set_opt(eng, "mask", maskmode)
        met, info = eng.strain_profiling(absolute, species_active=keep, **opts)
        outs.append((np.frombuffer(bytes(memoryview(met)), dtype=np.uint8).copy(),
                     [(i.n_candidates, i.status1, i.status2, i.iters1, i.iters2, i.n_rows, i.n_patterns, i.obj1, i.obj2) for i in info]))
    assert any(o[5] > 0 for o in outs[0][1])
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0])
        _same_infos(o[1], outs[0][1])


@pytest.mark.gpu
def test_species_of_a_million_nodes_keeps_the_batched_row_sort(eng, set_opt):
    """No limit on a species' size in the many-species row sort (round 4: a graph of more than 600 000 nodes used to send the whole
    batch through the radix sort): two species of 1.75e6 nodes each -- buckets of more than a thousand rows, the second wave kernel and the LDS
    network in use -- give the same step output, objectives, row and pattern counts as the radix pipeline."""
    import synthdata as synth
    sset = synth.NativeSet(20260777, 2, 4, 600_000, 28_000_000, present_frac=0.75, threads=8).make()
    assert min(len(g.node_len) for g in sset.species) > 600_000
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    outs = []
    for sort in (None, "radix"):
        set_opt(eng, "row_sort", sort)
        out = eng.profile_step(sset.avg_len())
        outs.append((out[0].copy(), bytes(out[2]), [(i.n_candidates, i.status1, i.status2, i.iters1, i.iters2, i.n_rows, i.n_patterns, i.obj1, i.obj2) for i in out[3]]))
    assert any(o[5] > 100_000 for o in outs[0][2])
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    _same_infos(outs[0][2], outs[1][2])


@pytest.mark.gpu
def test_trio_tables_fetched_after_a_step_are_the_stage_call_tables(eng):
    """A step's rebuild of the unique-trio index leaves out the row-order export copies (key, owner haplotype); trio_get
    after a step rebuilds with them: same tables as the stage call gave before the step, the coverage results of the step
    stay valid, and the next step is unaffected."""
    import synthdata as synth
    sset = synth.make_set(61, 3, 5, 20000, 30000, present_frac=0.6)
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    abc0, hap0, ln0, hto0 = eng.trio_nodes_info()
    out1 = eng.profile_step(sset.avg_len())
    met1 = bytes(out1[2]); keep1 = out1[0].copy()
    abc1, hap1, ln1, hto1 = eng.trio_nodes_info()
    assert np.array_equal(abc0, abc1) and np.array_equal(hap0, hap1) and np.array_equal(ln0, ln1) and np.array_equal(hto0, hto1)
    bases, cov, tb, nab = eng.get_node_abundances()           # recomputed or still valid: either way the same numbers as in the step
    out2 = eng.profile_step(sset.avg_len())
    assert bytes(out2[2]) == met1 and np.array_equal(out2[0], keep1)
    out3 = eng.profile_step(sset.avg_len(), rebuild_trio=False)
    assert bytes(out3[2]) == met1


@pytest.mark.gpu
@pytest.mark.parametrize("seed,S,H,every", [(81, 6, 12, 0), (82, 5, 70, 2), (83, 3, 1, 0)])
def test_rebuild_in_one_pass_files_the_rows_of_the_first_build(eng, seed, S, H, every, set_opt):
    """The first index build of a db goes visit kernel -> records -> prefix of the groups' counts -> rows kernel and keeps the groups' offsets with
    the visit table; every later build (a step's rebuild, db_reset + trio_index) decides and files in ONE kernel by those offsets
    (trio_file_kernel).  Same tables, same step, with and without the window starts (export copies), on a db of one route and of both; the
    option trio_two_pass sends every build down the first build's road."""
    import synthdata as synth
    kw = dict(single_strain_every=every) if every else {}
    sset = synth.make_set(seed, S, H, 40000, 12000, present_frac=0.5, **kw)
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    first = eng.trio_nodes_info()                       # first build (with the export copies)
    eng.db_reset()
    again = eng.trio_nodes_info()                       # one pass, with the window starts
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    step1 = eng.profile_step(sset.avg_len())            # one pass, without
    after = eng.trio_nodes_info()
    for a, b in zip(first, after):
        assert np.array_equal(a, b)
    set_opt(eng, "trio_two_pass", 1)
    step2 = eng.profile_step(sset.avg_len())
    two = eng.trio_nodes_info()
    set_opt(eng, "trio_two_pass", None)
    for a, b in zip(first, two):
        assert np.array_equal(a, b)
    assert bytes(step1[2]) == bytes(step2[2]) and np.array_equal(step1[0], step2[0])
    step3 = eng.profile_step(sset.avg_len())
    assert bytes(step3[2]) == bytes(step1[2])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,S,H,every", [(71, 4, 80, 2), (72, 5, 70, 3)])
def test_step_on_a_mixed_database(eng, seed, S, H, every, set_opt):
    """A database that holds species of BOTH kinds -- single-strain species (the visit table's) beside species of 70-80 strains (a node with
    more than 64 visits: node-block kernel).  The step's rebuild files the first kind's lookup rows from the visit kernel's records and the
    second kind's by the pass over their walks, behind them; the stage calls (which want the export copies) take the pass over the walks for
    the whole db.  Both give the same per-haplotype metrics, and those of the step with the rows forced through the walks (PANTAX_TRIO_ROWS=path);
    the integers of the stage calls are checked against the oracle."""
    from oracle import oracle as orc
    import synthdata as synth
    from tests.helpers import select_reads
    sset = synth.make_set(seed, S, H, 60000, 12000, present_frac=0.3, single_strain_every=every)
    assert any(g.n_paths == 1 for g in sset.species) and any(np.bincount(g.path_nodes).max() > 64 for g in sset.species)   # both kinds
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    out_mixed = eng.profile_step(sset.avg_len())
    set_opt(eng, "trio_rows", "path")
    out_path = eng.profile_step(sset.avg_len())
    set_opt(eng, "trio_rows", None)
    assert bytes(out_mixed[2]) == bytes(out_path[2]) and np.array_equal(out_mixed[0], out_path[0])
    # stage calls on the same db: trio tables and coverage against the oracle, species by species
    sp, *_ = eng.rcls_profile()
    eng.db_reset()
    abc, hap, ln, hto = eng.trio_nodes_info()
    bases, cov, tb, nab = eng.get_node_abundances()
    hb = np.cumsum([0] + [g.n_paths for g in sset.species])
    for si, g in enumerate(sset.species):
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(sset.reads, np.nonzero(sp == si)[0])
        b, c, t, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        lo, hi = int(eng.node_off[si]), int(eng.node_off[si + 1])
        u0, u1 = int(hto[hb[si]]), int(hto[hb[si + 1]])
        assert np.array_equal(bases[lo:hi], b) and np.array_equal(cov[lo:hi], c)
        assert u1 - u0 == T.n_unique and np.array_equal(abc[u0:u1], T.abc) and np.array_equal(tb[u0:u1], t)
    out_again = eng.profile_step(sset.avg_len())
    assert bytes(out_again[2]) == bytes(out_mixed[2])


@pytest.mark.gpu
@pytest.mark.parametrize("S", [1000, 1024, 1025, 1500])
def test_a_thousand_species_in_one_step(eng, S):
    """Many species on one device (BASELINE configs[3] has 1000): the binning kernel keeps its range tables and counters in
    LDS up to 1024 species (44 KB at 1000) and in memory beyond; every per-species launch geometry (statistics chunks, LP
    workgroups, segmented row sort) is exercised with S in the thousands.  Species decisions and counters bit-exact, the step's
    strain metrics against the oracle for a sample of species."""
    from oracle import oracle as orc
    import synthdata as synth
    from pantax_amd.engine import metrics_to_dicts
    sset = synth.make_set(4000 + S, S, 2, 150000, 3000, single_strain_every=4, present_frac=0.7)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    rs, re_ = [g.range_start for g in sset.species], [g.range_end for g in sset.species]
    assert np.array_equal(sp, orc.bin_reads(rd.step_off, rd.node_id, rs, re_))
    counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
    assert all(np.array_equal(a, b) for a, b in zip((rc, bs, lm, uq), counts))
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, (rc, bs, lm, uq), sset.avg_len())
    k2, a2, met, info, passed, sa, spp = eng.profile_step(sset.avg_len())
    assert np.array_equal(k2, keep) and np.allclose(a2, absolute, rtol=1e-12, atol=0)
    got = metrics_to_dicts(met, eng.H)
    hb = np.cumsum([0] + [g.n_paths for g in sset.species])
    checked = 0
    for si in list(range(0, S, 97)) + [S - 1]:
        if not keep[si]:
            continue
        g = sset.species[si]
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes); T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(rd, np.nonzero(sp == si)[0])
        b, c, t, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
        assert rc_ == 0 and info[si].n_candidates == nc and info[si].status1 == 0
        if nc:
            assert info[si].obj1 == pytest.approx(o1, rel=1e-9, abs=1e-12)
        orc.abundance_constraint(absolute[si], omet)
        for h, e in enumerate(orc.metrics_to_dicts(omet)):
            for key, ev in e.items():
                gv = got[hb[si] + h][key]
                if ev is None or gv is None or isinstance(ev, bool):
                    assert gv == ev, (si, h, key, gv, ev)
                else:
                    assert gv == pytest.approx(ev, rel=1e-7, abs=1e-9), (si, h, key, gv, ev)
        checked += 1
    assert checked >= 5


@pytest.mark.gpu
@pytest.mark.parametrize("sample_nodes", [2000, 500, 1237])
def test_strain_profiling_with_row_sampling(eng, sample_nodes):
    """--sample N (a11, profile.rs:1287-1295, :2738-2752): species with more valid rows than N solve the LP on the sampled
    rows only; the others are untouched.  Checked against the oracle's own restatement of the sampler and, for the rule
    itself, against the unsampled run."""
    from oracle import oracle as orc
    import synthdata as synth
    from pantax_amd.engine import metrics_to_dicts
    sset = synth.make_set(77, 3, 6, 60000, 40000, present_frac=0.5)
    rd = sset.reads
    eng.upload_db(sset.species)
    eng.upload_packed(rd)
    sp, rc, bs, lm, uq = eng.rcls_profile()
    keep, absolute, abundance = orc.species_profile(sp, rd.qlen, (rc, bs, lm, uq), sset.avg_len())
    eng.trio_nodes_info(fetch=False)
    eng.get_node_abundances(fetch=False)
    met0, info0 = eng.strain_profiling(absolute, species_active=keep)
    n_full = [info0[s].n_rows for s in range(eng.S)]
    met, info = eng.strain_profiling(absolute, species_active=keep, sample_nodes=sample_nodes)
    got = metrics_to_dicts(met, eng.H)
    ref = _oracle_cov_per_species(sset, sp)
    sampled = 0
    for si, (G, T, b, c, t, na) in enumerate(ref):
        if not keep[si]:
            continue
        rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t, sample_nodes=sample_nodes)
        assert rc_ == 0
        orc.abundance_constraint(absolute[si], omet)
        exp = orc.metrics_to_dicts(omet)
        h0 = int(eng.hap_off[si])
        assert info[si].n_candidates == nc and info[si].status1 == 0 and info[si].status2 == 0
        if nc:
            assert info[si].n_rows == min(n_full[si], sample_nodes)
            sampled += n_full[si] > sample_nodes
            assert info[si].obj1 == pytest.approx(o1, rel=1e-9, abs=1e-12)
            if not np.isnan(o2):
                assert info[si].obj2 == pytest.approx(o2, rel=1e-9, abs=1e-12)
        for h, e in enumerate(exp):
            g = got[h0 + h]
            for key, ev in e.items():
                gv = g[key]
                if ev is None or gv is None or isinstance(ev, bool):
                    assert gv == ev, (si, h, key, gv, ev)
                else:
                    assert gv == pytest.approx(ev, rel=1e-7, abs=1e-9), (si, h, key, gv, ev)
    assert sampled >= 1
    # a limit nobody reaches changes nothing
    met1, info1 = eng.strain_profiling(absolute, species_active=keep, sample_nodes=10**7)
    assert metrics_to_dicts(met1, eng.H) == metrics_to_dicts(met0, eng.H)


def _host_sorted(k0, k1, k2):
    order = np.lexsort((k2, k1, k0))
    return [k0[order], k1[order], k2[order]]


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [1, 2])
def test_device_sorts_against_host_sort(eng, algo):
    """Both sorts of the LP row grouping (LSD radix, sample sort) against numpy on crafted inputs: sizes around
    every internal boundary, massive ties (single-key buckets), presorted / reversed input, and inputs whose
    sampled rows are unrepresentative so that one bucket exceeds the LDS capacities (1024 / 4096 rows)."""
    rng = np.random.default_rng(99)
    cases = []
    for n in (1, 2, 63, 4095, 4096, 4097, 5000, 70000):
        cases.append((rng.integers(0, 3, n), rng.integers(0, 8, n), rng.integers(0, 2 ** 63, n)))
    n = 200000
    vals = rng.integers(0, 50, n)                                   # 50 distinct keys: every bucket is a single-key bucket
    cases.append((np.zeros(n, np.uint64), rng.integers(0, 2, n), vals))
    cases.append((np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.sort(rng.integers(0, 2 ** 62, n))))         # presorted
    cases.append((np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.sort(rng.integers(0, 2 ** 62, n))[::-1]))   # reversed
    for n in (8192, 12288, 20000):
        # rows 0, n/4096, 2n/4096, ... are the sample: give them small keys, everything else large distinct keys, so
        # that every other row lands above the last splitter (one bucket of n - 4096 rows)
        k2 = rng.permutation(n).astype(np.uint64) + np.uint64(1 << 40)
        pos = (np.arange(4096, dtype=np.uint64) * np.uint64(n)) // np.uint64(4096)
        k2[pos] = np.arange(4096, dtype=np.uint64)
        cases.append((np.zeros(n, np.uint64), np.zeros(n, np.uint64), k2))
    for k0, k1, k2 in cases:
        k0, k1, k2 = (np.asarray(x, dtype=np.uint64) for x in (k0, k1, k2))
        got = eng.sort_rows(k0, k1, k2, algo=algo)
        exp = _host_sorted(k0, k1, k2)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e), (algo, len(k0))


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [4, 5])
def test_node_order_sample_sort_against_host_sort(eng, algo):
    """The row sort of the many-species step (sample_sort_nodes.hip) through the host-buffer utility: it reads NODE arrays -- an entry
    with an empty mask or without a positive abundance is no row and drops out -- and leaves the rows of all segments back to back.
    Segments of every size class side by side: empty of rows, 1 row, <= 4096 nodes (sorted by the sample kernel), tens of
    thousands, massive ties (single-key buckets), presorted / reversed, one mask (only the abundance moves through the register
    network) and many, a segment of three million nodes (no limit on a species' size: its buckets hold thousands of rows), and
    unrepresentative samples whose one bucket exceeds the wave (512), the second wave kernel (1024) and the LDS (4096) capacities --
    the last kind sorted in place through memory.  algo 5: the segment number packed into the mask word, as the step does when the
    bits fit."""
    rng = np.random.default_rng(11)
    segs = []
    for n in (1, 2, 63, 64, 65, 4095, 4096, 4097, 5000, 70000):
        segs.append((rng.integers(0, 8, n), rng.integers(0, 2 ** 62, n)))                          # mask 0 = no row, 1 in 8
    segs.append((np.zeros(300, np.uint64), rng.integers(1, 2 ** 62, 300)))                        # a small segment without rows
    segs.append((np.zeros(9000, np.uint64), rng.integers(1, 2 ** 62, 9000)))                      # a large one without rows
    n = 150000
    segs.append((rng.integers(0, 3, n), rng.integers(0, 50, n)))                                  # 100 distinct keys, abundance 0 = no row
    segs.append((np.ones(n, np.uint64), np.sort(rng.integers(1, 2 ** 62, n))))                    # presorted, one mask
    segs.append((np.full(n, 7, np.uint64), np.sort(rng.integers(1, 2 ** 62, n))[::-1]))           # reversed
    a = rng.integers(1, 2 ** 62, n).astype(np.uint64)
    a[rng.random(n) < 0.5] |= np.uint64(1 << 63)                                                  # negative doubles: no rows
    segs.append((rng.integers(1, 200, n), a))
    segs.append((rng.integers(1, 4, 3000000), rng.integers(1, 2 ** 62, 3000000)))                # millions of nodes: buckets of thousands of rows, many above 4096
    for n, extra in ((8192, 0), (20000, 0), (60000, 0), (6000, 700), (9000, 2500)):
        # nodes 0, n/4096, 2n/4096, ... are the sample: small keys there, large distinct keys elsewhere -> one bucket holds the rest
        k2 = rng.permutation(n).astype(np.uint64) + np.uint64(1 << 40)
        pos = (np.arange(4096, dtype=np.uint64) * np.uint64(n)) // np.uint64(4096)
        k2[pos] = np.arange(1, 4097, dtype=np.uint64)
        m = np.ones(n, np.uint64)
        if extra:                                                                                # ... of about `extra` rows: the others are no rows
            rest = np.setdiff1d(np.arange(n), pos.astype(np.int64))
            m[rest[extra:]] = 0
        segs.append((m, k2))
    k0 = np.concatenate([np.full(len(a), 3 * i + 1, dtype=np.uint64) for i, (a, b) in enumerate(segs)])
    k1 = np.concatenate([np.asarray(a, dtype=np.uint64) for a, b in segs])
    k2 = np.concatenate([np.asarray(b, dtype=np.uint64) for a, b in segs])
    got = eng.sort_rows(k0, k1, k2, algo=algo)
    row = (k1 != 0) & (k2 != 0) & (k2 < np.uint64(0x7FF0000000000000))
    exp = _host_sorted(k0[row], k1[row], k2[row])
    nv = int(row.sum())
    for g, e in zip(got, exp):
        assert np.array_equal(g[:nv], e)
        assert not g[nv:].any()


@pytest.mark.gpu
def test_concurrent_solver_calls_on_one_ctx(eng, golden_dir):
    """The reference calls its solver from rayon workers (profile.rs:3297-3304): pantax_hip_pao_solve from several host
    threads on ONE ctx is legal -- calls are serialised inside -- and every thread gets the answer of its own call."""
    import os
    import threading
    z = np.load(os.path.join(golden_dir, "lp_cases.npz"))
    cases = []
    for i in range(int(z["n_cases"])):
        mask, a, ub = z["mask_%d" % i], z["a_%d" % i], z["ub_%d" % i]
        p = len(ub)
        po, pn = _paths_from_masks(mask, p)
        cases.append((a, po, pn, p, (ub == 0).astype(np.uint8), float(z["obj_%d" % i])))
    results, errors = {}, []

    def work(tid):
        try:
            for rep in range(3):
                for ci, (a, po, pn, p, fz, objh) in enumerate(cases):
                    if (ci + tid) % 2:
                        continue
                    x, ratio, obj, st = eng.pao_solve(np.ones(len(a), dtype=np.int64), a, np.zeros(len(a), dtype=np.uint64), po, pn, np.arange(p), fixed_zero=fz)
                    results[(tid, rep, ci)] = (obj, objh, st)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) > 40
    for (tid, rep, ci), (obj, objh, st) in results.items():
        assert st == 0 and obj == pytest.approx(objh, rel=1e-9, abs=1e-12), (tid, rep, ci)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2, 4])
def test_step_vs_literal_python_restatement_of_the_strain_level(eng, k):
    """The HIP path against fixtures from oracle/ref_literal_strain.py -- the literal Python reading of rcls.rs:237-258 and
    profile.rs:208-349, 1028-1285, 1297-1511 (LP by SciPy-HiGHS), 2884-3070, 3167-3248, which shares no code with the C oracle:
    species of every read, the species table, every HapMetrics field of every haplotype (--shift, --filtered off, --min_depth,
    single-strain species, null MAPQ among the cases) and the final strain rows; the one-call step and the table code."""
    from pantax_amd.engine import metrics_to_dicts
    from pantax_amd.pipeline import StepConfig, profile_step
    from tests.helpers import check_metrics_against_literal, load_literal_strain_case
    j, sset = load_literal_strain_case(k)
    a, ex = j["args"], j["expect"]
    names = [g.name for g in sset.species]
    haps = [h for g in sset.species for h in g.hap_names]
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)
    sp, *_ = eng.rcls_profile()
    assert [names[i] if i >= 0 else "U" for i in sp] == ex["read_species"]
    keep, absolute, met, info, passed, s_all, s_pass = eng.profile_step(sset.avg_len(), fr=a["fr"], fc=a["fc"], sr=a["sr"], sd=a["sd"], min_cov=a["min_cov"],
                                                                          min_depth=a["min_depth"], shift=a["shift"], filtered=a["filtered"],
                                                                          solver_semantics=1 if a.get("solver") == "highs" else 0)   # (case 4: --solver highs)
    d = metrics_to_dicts(met, eng.H)
    tab = {r["species_taxid"]: r for r in ex["species_profile"]}
    for s, g in enumerate(sset.species):
        assert bool(keep[s]) == (g.name in tab)
        if g.name in tab:
            assert absolute[s] == pytest.approx(tab[g.name]["predicted_coverage"], rel=1e-12)
        e = ex["per_species"].get(g.name)
        if e is None:
            continue
        h0, h1 = int(eng.hap_off[s]), int(eng.hap_off[s + 1])
        assert info[s].status1 == 0 and info[s].status2 == 0 and info[s].n_candidates == e["n_candidates"]
        if e["obj1"] is not None:
            assert info[s].obj1 == pytest.approx(e["obj1"], rel=1e-9, abs=1e-12)
        if e["obj2"] is not None:
            assert info[s].obj2 == pytest.approx(e["obj2"], rel=1e-9, abs=1e-12)
        check_metrics_against_literal(e["metrics"], d[h0:h1], g.name)
    # the tables as the stage API builds them (species order of the species table, normalisers, filters, sort)
    cfg = StepConfig(fr=a["fr"], fc=a["fc"], sr=a["sr"], sd=a["sd"], min_species_abundance=a["min_species_abundance"], min_cov=a["min_cov"],
                     min_depth=a["min_depth"], shift=a["shift"], filtered=a["filtered"], solver_semantics=1 if a.get("solver") == "highs" else 0)
    species_rows, strain_rows, stats = profile_step(eng, names, haps, sset.avg_len(), cfg)
    if a.get("solver") == "highs":   # ... and the slice does matter in this fixture: Gurobi's reading keeps more strains
        import dataclasses
        _, rows_g, _ = profile_step(eng, names, haps, sset.avg_len(), dataclasses.replace(cfg, solver_semantics=0))
        assert len(rows_g) > len(strain_rows)
    assert [r[0] for r in species_rows] == [r["species_taxid"] for r in ex["species_profile"]]
    for r, e in zip(species_rows, ex["species_profile"]):
        assert r[1] == pytest.approx(e["predicted_abundance"], rel=1e-12) and r[2] == pytest.approx(e["predicted_coverage"], rel=1e-12)
    assert sorted((r[0], r[1]) for r in strain_rows) == sorted((e["species_taxid"], e["hap_id"]) for e in ex["final_rows"])
    exp = {(e["species_taxid"], e["hap_id"]): e for e in ex["final_rows"]}
    for r in strain_rows:
        e = exp[(r[0], r[1])]
        assert r[2] == pytest.approx(e["predicted_coverage"], rel=1e-7) and r[3] == pytest.approx(e["predicted_abundance"], rel=1e-7)
    assert [r[3] for r in strain_rows] == sorted((r[3] for r in strain_rows), reverse=True)
