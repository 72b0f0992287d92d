"""Deterministic synthetic PanTax inputs.

Generates, per species, a pangenome graph in the reference's `Graph` shape
(types.rs:51-55: node lengths + one node walk per haplotype), the three DB side
files' contents (species_range.txt, species_genomes_stats.txt, genomes_info.txt)
and error-free reads as a packed structure-of-arrays stream (the layout the HIP
path consumes) that can also be rendered as GAF text.  numpy only; no reference
code is involved.  `seed = 20260501 + cfg_index` by convention.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class SpeciesGraph:
    name: str                 # species taxid string
    node_len: np.ndarray      # int64 [V]
    path_off: np.ndarray      # uint64 [H+1]
    path_nodes: np.ndarray    # uint32 [P], local 0-based
    hap_names: List[str]      # byte-wise sorted (BTreeMap order)
    range_start: int          # first global node id (1-based)
    range_end: int            # last global node id
    genome_len: np.ndarray    # int64 [H]
    truth_depth: np.ndarray   # float64 [H] expected depth of each strain

    @property
    def n_nodes(self):
        return len(self.node_len)

    @property
    def n_paths(self):
        return len(self.hap_names)


@dataclass
class PackedReads:
    step_off: np.ndarray      # uint64 [R+1]
    node_id: np.ndarray       # uint32 [T] global ids (1-based, as in the GAF)
    strand: np.ndarray        # uint8 [T] 0 '>' 1 '<' (GAF rendering only)
    pstart: np.ndarray        # int64 [R] GAF col 8
    pend: np.ndarray          # int64 [R] GAF col 9
    qlen: np.ndarray          # int64 [R] GAF col 2
    mapq: np.ndarray          # int64 [R] GAF col 12
    plen: np.ndarray          # int64 [R] GAF col 7
    read_id: List[str] = field(default_factory=list)  # only filled for small sets

    @property
    def n_reads(self):
        return len(self.pstart)


@dataclass
class SyntheticSet:
    species: List[SpeciesGraph]
    reads: PackedReads

    def range_table(self):
        return [(g.name, g.range_start, g.range_end, 1 if g.n_paths > 1 else 0) for g in self.species]

    def avg_len(self):
        return np.array([g.genome_len.mean() for g in self.species], dtype=np.float64)


def _random_clades(rng, H):
    """Random binary tree over H strains -> list of clade bitmasks (internal nodes + leaves)."""
    clades = []

    def split(members):
        if len(members) <= 1:
            return
        rng.shuffle(members)
        k = int(rng.integers(1, len(members)))
        left, right = members[:k], members[k:]
        for part in (left, right):
            m = 0
            for h in part:
                m |= 1 << int(h)
            clades.append(m)
            split(list(part))

    split(list(range(H)))
    return clades


def make_species(rng, name, H, genome_len, range_start, hap_prefix, frac_snp=0.30, frac_acc=0.12,
                 mean_len=32, present_frac=0.2, depth_mu=np.log(8.0), depth_sigma=1.0):
    """One species graph. H == 1 uses fixed 1024-bp chunks like build_eq1.rs:26-36."""
    hap_names = sorted("%s%03d.1" % (hap_prefix, h) for h in range(H))
    if H == 1:
        n = max(3, int(np.ceil(genome_len / 1024)))
        node_len = np.full(n, 1024, dtype=np.int64)
        node_len[-1] = max(1, genome_len - 1024 * (n - 1))
        path_nodes = np.arange(n, dtype=np.uint32)
        path_off = np.array([0, n], dtype=np.uint64)
        glen = np.array([node_len.sum()], dtype=np.int64)
        depth = np.array([rng.lognormal(depth_mu, depth_sigma)])
        return SpeciesGraph(name, node_len, path_off, path_nodes, hap_names, range_start, range_start + n - 1, glen, depth)
    if H > 63:   # strain membership no longer fits one word: the same construction on a boolean matrix
        return _make_species_wide(rng, name, H, genome_len, range_start, hap_names, frac_snp, frac_acc, mean_len, present_frac, depth_mu, depth_sigma)
    clades = _random_clades(rng, H)
    full = (1 << H) - 1
    est_sites = int(genome_len / (mean_len * (1 - frac_snp) + frac_snp) * 1.25) + 16
    kind = rng.choice(3, size=est_sites, p=[1 - frac_snp - frac_acc, frac_snp, frac_acc])  # 0 core 1 snp 2 acc
    kind[0] = 0
    kind[-1] = 0
    seg_len = np.minimum(1024, 1 + rng.geometric(1.0 / mean_len, size=est_sites)).astype(np.int64)
    clade_pick = np.array(clades, dtype=np.uint64)[rng.integers(0, len(clades), size=est_sites)]
    # expand sites to nodes: core -> 1 node (all strains); snp -> 2 nodes (clade / complement); acc -> 1 node (clade)
    n_nodes_site = np.where(kind == 1, 2, 1)
    node_site = np.repeat(np.arange(est_sites), n_nodes_site)
    first_of_site = np.concatenate([[0], np.cumsum(n_nodes_site)[:-1]])
    is_second = np.arange(len(node_site)) - first_of_site[node_site]
    nk = kind[node_site]
    member = np.where(nk == 0, np.uint64(full), clade_pick[node_site])
    member = np.where((nk == 1) & (is_second == 1), np.uint64(full) ^ clade_pick[node_site], member).astype(np.uint64)
    nlen = np.where(nk == 1, 1, seg_len[node_site]).astype(np.int64)
    # trim to genome_len of strain 0's walk
    on0 = (member & np.uint64(1)) != 0
    cum0 = np.cumsum(np.where(on0, nlen, 0))
    cut = int(np.searchsorted(cum0, genome_len)) + 1
    cut = min(max(cut, 8), len(nlen))
    # keep whole sites
    while cut < len(nlen) and node_site[cut] == node_site[cut - 1]:
        cut += 1
    member, nlen = member[:cut], nlen[:cut]
    V = cut
    paths = []
    glen = np.zeros(H, dtype=np.int64)
    for h in range(H):
        sel = np.nonzero((member >> np.uint64(h)) & np.uint64(1))[0].astype(np.uint32)
        paths.append(sel)
        glen[h] = nlen[sel].sum()
    path_off = np.zeros(H + 1, dtype=np.uint64)
    path_off[1:] = np.cumsum([len(p) for p in paths])
    path_nodes = np.concatenate(paths).astype(np.uint32)
    n_present = max(1, int(round(present_frac * H)))
    present = rng.choice(H, size=n_present, replace=False)
    depth = np.zeros(H)
    depth[present] = rng.lognormal(depth_mu, depth_sigma, size=n_present)
    return SpeciesGraph(name, nlen, path_off, path_nodes, hap_names, range_start, range_start + V - 1, glen, depth)


def _make_species_wide(rng, name, H, genome_len, range_start, hap_names, frac_snp, frac_acc, mean_len, present_frac, depth_mu, depth_sigma):
    """make_species for more than 63 strains (membership as a nodes x strains boolean matrix instead of one word per node)."""
    clades = _random_clades(rng, H)                      # Python integers: any number of strains
    cb = np.zeros((len(clades), H), dtype=bool)
    for i, m in enumerate(clades):
        for h in range(H):
            cb[i, h] = (m >> h) & 1
    est_sites = int(genome_len / (mean_len * (1 - frac_snp) + frac_snp) * 1.25) + 16
    kind = rng.choice(3, size=est_sites, p=[1 - frac_snp - frac_acc, frac_snp, frac_acc])  # 0 core 1 snp 2 acc
    kind[0] = 0
    kind[-1] = 0
    seg_len = np.minimum(1024, 1 + rng.geometric(1.0 / mean_len, size=est_sites)).astype(np.int64)
    clade_pick = rng.integers(0, len(clades), size=est_sites)
    n_nodes_site = np.where(kind == 1, 2, 1)
    node_site = np.repeat(np.arange(est_sites), n_nodes_site)
    first_of_site = np.concatenate([[0], np.cumsum(n_nodes_site)[:-1]])
    is_second = np.arange(len(node_site)) - first_of_site[node_site]
    nk = kind[node_site]
    member = cb[clade_pick[node_site]]
    member[nk == 0] = True
    flip = (nk == 1) & (is_second == 1)
    member[flip] = ~member[flip]
    nlen = np.where(nk == 1, 1, seg_len[node_site]).astype(np.int64)
    cum0 = np.cumsum(np.where(member[:, 0], nlen, 0))
    cut = int(np.searchsorted(cum0, genome_len)) + 1
    cut = min(max(cut, 8), len(nlen))
    while cut < len(nlen) and node_site[cut] == node_site[cut - 1]:
        cut += 1
    member, nlen = member[:cut], nlen[:cut]
    V = cut
    paths = [np.nonzero(member[:, h])[0].astype(np.uint32) for h in range(H)]
    glen = np.array([nlen[p].sum() for p in paths], dtype=np.int64)
    path_off = np.zeros(H + 1, dtype=np.uint64)
    path_off[1:] = np.cumsum([len(p) for p in paths])
    path_nodes = np.concatenate(paths).astype(np.uint32)
    n_present = max(1, int(round(present_frac * H)))
    present = rng.choice(H, size=n_present, replace=False)
    depth = np.zeros(H)
    depth[present] = rng.lognormal(depth_mu, depth_sigma, size=n_present)
    return SpeciesGraph(name, nlen, path_off, path_nodes, hap_names, range_start, range_start + V - 1, glen, depth)


def _walks_for_strain(g, h, pos, rlen):
    """Vectorised node walks of reads [pos, pos+rlen) on strain h (forward orientation).
    Returns (i0, i1, off0): first/last step index in the strain's path and offset in first node."""
    b, e = int(g.path_off[h]), int(g.path_off[h + 1])
    lens = g.node_len[g.path_nodes[b:e]]
    cum_end = np.cumsum(lens)
    cum_start = cum_end - lens
    i0 = np.searchsorted(cum_end, pos, side="right")
    i1 = np.searchsorted(cum_end, pos + rlen - 1, side="right")
    return i0, i1, pos - cum_start[i0], cum_end


def make_reads(rng, species, n_reads, read_len=150, long_reads=False, adversarial_frac=0.001, with_ids=False):
    """Error-free reads: path span == read length, pstart = offset in first node."""
    S = len(species)
    # read share: depth * genome_len weights
    w = []
    key = []
    for si, g in enumerate(species):
        for h in range(g.n_paths):
            if g.truth_depth[h] > 0:
                w.append(g.truth_depth[h] * g.genome_len[h])
                key.append((si, h))
    w = np.array(w, dtype=np.float64)
    counts = rng.multinomial(n_reads, w / w.sum())
    seg_off, seg_node, seg_strand = [], [], []
    pstart_l, pend_l, qlen_l = [], [], []
    total_steps = 0
    for (si, h), c in zip(key, counts):
        if c == 0:
            continue
        g = species[si]
        glen = int(g.genome_len[h])
        if long_reads:
            rl = np.clip(rng.normal(15000, 3000, size=c), 2000, 25000).astype(np.int64)
        else:
            rl = np.full(c, read_len, dtype=np.int64)
        rl = np.minimum(rl, glen)
        pos = (rng.random(c) * (glen - rl + 1)).astype(np.int64)
        i0, i1, off0, cum_end = _walks_for_strain(g, h, pos, rl)
        nsteps = (i1 - i0 + 1).astype(np.int64)
        rev = rng.random(c) < 0.5
        b = int(g.path_off[h])
        # flat step indices
        starts = np.cumsum(nsteps) - nsteps
        flat = np.arange(int(nsteps.sum()), dtype=np.int64) - np.repeat(starts, nsteps)
        fwd_idx = np.repeat(i0, nsteps) + flat
        rev_idx = np.repeat(i1, nsteps) - flat
        idx = np.where(np.repeat(rev, nsteps), rev_idx, fwd_idx)
        nodes = g.path_nodes[b + idx].astype(np.int64) + g.range_start  # global 1-based id
        # offset in first node of the walk orientation
        end_off = cum_end[i1] - (pos + rl)  # unused tail of last node (forward)
        ps = np.where(rev, end_off, off0)
        seg_node.append(nodes.astype(np.uint32))
        seg_strand.append(np.repeat(rev, nsteps).astype(np.uint8))
        seg_off.append(nsteps)
        pstart_l.append(ps)
        pend_l.append(ps + rl)
        qlen_l.append(rl)
        total_steps += int(nsteps.sum())
    nsteps = np.concatenate(seg_off)
    R = len(nsteps)
    # shuffle reads so species are interleaved like a real GAF
    perm = rng.permutation(R)
    starts = np.cumsum(nsteps) - nsteps
    node_all = np.concatenate(seg_node)
    strand_all = np.concatenate(seg_strand)
    ns_p = nsteps[perm]
    new_starts = np.cumsum(ns_p) - ns_p
    gather = np.repeat(starts[perm], ns_p) + (np.arange(int(ns_p.sum())) - np.repeat(new_starts, ns_p))
    node_id = node_all[gather]
    strand = strand_all[gather]
    step_off = np.zeros(R + 1, dtype=np.uint64)
    step_off[1:] = np.cumsum(ns_p)
    pstart = np.concatenate(pstart_l)[perm].astype(np.int64)
    pend = np.concatenate(pend_l)[perm].astype(np.int64)
    qlen = np.concatenate(qlen_l)[perm].astype(np.int64)
    mapq = np.where(rng.random(R) < 0.85, 60, rng.integers(0, 60, size=R)).astype(np.int64)
    # adversarial records (0.1 %): single-node end<start, repeated node, cross-species walk
    n_adv = int(R * adversarial_frac)
    if n_adv > 0:
        adv = rng.choice(R, size=n_adv, replace=False)
        for r in adv:
            b, e = int(step_off[r]), int(step_off[r + 1])
            k = e - b
            mode = int(rng.integers(0, 3))
            if mode == 0 and k == 1:
                pstart[r] += 50                                      # end < start on a single node (vg issue 4249),
                pend[r] = pstart[r] - int(rng.integers(1, 50))       # both still valid non-negative GAF values
            elif mode == 1 and k >= 3:
                node_id[b + 2] = node_id[b]                          # a,b,a repeat
            elif mode == 2 and k >= 2 and S > 1:
                other = species[int(rng.integers(0, S))]
                node_id[e - 1] = np.uint32(other.range_start)       # crosses species -> "U" (or same species)
    ids = []
    if with_ids:
        ids = ["S0R%d/1" % i for i in range(R)]
    return PackedReads(step_off, node_id, strand, pstart, pend, qlen, mapq, qlen.copy(), ids)


CONFIGS = {
    # name: (n_species, haps per species, reads, genome_len, long_reads)
    "tiny": (2, 4, 2000, 20000, False),
    "small": (3, 6, 20000, 60000, False),
    "cfg2": (1, 10, 1_000_000, 5_000_000, False),
    "cfg3": (100, 10, 10_000_000, 5_000_000, False),
    "cfg4_shard": (125, 10, 12_500_000, 5_000_000, False),  # 1/8 of cfg4 (1k species / 100M reads)
}


def make_set(seed, n_species, H, n_reads, genome_len, long_reads=False, with_ids=False, adversarial_frac=0.001,
             single_strain_every=0, present_frac=0.2):
    rng = np.random.default_rng(seed)
    species = []
    start = 1
    for s in range(n_species):
        h = 1 if (single_strain_every and s % single_strain_every == single_strain_every - 1) else H
        g = make_species(rng, str(1000 + s), h, genome_len, start, "GCF_%06d" % (s + 1), present_frac=present_frac)
        species.append(g)
        start = g.range_end + 1
    reads = make_reads(rng, species, n_reads, long_reads=long_reads, with_ids=with_ids,
                       adversarial_frac=adversarial_frac)
    return SyntheticSet(species, reads)


def _species_job(job):
    seed_seq, name, h, genome_len, prefix, present_frac = job
    return make_species(np.random.default_rng(seed_seq), name, h, genome_len, 1, prefix, present_frac=present_frac)


def make_set_mp(seed, n_species, H, n_reads, genome_len, long_reads=False, adversarial_frac=0.001, present_frac=0.2, workers=None):
    """make_set for the big configurations: species generated by a fork pool from per-species seed sequences (so the
    result does not depend on the worker count), reads from one more child sequence.  Different sets than make_set for
    the same seed -- the bench keeps make_set."""
    import multiprocessing as mp
    import os
    workers = workers or min(os.cpu_count() or 1, 16)
    ss = np.random.SeedSequence(seed).spawn(n_species + 1)
    jobs = [(ss[s], str(1000 + s), H, genome_len, "GCF_%06d" % (s + 1), present_frac) for s in range(n_species)]
    if workers > 1 and n_species > 1:
        with mp.get_context("fork").Pool(min(workers, n_species)) as pool:
            species = pool.map(_species_job, jobs, chunksize=1)
    else:
        species = [_species_job(j) for j in jobs]
    start = 1
    for g in species:
        n = g.n_nodes
        g.range_start, g.range_end = start, start + n - 1
        start += n
    reads = make_reads(np.random.default_rng(ss[n_species]), species, n_reads, long_reads=long_reads, adversarial_frac=adversarial_frac)
    return SyntheticSet(species, reads)


# ---------------------------------------------------------------- native generator (tools/native/synth_set.c)
N_CHUNKS = 256   # the reads of a native set come in this many independently generated chunks (ranks / samples take chunk ranges)


def _native_set_lib():
    """tools/native/libsynthset.so (C, built by __graft_entry__.build(); built on demand when missing)."""
    import ctypes as C
    import os
    import subprocess
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "native")
    so = os.path.join(d, "libsynthset.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", d])
    lib = C.CDLL(so)
    lib.synth_gaf_size.restype = C.c_uint64
    lib.synth_gaf_write_at.restype = C.c_int64
    return lib


class NativeSet:
    """The synthetic set of SURVEY 8d as a pure function of (seed, species) and (seed, chunk, read) -- see synth_set.c.
    Constructing it computes only the dimensions of every species (milliseconds each); graphs and reads are generated on
    request, on `threads` host threads (the C calls release the GIL):
        ns = NativeSet(seed, 1000, 10, 100_000_000, 5_000_000)
        sset = ns.make()                      # everything: SyntheticSet(all graphs, all reads)
        rd = ns.reads(32, 64)                 # chunks [32, 64) of N_CHUNKS: what rank 1 of 8 reads
        gs = ns.graphs([3, 17])               # the graphs a rank owns
    The set does not depend on the number of threads, chunks requested or ranks."""

    def __init__(self, seed, n_species, H, n_reads, genome_len, long_reads=False, adversarial_frac=0.001, present_frac=0.2, read_len=150,
                 threads=None):
        import ctypes as C
        import os
        from concurrent.futures import ThreadPoolExecutor
        if not 1 <= H <= 63:
            raise ValueError("NativeSet: 1..63 strains per species")
        self.lib = _native_set_lib()
        self.seed, self.S, self.H, self.n_reads, self.genome_len = int(seed), int(n_species), int(H), int(n_reads), int(genome_len)
        self.long_reads, self.adversarial_frac, self.read_len = bool(long_reads), float(adversarial_frac), int(read_len)
        self.threads = threads or min(os.cpu_count() or 1, 64)

        class Params(C.Structure):
            _fields_ = [("seed", C.c_uint64), ("H", C.c_uint32), ("genome_len", C.c_int64), ("frac_snp", C.c_double), ("frac_acc", C.c_double),
                        ("mean_len", C.c_double), ("present_frac", C.c_double), ("depth_mu", C.c_double), ("depth_sigma", C.c_double)]
        self._par = Params(self.seed, self.H, self.genome_len, 0.30, 0.12, 32.0, float(present_frac), float(np.log(8.0)), 1.0)
        S = self.S
        self.V = np.zeros(S, dtype=np.uint64)
        self.P = np.zeros(S, dtype=np.uint64)
        self.path_len = np.zeros((S, H), dtype=np.uint64)
        self.glen = np.zeros((S, H), dtype=np.int64)
        self.depth = np.zeros((S, H), dtype=np.float64)
        pv = lambda a, i: C.c_void_p(a.ctypes.data + i * a.strides[0])

        def dims(s):
            rc = self.lib.synth_species_dims(C.byref(self._par), C.c_uint32(s), pv(self.V, s), pv(self.P, s), pv(self.path_len, s), pv(self.glen, s),
                                             pv(self.depth, s))
            if rc != 0:
                raise RuntimeError("synth_species_dims(%d) = %d" % (s, rc))
        with ThreadPoolExecutor(self.threads) as ex:
            list(ex.map(dims, range(S)))
        self.range_start = np.ones(S, dtype=np.int64)
        self.range_start[1:] = 1 + np.cumsum(self.V[:-1].astype(np.int64))
        self.range_end = self.range_start + self.V.astype(np.int64) - 1
        self.names = [str(1000 + s) for s in range(S)]
        self._graphs = {}

    def hap_names(self, s):
        return sorted("GCF_%06d%03d.1" % (s + 1, h) for h in range(self.H))

    def avg_len(self):
        return self.glen.mean(axis=1).astype(np.float64)

    def chunk_reads(self, c):
        return self.n_reads // N_CHUNKS + (1 if c < self.n_reads % N_CHUNKS else 0)

    def _graph(self, s):
        import ctypes as C
        g = self._graphs.get(s)
        if g is None:
            V, P = int(self.V[s]), int(self.P[s])
            node_len = np.empty(V, dtype=np.int64)
            path_off = np.empty(self.H + 1, dtype=np.uint64)
            path_nodes = np.empty(P, dtype=np.uint32)
            rc = self.lib.synth_species_fill(C.byref(self._par), C.c_uint32(s), C.c_uint64(V), C.c_void_p(node_len.ctypes.data),
                                             C.c_void_p(path_off.ctypes.data), C.c_void_p(path_nodes.ctypes.data))
            if rc != 0:
                raise RuntimeError("synth_species_fill(%d) = %d" % (s, rc))
            g = SpeciesGraph(self.names[s], node_len, path_off, path_nodes, self.hap_names(s), int(self.range_start[s]), int(self.range_end[s]),
                             self.glen[s].copy(), self.depth[s].copy())
            self._graphs[s] = g
        return g

    def graphs(self, idx=None, keep=True):
        """SpeciesGraph objects of the species `idx` (all by default), generated on the host threads; keep=False does not
        cache them inside the set."""
        from concurrent.futures import ThreadPoolExecutor
        idx = list(range(self.S)) if idx is None else [int(i) for i in idx]
        with ThreadPoolExecutor(self.threads) as ex:
            gs = list(ex.map(self._graph, idx))
        if not keep:
            for i in idx:
                self._graphs.pop(i, None)
        return gs

    def drop_graphs(self, keep_idx=()):
        keep_idx = set(int(i) for i in keep_idx)
        for i in list(self._graphs):
            if i not in keep_idx:
                del self._graphs[i]

    def reads(self, chunk_lo=0, chunk_hi=N_CHUNKS):
        """PackedReads of chunks [chunk_lo, chunk_hi).  Needs the walks of every present strain of every species: the graphs are
        generated (and cached; drop_graphs() frees them) first."""
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor
        lib = self.lib
        gs = self.graphs()
        # present strains, read share = depth x genome length
        ent = [(s, h) for s in range(self.S) for h in range(self.H) if self.depth[s, h] > 0]
        w = np.array([self.depth[s, h] * self.glen[s, h] for s, h in ent], dtype=np.float64)
        cum_w = np.cumsum(w) / w.sum()
        cum_w[-1] = 1.0
        n_st = len(ent)
        walk_nodes = np.zeros(n_st, dtype=np.uint64)
        walk_cum = np.zeros(n_st, dtype=np.uint64)
        walk_len = np.zeros(n_st, dtype=np.uint64)
        st_glen = np.zeros(n_st, dtype=np.int64)
        st_start = np.zeros(n_st, dtype=np.int64)
        cums = [None] * n_st

        def prep(i):
            s, h = ent[i]
            g = gs[s]
            b, e = int(g.path_off[h]), int(g.path_off[h + 1])
            cum = np.empty(e - b, dtype=np.uint32)
            lib.synth_walk_cum(C.c_void_p(g.node_len.ctypes.data), C.c_void_p(g.path_nodes.ctypes.data + 4 * b), C.c_uint64(e - b), C.c_void_p(cum.ctypes.data))
            cums[i] = cum
            walk_nodes[i] = g.path_nodes.ctypes.data + 4 * b
            walk_cum[i] = cum.ctypes.data
            walk_len[i] = e - b
            st_glen[i] = self.glen[s, h]
            st_start[i] = self.range_start[s]
        with ThreadPoolExecutor(self.threads) as ex:
            list(ex.map(prep, range(n_st)))

        class Ctx(C.Structure):
            _fields_ = [("seed", C.c_uint64), ("n_strains", C.c_uint32), ("cum_w", C.c_void_p), ("walk_nodes", C.c_void_p), ("walk_cum", C.c_void_p),
                        ("walk_len", C.c_void_p), ("genome_len", C.c_void_p), ("range_start", C.c_void_p), ("n_species", C.c_uint32),
                        ("species_start", C.c_void_p), ("long_reads", C.c_int32), ("read_len", C.c_int64), ("adversarial_frac", C.c_double)]
        ctx = Ctx(self.seed, n_st, cum_w.ctypes.data, walk_nodes.ctypes.data, walk_cum.ctypes.data, walk_len.ctypes.data, st_glen.ctypes.data,
                  st_start.ctypes.data, self.S, self.range_start.ctypes.data, int(self.long_reads), self.read_len, self.adversarial_frac)
        chunks = list(range(int(chunk_lo), int(chunk_hi)))
        cnt = np.array([self.chunk_reads(c) for c in chunks], dtype=np.int64)
        first = np.concatenate([[0], np.cumsum(cnt)])
        R = int(first[-1])
        n_steps = np.empty(R, dtype=np.uint32)
        at = lambda a, i: C.c_void_p(a.ctypes.data + int(i) * a.itemsize)

        def count(k):
            lib.synth_reads_count(C.byref(ctx), C.c_uint64(chunks[k]), C.c_uint64(0), C.c_uint64(int(cnt[k])), at(n_steps, first[k]))
        with ThreadPoolExecutor(self.threads) as ex:
            list(ex.map(count, range(len(chunks))))
        step_off = np.zeros(R + 1, dtype=np.uint64)
        np.cumsum(n_steps, out=step_off[1:])
        del n_steps
        T = int(step_off[-1])
        node_id = np.empty(T, dtype=np.uint32)
        strand = np.empty(T, dtype=np.uint8)
        pstart, pend, qlen, mapq = (np.empty(R, dtype=np.int64) for _ in range(4))

        def fill(k):
            a = int(first[k])
            lib.synth_reads_fill(C.byref(ctx), C.c_uint64(chunks[k]), C.c_uint64(0), C.c_uint64(int(cnt[k])), at(step_off, a), C.c_void_p(node_id.ctypes.data),
                                 C.c_void_p(strand.ctypes.data), at(pstart, a), at(pend, a), at(qlen, a), at(mapq, a))
        with ThreadPoolExecutor(self.threads) as ex:
            list(ex.map(fill, range(len(chunks))))
        return PackedReads(step_off, node_id, strand, pstart, pend, qlen, mapq, qlen, [])

    def make(self, chunk_lo=0, chunk_hi=N_CHUNKS):
        rd = self.reads(chunk_lo, chunk_hi)
        return SyntheticSet(self.graphs(), rd)


def make_set_native(seed, n_species, H, n_reads, genome_len, **kw):
    """SyntheticSet from the native generator (all graphs, all reads)."""
    return NativeSet(seed, n_species, H, n_reads, genome_len, **kw).make()


def write_gaf_parallel(reads, path, tags="NM:i:0\tAS:i:150\tdv:f:0\tid:f:1", threads=None):
    """write_gaf on host threads (native set writer): the byte size of every slice of reads is computed first, then the slices
    are written at their offsets of one pre-sized file.  Same bytes as write_gaf.  -> bytes written."""
    import ctypes as C
    import os
    from concurrent.futures import ThreadPoolExecutor
    lib = _native_set_lib()
    threads = threads or min(os.cpu_count() or 1, 64)
    R = reads.n_reads
    a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
    so, nid, st = a(reads.step_off, np.uint64), a(reads.node_id, np.uint32), a(reads.strand, np.uint8)
    ps, pe, ql, mq = (a(x, np.int64) for x in (reads.pstart, reads.pend, reads.qlen, reads.mapq))
    if not np.array_equal(a(reads.plen, np.int64), ql):
        raise ValueError("write_gaf_parallel: plen == qlen only")
    pp = lambda x: C.c_void_p(x.ctypes.data)
    n_sl = max(1, min(4 * threads, R // 4096 + 1))
    cuts = np.linspace(0, R, n_sl + 1).astype(np.int64)
    tg = tags.encode()

    def size(i):
        return lib.synth_gaf_size(C.c_uint64(int(cuts[i])), C.c_uint64(int(cuts[i + 1])), C.c_uint64(0), pp(so), pp(nid), pp(ps), pp(pe), pp(ql), pp(mq),
                                  C.c_uint64(len(tg)))
    with ThreadPoolExecutor(threads) as ex:
        sizes = list(ex.map(size, range(n_sl)))
    off = np.concatenate([[0], np.cumsum(np.array(sizes, dtype=np.int64))])
    with open(path, "wb") as f:
        f.truncate(int(off[-1]))

    def write(i):
        n = lib.synth_gaf_write_at(str(path).encode(), C.c_uint64(int(off[i])), C.c_uint64(int(cuts[i])), C.c_uint64(int(cuts[i + 1])), C.c_uint64(0),
                                   pp(so), pp(nid), pp(st), pp(ps), pp(pe), pp(ql), pp(mq), tg)
        if n != sizes[i]:
            raise IOError("synth_gaf_write_at(%s) slice %d: %d of %d bytes" % (path, i, n, sizes[i]))
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(write, range(n_sl)))
    return int(off[-1])


def cached_set(seed, n_species, H, n_reads, genome_len, cache_dir=None, **kw):
    """make_set through a pickle cache (profiling drivers re-run the same workload once per counter pass); cache_dir
    None = $PANTAX_SYNTH_CACHE, unset = no cache."""
    import os
    import pickle
    cache_dir = cache_dir or os.environ.get("PANTAX_SYNTH_CACHE")
    if not cache_dir:
        return make_set(seed, n_species, H, n_reads, genome_len, **kw)
    os.makedirs(cache_dir, exist_ok=True)
    tag = "_".join("%s%s" % (k, v) for k, v in sorted(kw.items()))
    fn = os.path.join(cache_dir, "set_%d_%d_%d_%d_%d_%s.pkl" % (seed, n_species, H, n_reads, genome_len, tag))
    if os.path.exists(fn):
        with open(fn, "rb") as f:
            return pickle.load(f)
    sset = make_set(seed, n_species, H, n_reads, genome_len, **kw)
    tmp = fn + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        pickle.dump(sset, f, protocol=4)
    os.replace(tmp, fn)
    return sset


def shard_set(full, rank, world, partition):
    """The share of `full` that rank `rank` of `world` owns: species packed by weight (8 x reads + graph nodes, the rule of the
    C file seam; `partition` = pipeline.partition_species), reads follow the species of their FIRST node (a walk that
    leaves its species is "U" on any rank).  Graph node ids stay global, so every rank bins against its own ranges."""
    starts = np.array([g.range_start for g in full.species], dtype=np.int64)
    rd = full.reads
    so = rd.step_off.astype(np.int64)
    k = so[1:] - so[:-1]
    first = np.where(k > 0, rd.node_id[np.minimum(so[:-1], max(len(rd.node_id) - 1, 0))].astype(np.int64), 0)
    sp_of = np.searchsorted(starts, first, side="right") - 1
    sp_of = np.where(k > 0, sp_of, 0)
    cnt = np.bincount(sp_of, minlength=len(full.species))
    owner = np.array(partition([8.0 * cnt[i] + g.n_nodes for i, g in enumerate(full.species)], world))
    mine = np.nonzero(owner == rank)[0]
    sel = np.nonzero(owner[sp_of] == rank)[0]
    ns = k[sel]
    new_off = np.zeros(len(sel) + 1, dtype=np.uint64)
    new_off[1:] = np.cumsum(ns)
    idx = np.repeat(so[:-1][sel], ns) + (np.arange(int(ns.sum())) - np.repeat(new_off[:-1].astype(np.int64), ns))
    reads = PackedReads(new_off, rd.node_id[idx], rd.strand[idx], rd.pstart[sel], rd.pend[sel], rd.qlen[sel], rd.mapq[sel], rd.plen[sel],
                        [rd.read_id[i] for i in sel] if rd.read_id else [])
    return SyntheticSet([full.species[i] for i in mine], reads)


def make_config(name, cfg_index=0, **kw):
    S, H, R, L, lr = CONFIGS[name]
    return make_set(20260501 + cfg_index, S, H, R, L, long_reads=lr, **kw)


# ---------------------------------------------------------------- file writers
def write_db(sset, db_dir, write_gfa=True, write_bin=True, threads=1):
    """Write species_range.txt, species_genomes_stats.txt, genomes_info.txt,
    species_gfa/<sp>.gfa (W lines) and species_graph_info/<sp>.bin (bincode-1 layout,
    zip.rs:171-190: u64 len + i64s; u64 map len; per entry u64 key len + bytes + u64 vec len + u64s).
    threads > 1: the per-species files are written on that many host threads (numpy casts and file writes release the GIL):
    the 1 000 .bin files of cfg4 are 20 GB."""
    import os
    import struct
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(db_dir, "species_gfa"), exist_ok=True)
    os.makedirs(os.path.join(db_dir, "species_graph_info"), exist_ok=True)
    with open(os.path.join(db_dir, "species_range.txt"), "w") as f:
        for name, s, e, pan in sset.range_table():
            f.write("%s\t%d\t%d\t%d\n" % (name, s, e, pan))
    with open(os.path.join(db_dir, "species_genomes_stats.txt"), "w") as f:
        for g in sset.species:
            f.write("%s\t%s\n" % (g.name, repr(float(g.genome_len.mean()))))
    with open(os.path.join(db_dir, "genomes_info.txt"), "w") as f:
        f.write("genome_ID\tstrain_taxid\tspecies_taxid\torganism_name\tid\n")
        for g in sset.species:
            for h, hn in enumerate(g.hap_names):
                gid = "%s_ASM%sv1" % (hn, hn[4:10])
                f.write("%s\t%s.%d\t%s\tSynthetic species %s\t/path/to/%s_genomic.fna\n" % (gid, g.name, h + 1, g.name, g.name, gid))

    def one(g):
        if write_gfa:
            with open(os.path.join(db_dir, "species_gfa", g.name + ".gfa"), "w") as f:
                f.write("H\tVN:Z:1.1\n")
                for v in range(g.n_nodes):
                    f.write("S\t%d\t%s\n" % (v + 1, "A" * int(g.node_len[v])))
                for h, hn in enumerate(g.hap_names):
                    b, e = int(g.path_off[h]), int(g.path_off[h + 1])
                    walk = "".join(">%d" % (int(v) + 1) for v in g.path_nodes[b:e])
                    f.write("W\t%s\t1\tctg1\t0\t%d\t%s\n" % (hn, int(g.genome_len[h]), walk))
        if write_bin:
            with open(os.path.join(db_dir, "species_graph_info", g.name + ".bin"), "wb") as f:
                f.write(struct.pack("<Q", g.n_nodes))
                g.node_len.astype("<i8").tofile(f)
                f.write(struct.pack("<Q", g.n_paths))
                for h, hn in enumerate(g.hap_names):
                    b, e = int(g.path_off[h]), int(g.path_off[h + 1])
                    hb = hn.encode()
                    f.write(struct.pack("<Q", len(hb)))
                    f.write(hb)
                    f.write(struct.pack("<Q", e - b))
                    g.path_nodes[b:e].astype("<u8").tofile(f)
    if threads > 1:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one, sset.species))
    else:
        for g in sset.species:
            one(g)


def _native_gaf_writer():
    """tools/native/libsynthgaf.so (C, built by __graft_entry__.build()): the same bytes as the loop below, ~1 GB/s."""
    import ctypes as C
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "native", "libsynthgaf.so")
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    lib.synth_write_gaf.restype = C.c_int64
    return lib


def head_reads(reads, k):
    """The first k reads as a PackedReads of their own (views, no copies)."""
    k = min(int(k), reads.n_reads)
    t = int(reads.step_off[k])
    return PackedReads(reads.step_off[:k + 1], reads.node_id[:t], reads.strand[:t], reads.pstart[:k], reads.pend[:k], reads.qlen[:k],
                       reads.mapq[:k], reads.plen[:k], reads.read_id[:k] if reads.read_id else [])


def write_gaf(reads, path, tags="NM:i:0\tAS:i:150\tdv:f:0\tid:f:1", native=True):
    """12 mandatory GAF columns + 4 fixed tags (constant column count)."""
    lib = _native_gaf_writer() if (native and not reads.read_id) else None
    if lib is not None:
        import ctypes as C
        a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
        so, nid, st = a(reads.step_off, np.uint64), a(reads.node_id, np.uint32), a(reads.strand, np.uint8)
        cols = [a(x, np.int64) for x in (reads.pstart, reads.pend, reads.qlen, reads.mapq, reads.plen)]
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        n = lib.synth_write_gaf(str(path).encode(), 0, C.c_uint64(0), C.c_uint64(reads.n_reads), C.c_uint64(0), p(so), p(nid), p(st),
                                *[p(c) for c in cols], tags.encode())
        if n < 0:
            raise IOError("synth_write_gaf(%s) failed: %d" % (path, n))
        return
    with open(path, "w") as f:
        R = reads.n_reads
        for r in range(R):
            b, e = int(reads.step_off[r]), int(reads.step_off[r + 1])
            walk = "".join(("<" if reads.strand[i] else ">") + str(int(reads.node_id[i])) for i in range(b, e))
            if not walk:
                walk = "*"
            rid = reads.read_id[r] if reads.read_id else "S0R%d/1" % r
            ql = int(reads.qlen[r])
            f.write("%s\t%d\t0\t%d\t+\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%s\n" % (
                rid, ql, ql, walk, int(reads.plen[r]), int(reads.pstart[r]), int(reads.pend[r]), ql, ql,
                int(reads.mapq[r]), tags))


# ---------------------------------------------------------------------------------------------------------------------------------------
# A workload shaped like the database PanTax ships: the reference's genomes_info.txt lists 13 404 genomes of
# 8 778 species -- 7 465 of them with ONE genome, whose "pangenome" is the genome cut into 1024-bp chunks (build_eq1.rs:26-36,
# constants.rs:3), the others with 2 .. 10.  Only the HISTOGRAM of genomes per species is taken from that file (numbers, no content).
# The set is composed of blocks of species with the same number of strains, one after the other in node-id order: the singletons (chunk
# graphs, reads drawn here) and one NativeSet per strain count 2 .. 10 (its node ids shifted behind the blocks before it).
# ---------------------------------------------------------------------------------------------------------------------------------------
REFDB_STRAINS_PER_SPECIES = {1: 7465, 2: 514, 3: 219, 4: 118, 5: 73, 6: 57, 7: 51, 8: 30, 9: 32, 10: 219}


class RefDbSet:
    """ns = RefDbSet(seed, n_reads); ns.make() -> SyntheticSet.  `scale` shrinks every species count (tests): scale=0.01 -> 88 species.
    Interface of NativeSet where bench.py / the tests use it: S, names, range_start, range_end, V, P, graphs(), reads(), avg_len(), make()."""

    def __init__(self, seed, n_reads, genome_len=5_000_000, scale=1.0, present_frac=0.2, threads=None, read_len=150, adversarial_frac=0.001):
        import os
        self.seed, self.n_reads, self.genome_len, self.read_len = int(seed), int(n_reads), int(genome_len), int(read_len)
        self.threads = threads or min(os.cpu_count() or 1, 64)
        rng = np.random.default_rng(self.seed)
        counts = {h: max(1, int(round(c * scale))) for h, c in REFDB_STRAINS_PER_SPECIES.items()}
        self.counts = counts
        # ---- block of singletons: genome length jittered +-10 %, a fifth of the species present with a log-normal depth
        n1 = counts[1]
        self.single_glen = (genome_len * (0.9 + 0.2 * rng.random(n1))).astype(np.int64)
        self.single_depth = np.where(rng.random(n1) < present_frac, rng.lognormal(np.log(8.0), 1.0, n1), 0.0)
        if not self.single_depth.any():
            self.single_depth[0] = 8.0
        self.single_V = np.maximum(3, -(-self.single_glen // 1024)).astype(np.int64)          # as _species_graph of make_set: at least three chunks
        # ---- one NativeSet per strain count (dimensions only until graphs() / reads())
        self.blocks = []                                                                       # (h, NativeSet)
        for h in range(2, 11):
            self.blocks.append((h, NativeSet(self.seed + 100 * h, counts[h], h, 1, genome_len, present_frac=present_frac, read_len=read_len,
                                             adversarial_frac=adversarial_frac, threads=self.threads)))
        # read shares: depth x genome length of the present strains
        w = [float((self.single_depth * self.single_glen).sum())] + [float((b.depth * b.glen).sum()) for _, b in self.blocks]
        share = np.array(w) / sum(w)
        nr = np.floor(share * self.n_reads).astype(np.int64)
        nr[0] += self.n_reads - int(nr.sum())
        self.block_reads = nr
        for (h, b), n in zip(self.blocks, nr[1:]):
            b.n_reads = int(n)
        # ---- global numbering: the singletons first, then the blocks
        Vs = [self.single_V] + [b.V.astype(np.int64) for _, b in self.blocks]
        self.V = np.concatenate(Vs).astype(np.uint64)
        self.P = np.concatenate([self.single_V] + [b.P.astype(np.int64) for _, b in self.blocks]).astype(np.uint64)
        self.S = len(self.V)
        self.range_start = np.ones(self.S, dtype=np.int64)
        self.range_start[1:] = 1 + np.cumsum(self.V[:-1].astype(np.int64))
        self.range_end = self.range_start + self.V.astype(np.int64) - 1
        self.block_first_species = np.concatenate([[0], np.cumsum([len(v) for v in Vs])]).astype(np.int64)
        self.n_haps = np.concatenate([np.ones(n1, dtype=np.int64)] + [np.full(b.S, h, dtype=np.int64) for h, b in self.blocks])
        self.names = [str(100000 + s) for s in range(self.S)]
        self._graphs = {}

    def hap_names(self, s):
        return sorted("GCF_%07d%03d.1" % (s + 1, k) for k in range(int(self.n_haps[s])))

    def avg_len(self):
        out = [self.single_glen.astype(np.float64)] + [b.avg_len() for _, b in self.blocks]
        return np.concatenate(out)

    def _block_of(self, s):
        k = int(np.searchsorted(self.block_first_species, s, side="right") - 1)
        return k, s - int(self.block_first_species[k])

    def _graph(self, s):
        g = self._graphs.get(s)
        if g is None:
            k, j = self._block_of(s)
            if k == 0:
                n = int(self.single_V[j])
                node_len = np.full(n, 1024, dtype=np.int64)
                node_len[-1] = max(1, int(self.single_glen[j]) - 1024 * (n - 1))
                g = SpeciesGraph(self.names[s], node_len, np.array([0, n], dtype=np.uint64), np.arange(n, dtype=np.uint32), self.hap_names(s),
                                 int(self.range_start[s]), int(self.range_end[s]), np.array([int(node_len.sum())], dtype=np.int64),
                                 np.array([float(self.single_depth[j])]))
            else:
                h, b = self.blocks[k - 1]
                sub = b._graph(j)
                g = SpeciesGraph(self.names[s], sub.node_len, sub.path_off, sub.path_nodes, self.hap_names(s), int(self.range_start[s]), int(self.range_end[s]),
                                 sub.genome_len, sub.truth_depth)
            self._graphs[s] = g
        return g

    def graphs(self, idx=None, keep=True):
        from concurrent.futures import ThreadPoolExecutor
        idx = list(range(self.S)) if idx is None else [int(i) for i in idx]
        with ThreadPoolExecutor(self.threads) as ex:
            gs = list(ex.map(self._graph, idx))
        if not keep:
            for i in idx:
                self._graphs.pop(i, None)
        return gs

    def drop_graphs(self, keep_idx=()):
        keep_idx = set(int(i) for i in keep_idx)
        for i in list(self._graphs):
            if i not in keep_idx:
                del self._graphs[i]
        for _, b in self.blocks:
            b.drop_graphs(())

    def _single_reads(self):
        """Reads on the chunk graphs: uniform start on a present genome, read_len bases, either strand, MAPQ as SURVEY 8d."""
        n = int(self.block_reads[0])
        rng = np.random.default_rng(self.seed + 7)
        w = self.single_depth * self.single_glen
        sp = rng.choice(len(w), size=n, p=w / w.sum())
        glen = np.array([max(1, int(self.single_glen[j]) - 1024 * (int(self.single_V[j]) - 1)) + 1024 * (int(self.single_V[j]) - 1) for j in range(len(w))], dtype=np.int64)
        L = self.read_len
        start = (rng.random(n) * np.maximum(glen[sp] - L, 1)).astype(np.int64)
        n0, n1 = start // 1024, np.minimum((start + L - 1) // 1024, self.single_V[sp] - 1)
        k = (n1 - n0 + 1).astype(np.int64)                          # one or two chunks (a read is shorter than a chunk)
        rev = rng.random(n) < 0.5
        base = self.range_start[sp]                                 # global id of the species' chunk 0
        step_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(k, out=step_off[1:])
        T = int(step_off[-1])
        node_id = np.empty(T, dtype=np.uint32)
        strand = np.repeat(rev.astype(np.uint8), k)
        first = step_off[:-1].astype(np.int64)
        a = np.where(rev, n1, n0)                                   # first step of the walk
        node_id[first] = (base + a).astype(np.uint32)
        two = k == 2
        node_id[first[two] + 1] = (base[two] + np.where(rev[two], n0[two], n1[two])).astype(np.uint32)
        last_len = (glen[sp] - 1024 * (self.single_V[sp] - 1))
        len1 = np.where(n1 == self.single_V[sp] - 1, last_len, 1024)   # length of the read's last chunk (forward order)
        plen = np.where(two, 1024 + len1, np.where(n0 == self.single_V[sp] - 1, last_len, 1024))
        ps_f = start - 1024 * n0
        ps = np.where(rev, plen - (ps_f + L), ps_f)
        pe = ps + L
        mapq = np.where(rng.random(n) < 0.85, 60, rng.integers(0, 60, n)).astype(np.int64)
        qlen = np.full(n, L, dtype=np.int64)
        return PackedReads(step_off, node_id, strand, ps.astype(np.int64), pe.astype(np.int64), qlen, mapq, qlen, [])   # (GAF column 7 = read length, as NativeSet writes it)

    def reads(self, chunk_lo=0, chunk_hi=N_CHUNKS):
        if (chunk_lo, chunk_hi) != (0, N_CHUNKS):
            raise ValueError("RefDbSet: the whole set only (one GPU)")
        parts = [self._single_reads()]
        for k, (h, b) in enumerate(self.blocks):
            if b.n_reads <= 0:
                continue
            r = b.reads()
            shift = np.uint32(int(self.range_start[int(self.block_first_species[k + 1])]) - 1)      # the block's id 1 -> its first global id
            r.node_id += shift
            parts.append(r)
            b.drop_graphs(())
        offs = np.concatenate([[0]] + [p.step_off[1:].astype(np.int64) + sum(int(q.step_off[-1]) for q in parts[:i]) for i, p in enumerate(parts)]).astype(np.uint64)
        cat = lambda f: np.concatenate([getattr(p, f) for p in parts])
        return PackedReads(offs, cat("node_id"), cat("strand"), cat("pstart"), cat("pend"), cat("qlen"), cat("mapq"), cat("plen"), [])

    def make(self):
        rd = self.reads()
        return SyntheticSet(self.graphs(), rd)
