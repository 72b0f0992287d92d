"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under pantax_amd/ does. See oracle/pantax_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpantax_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


class OrcGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("node_len", C.c_void_p), ("n_paths", C.c_uint32),
                ("path_off", C.c_void_p), ("path_nodes", C.c_void_p)]


class OrcTrio(C.Structure):
    _fields_ = [("n_unique", C.c_uint64), ("abc", C.POINTER(C.c_uint32)), ("hap", C.POINTER(C.c_uint32)),
                ("len", C.POINTER(C.c_int64)), ("hap_off", C.POINTER(C.c_uint64)),
                ("sorted_abc", C.POINTER(C.c_uint32)), ("sorted_row", C.POINTER(C.c_uint64))]


class OrcHapMetrics(C.Structure):
    _fields_ = [("has", C.c_uint32), ("unique_trio_nodes_fraction", C.c_double), ("frequencies_mean", C.c_double),
                ("path_cov_ratio", C.c_double), ("first_sol", C.c_double), ("divergence", C.c_double),
                ("second_sol", C.c_double), ("total_cov_diff", C.c_double), ("is_rescue", C.c_int32)]


class OrcStrainConfig(C.Structure):
    _fields_ = [("unique_trio_nodes_fraction", C.c_double), ("unique_trio_nodes_mean_count_f", C.c_double),
                ("single_cov_ratio", C.c_double), ("min_depth", C.c_int64), ("shift", C.c_int32), ("sample_nodes", C.c_int32),
                ("solver_semantics", C.c_int32)]


HAS = dict(fraction=1, freq_mean=2, ratio=4, first=8, divergence=16, second=32, rescue=64, total_diff=128)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Graph:
    """One species graph in oracle layout (types.rs:51-55)."""

    def __init__(self, node_len, path_off, path_nodes):
        self.node_len = np.ascontiguousarray(node_len, dtype=np.int64)
        self.path_off = np.ascontiguousarray(path_off, dtype=np.uint64)
        self.path_nodes = np.ascontiguousarray(path_nodes, dtype=np.uint32)
        self.c = OrcGraph(len(self.node_len), _p(self.node_len).value, len(self.path_off) - 1,
                          _p(self.path_off).value, _p(self.path_nodes).value)

    @property
    def n_nodes(self):
        return len(self.node_len)

    @property
    def n_paths(self):
        return len(self.path_off) - 1


class TrioTable:
    def __init__(self, graph):
        self.graph = graph
        self.c = OrcTrio()
        lib().orc_trio_index(C.byref(graph.c), C.byref(self.c))
        U = self.c.n_unique
        self.n_unique = U
        self.abc = np.ctypeslib.as_array(self.c.abc, shape=(max(U, 1) * 3,))[: 3 * U].reshape(-1, 3).copy()
        self.hap = np.ctypeslib.as_array(self.c.hap, shape=(max(U, 1),))[:U].copy()
        self.len = np.ctypeslib.as_array(self.c.len, shape=(max(U, 1),))[:U].copy()
        self.hap_off = np.ctypeslib.as_array(self.c.hap_off, shape=(graph.n_paths + 1,)).copy()

    def __del__(self):
        try:
            lib().orc_trio_free(C.byref(self.c))
        except Exception:
            pass


def bin_reads(step_off, node_id, range_start, range_end):
    step_off = np.ascontiguousarray(step_off, dtype=np.uint64)
    node_id = np.ascontiguousarray(node_id, dtype=np.uint32)
    rs = np.ascontiguousarray(range_start, dtype=np.int64)
    re = np.ascontiguousarray(range_end, dtype=np.int64)
    n = len(step_off) - 1
    out = np.empty(n, dtype=np.int32)
    lib().orc_bin_reads(C.c_uint64(n), _p(step_off), _p(node_id), C.c_uint32(len(rs)), _p(rs), _p(re), _p(out))
    return out


def species_counts(species_idx, read_len, mapq, n_ranges):
    species_idx = np.ascontiguousarray(species_idx, dtype=np.int32)
    read_len = np.ascontiguousarray(read_len, dtype=np.int64)
    mapq = np.ascontiguousarray(mapq, dtype=np.int64)
    outs = [np.zeros(n_ranges, dtype=np.int64) for _ in range(4)]
    lib().orc_species_counts(C.c_uint64(len(species_idx)), _p(species_idx), _p(read_len), _p(mapq),
                             C.c_uint32(n_ranges), *[_p(o) for o in outs])
    return outs


def species_profile(species_idx, read_len, counts, avg_len, filtered=True):
    species_idx = np.ascontiguousarray(species_idx, dtype=np.int32)
    read_len = np.ascontiguousarray(read_len, dtype=np.int64)
    avg_len = np.ascontiguousarray(avg_len, dtype=np.float64)
    S = len(avg_len)
    keep = np.zeros(S, dtype=np.uint8)
    absolute = np.zeros(S)
    abundance = np.zeros(S)
    lib().orc_species_profile(C.c_uint64(len(species_idx)), _p(species_idx), _p(read_len), C.c_uint32(S),
                              *[_p(np.ascontiguousarray(c, dtype=np.int64)) for c in counts], _p(avg_len),
                              C.c_int(int(filtered)), _p(keep), _p(absolute), _p(abundance))
    return keep, absolute, abundance


def node_coverage(graph, trio, range_start, step_off, node_id, pstart, pend):
    step_off = np.ascontiguousarray(step_off, dtype=np.uint64)
    node_id = np.ascontiguousarray(node_id, dtype=np.uint32)
    pstart = np.ascontiguousarray(pstart, dtype=np.int64)
    pend = np.ascontiguousarray(pend, dtype=np.int64)
    V = graph.n_nodes
    U = trio.n_unique if trio is not None else 0
    bases = np.zeros(V, dtype=np.int64)
    cov = np.zeros(V, dtype=np.uint64)
    tb = np.zeros(max(U, 1), dtype=np.int64)
    n_abort = C.c_uint64(0)
    lib().orc_node_coverage(C.byref(graph.c), C.byref(trio.c) if trio is not None else None,
                            C.c_int64(range_start), C.c_uint64(len(step_off) - 1), _p(step_off), _p(node_id),
                            _p(pstart), _p(pend), _p(bases), _p(cov), _p(tb), C.byref(n_abort))
    return bases, cov, tb[:U], n_abort.value


def hap_trio_stats(trio, n_paths, trio_bases):
    tb = np.ascontiguousarray(trio_bases, dtype=np.int64)
    nt = np.zeros(n_paths, dtype=np.uint64)
    nz = np.zeros(n_paths, dtype=np.uint64)
    mf = np.zeros(n_paths)
    lib().orc_hap_trio_stats(C.byref(trio.c), C.c_uint32(n_paths), _p(tb), _p(nt), _p(nz), _p(mf))
    return nt, nz, mf


def path_masks(graph, cand, node_base_cov):
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    cov = np.ascontiguousarray(node_base_cov, dtype=np.uint64)
    nw = max(1, (len(cand) + 63) // 64)     # one word per node up to 64 candidates, else an (n_nodes, nw) array
    mask = np.zeros(graph.n_nodes if nw == 1 else (graph.n_nodes, nw), dtype=np.uint64)
    ratio = np.zeros(len(cand), dtype=np.float32)
    rc = lib().orc_path_masks(C.byref(graph.c), C.c_uint32(len(cand)), _p(cand), _p(cov), _p(mask), _p(ratio))
    assert rc == 0
    return mask, ratio


def _mask_words(mask, n_cand):
    """(n,) uint64 for up to 64 columns, (n, ceil(n_cand / 64)) beyond."""
    mask = np.ascontiguousarray(mask, dtype=np.uint64)
    nw = max(1, (n_cand + 63) // 64)
    assert mask.shape == ((len(mask),) if nw == 1 and mask.ndim == 1 else (len(mask), nw)), (mask.shape, n_cand)
    return mask


def lad_solve(mask, abund, n_cand, ub):
    mask = _mask_words(mask, n_cand)
    abund = np.ascontiguousarray(abund, dtype=np.float64)
    ub = np.ascontiguousarray(ub, dtype=np.float64)
    x = np.zeros(n_cand)
    obj = C.c_double(0)
    it = C.c_int32(0)
    st = C.c_int32(0)
    lib().orc_lad_solve(C.c_uint64(len(mask)), _p(mask), _p(abund), C.c_uint32(n_cand), _p(ub), _p(x),
                        C.byref(obj), C.byref(it), C.byref(st))
    return x, obj.value, it.value, st.value


def lad_objective(mask, abund, x):
    mask = _mask_words(mask, len(x))
    abund = np.ascontiguousarray(abund, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().orc_lad_objective.restype = C.c_double
    return lib().orc_lad_objective(C.c_uint64(len(mask)), _p(mask), _p(abund), C.c_uint32(len(x)), _p(x))


def optimize_species(graph, trio, bases, cov, trio_bases, fr=0.3, fc=0.46, sr=0.85, min_depth=0, shift=False, sample_nodes=0, solver_semantics=0):
    cfg = OrcStrainConfig(fr, fc, sr, min_depth, int(shift), int(sample_nodes), int(solver_semantics))
    H = graph.n_paths
    met = (OrcHapMetrics * H)()
    nc = C.c_uint32(0)
    o1 = C.c_double(0)
    o2 = C.c_double(0)
    bases = np.ascontiguousarray(bases, dtype=np.int64)
    cov = np.ascontiguousarray(cov, dtype=np.uint64)
    tb = np.ascontiguousarray(trio_bases if len(trio_bases) else np.zeros(1), dtype=np.int64)
    rc = lib().orc_optimize_species(C.byref(graph.c), C.byref(trio.c), _p(bases), _p(cov), _p(tb), C.byref(cfg), met,
                                    C.byref(nc), C.byref(o1), C.byref(o2))
    return rc, met, nc.value, o1.value, o2.value


def par_bin_reads(step_off, node_id, range_start, range_end, threads):
    """bin_reads in `threads` read slices on pthreads (oracle_parallel.c; bench.py's cpu_baseline)"""
    step_off = np.ascontiguousarray(step_off, dtype=np.uint64)
    node_id = np.ascontiguousarray(node_id, dtype=np.uint32)
    rs = np.ascontiguousarray(range_start, dtype=np.int64)
    re = np.ascontiguousarray(range_end, dtype=np.int64)
    n = len(step_off) - 1
    out = np.empty(n, dtype=np.int32)
    rc = lib().orc_par_bin_reads(C.c_int(int(threads)), C.c_uint64(n), _p(step_off), _p(node_id), C.c_uint32(len(rs)), _p(rs), _p(re), _p(out))
    assert rc == 0
    return out


def group_reads(species_idx, n_species):
    """group_reads_by_species (profile.rs:439-463) as a stable counting sort -> first [S+1], order"""
    sp = np.ascontiguousarray(species_idx, dtype=np.int32)
    first = np.zeros(n_species + 1, dtype=np.uint64)
    order = np.empty(len(sp), dtype=np.uint64)
    rc = lib().orc_group_reads(C.c_uint64(len(sp)), _p(sp), C.c_uint32(n_species), _p(first), _p(order))
    assert rc == 0
    return first, order[: int(first[-1])]


def par_profile_species(graphs, range_start, step_off, node_id, pstart, pend, first, order, keep, absolute, todo, threads,
                        fr=0.3, fc=0.46, sr=0.85, min_depth=0, shift=False, sample_nodes=0, want_metrics=False, solver_semantics=0):
    """One species per worker thread, handed out in the order of `todo` (the rayon par_iter of profile.rs:3297-3319):
    trio index, the species' reads, node coverage, filters + both solves, abundace_constraint -- all through the
    single-threaded functions above.  graphs: list of Graph.  -> dict of per-species arrays (+ metrics when asked)."""
    S = len(graphs)
    garr = (OrcGraph * S)(*[g.c for g in graphs])
    rs = np.ascontiguousarray(range_start, dtype=np.int64)
    step_off = np.ascontiguousarray(step_off, dtype=np.uint64)
    node_id = np.ascontiguousarray(node_id, dtype=np.uint32)
    pstart = np.ascontiguousarray(pstart, dtype=np.int64)
    pend = np.ascontiguousarray(pend, dtype=np.int64)
    first = np.ascontiguousarray(first, dtype=np.uint64)
    order = np.ascontiguousarray(order, dtype=np.uint64)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    absolute = np.ascontiguousarray(absolute, dtype=np.float64)
    todo = np.ascontiguousarray(todo, dtype=np.uint32)
    hap_off = np.zeros(S + 1, dtype=np.uint64)
    hap_off[1:] = np.cumsum([g.n_paths for g in graphs])
    met = (OrcHapMetrics * max(int(hap_off[-1]), 1))() if want_metrics else None
    cfg = OrcStrainConfig(fr, fc, sr, min_depth, int(shift), int(sample_nodes), int(solver_semantics))
    out = dict(rc=np.zeros(S, dtype=np.int32), n_cand=np.zeros(S, dtype=np.uint32), n_rows=np.zeros(S, dtype=np.uint64), obj1=np.zeros(S), obj2=np.zeros(S),
               t_trio=np.zeros(S), t_cov=np.zeros(S), t_lp=np.zeros(S))
    rc = lib().orc_par_profile_species(C.c_int(int(threads)), C.c_uint32(S), garr, _p(rs), _p(step_off), _p(node_id), _p(pstart), _p(pend), _p(first), _p(order),
                                       _p(keep), _p(absolute), C.byref(cfg), C.c_uint32(len(todo)), _p(todo), _p(hap_off), met,
                                       *[_p(out[k]) for k in ("rc", "n_cand", "n_rows", "obj1", "obj2", "t_trio", "t_cov", "t_lp")])
    assert rc == 0
    out["metrics"], out["hap_off"] = met, hap_off
    return out


def gaf_filter(text):
    """SURVEY 8f-3 (gaf_filter.rs:44-97): bool per raw line of `text` (bytes) = the line is written; also #records."""
    keep = np.zeros(text.count(b"\n") + 2, dtype=np.uint8)
    nrec = C.c_uint64(0)
    lib().orc_gaf_filter.restype = C.c_int64
    n = lib().orc_gaf_filter(C.c_char_p(text), C.c_uint64(len(text)), _p(keep), C.byref(nrec))
    assert n >= 0
    return keep[:n].astype(bool), nrec.value


def sample_sorted_positions(length, amount, seed=42):
    """a11 (profile.rs:1287-1295): ascending positions rand 0.9.2's choose_multiple keeps (restated, parity unpinned)."""
    out = np.zeros(max(amount, 1), dtype=np.uint32)
    rc = lib().orc_sample_sorted_positions(C.c_uint32(length), C.c_uint32(amount), C.c_uint64(seed), _p(out))
    assert rc == 0
    return out[:amount]


def chacha_block(key8, counter, rounds):
    key = np.ascontiguousarray(key8, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    lib().orc_chacha_block(_p(key), C.c_uint64(counter), C.c_int(rounds), _p(out))
    return out


def abundance_constraint(species_cov, met):
    lib().orc_abundance_constraint(C.c_double(species_cov), C.c_uint32(len(met)), met)
    return met


def metrics_to_dicts(met):
    out = []
    for m in met:
        d = {}
        for name, bit, attr in [("unique_trio_fraction", 1, "unique_trio_nodes_fraction"),
                                ("uniq_trio_cov_mean", 2, "frequencies_mean"), ("path_base_cov", 4, "path_cov_ratio"),
                                ("first_sol", 8, "first_sol"), ("strain_cov_diff", 16, "divergence"),
                                ("predicted_coverage", 32, "second_sol"), ("total_cov_diff", 128, "total_cov_diff")]:
            d[name] = getattr(m, attr) if m.has & bit else None
        d["is_rescue"] = bool(m.is_rescue) if m.has & 64 else None
        out.append(d)
    return out
