"""Host-side mirror of the reference's profiling interface over the C ABI.

Method names follow the reference functions they stand in for (profile.rs / rcls.rs)
so the parity tests read like tests of the reference:
    rcls_profile            rcls.rs:452-458   (+ species counters profile.rs:208-297)
    trio_nodes_info         profile.rs:658-740
    get_node_abundances     profile.rs:743-1026
    strain_profiling        profile.rs:3291-3323 (optimize_otu + abundace_constraint per species)
    pao_solve               the X_opt solver seam, profile.rs:2690-2698
Everything here is plumbing: numpy arrays in, ctypes call, numpy arrays out.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import PantaxHipError, as_c, p


class Engine:
    def __init__(self, device=0):
        self.lib = _ffi.load()
        self.ctx = C.c_void_p()
        dev = (C.c_int * 1)(device)
        rc = self.lib.pantax_hip_init(C.byref(self.ctx), dev, 1)
        if rc != 0:
            raise PantaxHipError(rc, self.lib.pantax_hip_last_error(None).decode())
        self.db = None
        self.reads = None
        self._keep = []
        self._step_buf = None
        self._collect_buf = None
        self._inflight = 0

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise PantaxHipError(rc, self.lib.pantax_hip_last_error(self.ctx).decode())

    def close(self):
        if self.ctx:
            if self.reads:
                self.lib.pantax_hip_reads_free(self.ctx, self.reads)
                self.reads = None
            if self.db:
                self.lib.pantax_hip_db_free(self.ctx, self.db)
                self.db = None
            self.lib.pantax_hip_destroy(self.ctx)
            self.ctx = None

    def release(self):
        """free the resident db and reads (HBM), keep the ctx"""
        if self.ctx:
            if self.db and self._inflight:
                self.drain_steps()                                # enqueued steps still read the db and the reads
            if self.reads:
                self.lib.pantax_hip_reads_free(self.ctx, self.reads)
                self.reads = None
            if self.db:
                self.lib.pantax_hip_db_free(self.ctx, self.db)
                self.db = None
            self._inflight = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._check(self.lib.pantax_hip_sync(self.ctx))

    def set_option(self, name, value=None):
        """pantax_hip_set_option: a switch of this ctx (common.hpp CtxConfig); value None = its default.  The library reads the
        environment only once, at init."""
        self._check(self.lib.pantax_hip_set_option(self.ctx, name.encode(), None if value is None else str(value).encode()))

    # ------------------------------------------------------------------ uploads
    def upload_db(self, species):
        """species: list of objects with node_len, path_off, path_nodes, range_start, range_end
        (synthdata.SpeciesGraph or pantax_amd.io graphs), haplotypes in byte order."""
        if self.db:
            self.lib.pantax_hip_db_free(self.ctx, self.db)
            self.db = None
            self._inflight = 0
        S = len(species)
        self.S = S
        self.range_start = as_c([g.range_start for g in species], np.int64)
        self.range_end = as_c([g.range_end for g in species], np.int64)
        self.node_off = np.zeros(S + 1, dtype=np.uint64)
        self.node_off[1:] = np.cumsum([len(g.node_len) for g in species])
        self.hap_off = np.zeros(S + 1, dtype=np.uint64)
        self.hap_off[1:] = np.cumsum([len(g.path_off) - 1 for g in species])
        self.V = int(self.node_off[-1])
        self.H = int(self.hap_off[-1])
        # one part per species (pantax_hip_db_upload_parts): the arrays travel as they are, nothing is concatenated on the host
        keep = []
        parts = (_ffi.GraphPart * S)()
        for i, g in enumerate(species):
            nl, po, pn = as_c(g.node_len, np.int64), as_c(g.path_off, np.uint64), as_c(g.path_nodes, np.uint32)
            keep.append((nl, po, pn))
            parts[i] = _ffi.GraphPart(len(nl), len(po) - 1, nl.ctypes.data, po.ctypes.data, pn.ctypes.data if len(pn) else None)
        db = C.c_void_p()
        self._check(self.lib.pantax_hip_db_upload_parts(self.ctx, C.c_uint32(S), p(self.range_start), p(self.range_end), parts, C.byref(db)))
        self.db = db
        self.U = None

    def upload_db_flat(self, species):
        """The same db through pantax_hip_db_upload (SURVEY 8b's struct of flat arrays: all species concatenated by the caller)."""
        if self.db:
            self.lib.pantax_hip_db_free(self.ctx, self.db)
            self.db = None
            self._inflight = 0
        S = len(species)
        self.S = S
        self.range_start = as_c([g.range_start for g in species], np.int64)
        self.range_end = as_c([g.range_end for g in species], np.int64)
        self.node_off = np.zeros(S + 1, dtype=np.uint64)
        self.node_off[1:] = np.cumsum([len(g.node_len) for g in species])
        self.hap_off = np.zeros(S + 1, dtype=np.uint64)
        self.hap_off[1:] = np.cumsum([len(g.path_off) - 1 for g in species])
        node_len = as_c(np.concatenate([g.node_len for g in species]), np.int64)
        offs = [np.zeros(1, dtype=np.uint64)]
        base = 0
        for g in species:
            offs.append(np.asarray(g.path_off[1:], dtype=np.uint64) + np.uint64(base))
            base += int(g.path_off[-1])
        path_off = as_c(np.concatenate(offs), np.uint64)
        path_nodes = as_c(np.concatenate([g.path_nodes for g in species]), np.uint32)
        self.V = int(self.node_off[-1])
        self.H = int(self.hap_off[-1])
        gs = _ffi.Graphs(S, p(self.range_start), p(self.range_end), p(self.node_off), p(node_len),
                         p(self.hap_off), p(path_off), p(path_nodes))
        db = C.c_void_p()
        self._check(self.lib.pantax_hip_db_upload(self.ctx, C.byref(gs), C.byref(db)))
        self.db = db
        self.U = None

    def upload_ranges(self, range_start, range_end):
        """A db of species RANGES only (no graphs): what the binning of a slice of reads needs before the reads are routed
        to the owners of their species; the file seam bins against such a db too."""
        if self.db:
            self.lib.pantax_hip_db_free(self.ctx, self.db)
            self.db = None
            self._inflight = 0
        self.range_start = as_c(range_start, np.int64)
        self.range_end = as_c(range_end, np.int64)
        self.S = len(self.range_start)
        self.V = self.H = 0
        gs = _ffi.Graphs(self.S, p(self.range_start), p(self.range_end), None, None, None, None, None)
        db = C.c_void_p()
        self._check(self.lib.pantax_hip_db_upload(self.ctx, C.byref(gs), C.byref(db)))
        self.db = db
        self.U = None

    def upload_reads(self, step_off, node_id, pstart, pend, qlen, mapq, flags=None):
        if self.reads:
            self.lib.pantax_hip_reads_free(self.ctx, self.reads)
            self.reads = None
        for name, a in (("step_off", step_off), ("node_id", node_id), ("pstart", pstart), ("pend", pend), ("qlen", qlen)):
            a = np.asarray(a)
            if a.size and (a.min() < 0 or a.max() > 0xFFFFFFFF):
                raise ValueError("%s outside the packed u32 range [0, 2^32): GAF columns are non-negative" % name)
        arrs = dict(step_off=as_c(step_off, np.uint32), node_id=as_c(node_id, np.uint32), pstart=as_c(pstart, np.uint32),
                    pend=as_c(pend, np.uint32), qlen=as_c(qlen, np.uint32), mapq=as_c(mapq, np.uint8),
                    flags=None if flags is None else as_c(flags, np.uint8))
        R = len(arrs["pstart"])
        self.R = R
        self.T = len(arrs["node_id"])
        pr = _ffi.PackedReads(R, self.T, p(arrs["step_off"]), p(arrs["node_id"]), p(arrs["pstart"]), p(arrs["pend"]),
                              p(arrs["qlen"]), p(arrs["mapq"]), p(arrs["flags"]))
        rd = C.c_void_p()
        self._check(self.lib.pantax_hip_reads_upload(self.ctx, C.byref(pr), C.byref(rd)))
        self.reads = rd

    def upload_packed(self, reads, flags=None):
        """reads: synthdata.PackedReads (int64 host arrays)"""
        mapq = np.where((reads.mapq < 0) | (reads.mapq > 254), 255, reads.mapq)
        self.upload_reads(reads.step_off, reads.node_id, reads.pstart, reads.pend, reads.qlen, mapq, flags)

    def load_reads_from_gaf(self, path, columns=True):
        """GAF file -> packed reads resident in HBM, tokenised on the device (pantax_hip_reads_load_gaf).
        -> dict(qlen, mapq, flags) of the host-side columns (columns=False: None, nothing is copied out); the walks stay on
        the device."""
        if self.reads:
            self.lib.pantax_hip_reads_free(self.ctx, self.reads)
            self.reads = None
        rd, gaf = C.c_void_p(), C.c_void_p()
        if not columns:   # no gaf handle: the library does not bring the per-read host columns back at all
            self._check(self.lib.pantax_hip_reads_load_gaf(self.ctx, str(path).encode(), C.byref(rd), None))
            self.reads = rd
            self.R = self.T = None       # not known on the host (the counters of a binning pass tell)
            return None
        self._check(self.lib.pantax_hip_reads_load_gaf(self.ctx, str(path).encode(), C.byref(rd), C.byref(gaf)))
        self.reads = rd
        try:
            v = _ffi.PackedReads()
            self.lib.pantax_hip_gaf_view(gaf, C.byref(v))
            self.R = int(v.n_reads)
            arr = lambda ptr, dt: (np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(self.R,)).copy()
                                   if self.R else np.zeros(0, dtype=dt))
            return dict(qlen=arr(v.qlen, np.uint32), mapq=arr(v.mapq, np.uint8), flags=arr(v.flags, np.uint8))
        finally:
            self.lib.pantax_hip_gaf_free(gaf)

    # ------------------------------------------------------------------ reads routed to the owner of their species
    def route_pack(self, owner_of_species, world):
        """The resident reads (binned against the resident db) as one message per owner rank (stable partition on the
        device).  -> (route handle, n_reads_to [W], n_steps_to [W]); free the handle with route_free."""
        own = as_c(owner_of_species, np.int32)
        assert len(own) == self.S
        nr = np.zeros(world, dtype=np.uint64)
        nt = np.zeros(world, dtype=np.uint64)
        rt = C.c_void_p()
        self._check(self.lib.pantax_hip_reads_route_pack(self.ctx, self.db, self.reads, p(own), int(world), C.byref(rt), p(nr), p(nt)))
        return rt, nr, nt

    def route_buffer(self, route, world, on_device=False):
        """-> (address of the W messages back to back, word offsets [W+1]); device pointer when on_device, else pinned host memory."""
        buf = C.c_void_p()
        off = np.zeros(world + 1, dtype=np.uint64)
        self._check(self.lib.pantax_hip_route_buffer(self.ctx, route, int(bool(on_device)), C.byref(buf), p(off)))
        return buf.value or 0, off

    def route_messages(self, route, world):
        """host copies of the W messages (uint32 arrays)"""
        addr, off = self.route_buffer(route, world, on_device=False)
        n = int(off[-1])
        if n == 0:
            return [np.zeros(0, dtype=np.uint32) for _ in range(world)]
        whole = np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint32)), shape=(n,))
        return [whole[int(off[d]):int(off[d + 1])].copy() for d in range(world)]

    def route_free(self, route):
        self.lib.pantax_hip_route_free(self.ctx, route)

    def reads_from_routed(self, recv, n_reads_from, n_steps_from, on_device=False):
        """recv: the messages of ranks 0..W-1 for this rank back to back -- a uint32 numpy array (host) or, with on_device, the
        integer address of device memory.  Replaces the resident reads."""
        if self.reads:
            self.lib.pantax_hip_reads_free(self.ctx, self.reads)
            self.reads = None
        nr = as_c(n_reads_from, np.uint64)
        nt = as_c(n_steps_from, np.uint64)
        if on_device:
            ptr = C.c_void_p(int(recv))
        else:
            recv = as_c(recv, np.uint32)
            ptr = p(recv)
        rd = C.c_void_p()
        self._check(self.lib.pantax_hip_reads_from_routed(self.ctx, ptr, int(bool(on_device)), len(nr), p(nr), p(nt), C.byref(rd)))
        self.reads = rd
        self.R = int(nr.sum())
        self.T = int(nt.sum())

    def set_read_flags(self, flags):
        f = None if flags is None else as_c(flags, np.uint8)
        self._check(self.lib.pantax_hip_reads_set_flags(self.ctx, self.reads, p(f)))

    # ------------------------------------------------------------------ stages
    def rcls_profile(self, want_species=True):
        """-> (species_idx [R] int32 or None, read_count, base_sum, less_multi, uniq_count [S] int64)"""
        sp = np.empty(self.R, dtype=np.int32) if want_species else None
        outs = [np.zeros(self.S, dtype=np.int64) for _ in range(4)]
        self._check(self.lib.pantax_hip_bin_reads(self.ctx, self.db, self.reads, p(sp), *[p(o) for o in outs]))
        return (sp, *outs)

    def species_profiling(self, counts, avg_len, filtered=True):
        """profile.rs:299-349 finishing -> keep [S] uint8, predicted_coverage [S], predicted_abundance [S]"""
        rc, bs, lm, uq = [as_c(c, np.int64) for c in counts]
        avg = as_c(avg_len, np.float64)
        keep = np.zeros(self.S, dtype=np.uint8)
        absolute = np.zeros(self.S)
        abundance = np.zeros(self.S)
        self._check(self.lib.pantax_hip_species_profile(self.ctx, self.db, self.reads, p(rc), p(bs), p(lm), p(uq), p(avg),
                                                        int(filtered), p(keep), p(absolute), p(abundance)))
        return keep, absolute, abundance

    def db_reset(self):
        self._check(self.lib.pantax_hip_db_reset(self.ctx, self.db))
        self.U = None

    def abundance_filter(self, met, species_reported=None, single_cov_diff=0.2, min_cov=0):
        """abundance_est filters (profile.rs:3219-3245) -> pass [H] uint8, per-species sum_all [S], sum_pass [S]"""
        rep = None if species_reported is None else as_c(species_reported, np.uint8)
        passed = np.zeros(max(self.H, 1), dtype=np.uint8)
        sa = C.c_double(0)
        sp = C.c_double(0)
        s_all = np.zeros(self.S)
        s_pass = np.zeros(self.S)
        rc = self.lib.pantax_hip_abundance_filter(C.c_uint32(self.S), p(self.hap_off), met, p(rep), C.c_double(single_cov_diff),
                                                  C.c_int64(min_cov), p(passed), C.byref(sa), C.byref(sp), p(s_all), p(s_pass))
        self._check(rc)
        return passed[: self.H], s_all, s_pass

    def trio_nodes_info(self, fetch=True):
        n = C.c_uint64(0)
        self._check(self.lib.pantax_hip_trio_index(self.ctx, self.db, C.byref(n)))
        self.U = n.value
        if not fetch:
            return self.U
        U = self.U
        abc = np.zeros(max(U, 1) * 3, dtype=np.uint32)
        hap = np.zeros(max(U, 1), dtype=np.uint32)
        ln = np.zeros(max(U, 1), dtype=np.int64)
        hto = np.zeros(self.H + 1, dtype=np.uint64)
        self._check(self.lib.pantax_hip_trio_get(self.ctx, self.db, p(abc), p(hap), p(ln), p(hto)))
        return abc[: 3 * U].reshape(-1, 3), hap[:U], ln[:U], hto

    def get_node_abundances(self, species_active=None, fetch=True, with_trio=True):
        """-> bases_per_node [V] int64, node_base_cov [V] uint64, trio_bases [U] int64, n_abort"""
        act = None if species_active is None else as_c(species_active, np.uint8)
        n_abort = C.c_uint64(0)
        if not fetch:   # results stay resident for strain_profiling; no host round trip
            self._check(self.lib.pantax_hip_node_coverage(self.ctx, self.db, self.reads, p(act), None, None, None, None))
            return None
        bases = np.zeros(self.V, dtype=np.int64)
        cov = np.zeros(self.V, dtype=np.uint64)
        tb = None
        if with_trio and self.U is not None:
            tb = np.zeros(max(self.U, 1), dtype=np.int64)
        self._check(self.lib.pantax_hip_node_coverage(self.ctx, self.db, self.reads, p(act), p(bases), p(cov), p(tb),
                                                      C.byref(n_abort)))
        return bases, cov, (tb[: self.U] if tb is not None else None), n_abort.value

    def strain_profiling(self, species_coverage, species_active=None, fr=0.3, fc=0.46, sr=0.85, min_depth=0,
                         shift=False, sample_nodes=0, solver_semantics=0):
        """solver_semantics: 0 = Gurobi's handling of the second solve (profile.rs:1500-1508), 1 = highs_opt's (profile.rs:2865-2879)"""
        cfg = _ffi.StrainConfig(fr, fc, sr, min_depth, int(shift), sample_nodes, int(solver_semantics))
        act = None if species_active is None else as_c(species_active, np.uint8)
        cov = as_c(species_coverage, np.float64)
        met = (_ffi.HapMetrics * max(self.H, 1))()
        info = (_ffi.SolveInfo * self.S)()
        self._check(self.lib.pantax_hip_strain_profile(self.ctx, self.db, C.byref(cfg), p(act), p(cov), met, info))
        return met, info

    def profile_step(self, avg_len, fr=0.3, fc=0.46, sr=0.85, sd=0.2, min_cov=0, min_depth=0, shift=False, filtered=True,
                     rebuild_trio=True, sample_nodes=0, solver_semantics=0):
        """One resident pass of the hot path in a single call (pantax_hip_profile_step): the host waits once.
        -> keep [S] uint8, predicted_coverage [S], metrics [H], info [S], pass [H] uint8, sum_all [S], sum_pass [S]"""
        if self._step_buf is None or self._step_buf[0] != (self.S, self.H):
            self._step_buf = ((self.S, self.H), np.zeros(self.S, dtype=np.uint8), np.zeros(self.S),
                              (_ffi.HapMetrics * max(self.H, 1))(), (_ffi.SolveInfo * self.S)(),
                              np.zeros(max(self.H, 1), dtype=np.uint8), np.zeros(self.S), np.zeros(self.S))
            self._step_ptr = [p(a) if isinstance(a, np.ndarray) else a for a in self._step_buf[1:]]
        _, keep, absolute, met, info, passed, s_all, s_pass = self._step_buf
        avg = as_c(avg_len, np.float64)
        cfg = _ffi.StepConfig(fr, fc, sr, sd, min_cov, min_depth, int(shift), int(filtered), int(sample_nodes), int(rebuild_trio), int(solver_semantics))
        self._check(self.lib.pantax_hip_profile_step(self.ctx, self.db, self.reads, p(avg), C.byref(cfg), *self._step_ptr))
        if rebuild_trio:
            self.U = None
        return keep, absolute, met, info, passed[: self.H], s_all, s_pass

    def trio_index_prefetch(self):
        """pantax_hip_trio_index_prefetch: the unique-trio index of the COMING step is started now (side stream), e.g. before that
        run's reads are loaded; the next step with rebuild_trio uses it instead of building again."""
        self._check(self.lib.pantax_hip_trio_index_prefetch(self.ctx, self.db))
        self.U = None

    def profile_step_enqueue(self, avg_len, fr=0.3, fc=0.46, sr=0.85, sd=0.2, min_cov=0, min_depth=0, shift=False, filtered=True,
                             rebuild_trio=True, sample_nodes=0, solver_semantics=0):
        """First half of profile_step (pantax_hip_profile_step_enqueue): the whole step goes onto the device, nothing is waited
        for.  Up to two steps may be in flight; the device runs their main-stream work one step after the other (the unique-trio
        rebuild of the next step may start behind the previous step's first filter, its last reader)."""
        avg = as_c(avg_len, np.float64)
        cfg = _ffi.StepConfig(fr, fc, sr, sd, min_cov, min_depth, int(shift), int(filtered), int(sample_nodes), int(rebuild_trio), int(solver_semantics))
        self._check(self.lib.pantax_hip_profile_step_enqueue(self.ctx, self.db, self.reads, p(avg), C.byref(cfg)))
        self._inflight += 1
        if rebuild_trio:
            self.U = None

    def profile_step_collect(self):
        """Second half: the oldest enqueued step's host wait + results (same tuple as profile_step).  The arrays are reused by
        the collect after next: copy what must live longer."""
        if self._collect_buf is None or self._collect_buf[0] != (self.S, self.H):
            mk = lambda: (np.zeros(self.S, dtype=np.uint8), np.zeros(self.S), (_ffi.HapMetrics * max(self.H, 1))(), (_ffi.SolveInfo * self.S)(),
                          np.zeros(max(self.H, 1), dtype=np.uint8), np.zeros(self.S), np.zeros(self.S))
            sets = [mk(), mk()]
            self._collect_buf = ((self.S, self.H), sets, [[p(a) if isinstance(a, np.ndarray) else a for a in st] for st in sets], 0)
        key, sets, ptrs, turn = self._collect_buf
        self._collect_buf = (key, sets, ptrs, turn ^ 1)
        keep, absolute, met, info, passed, s_all, s_pass = sets[turn]
        self._inflight = max(self._inflight - 1, 0)           # the library frees the slot also when the step reports a failure
        self._check(self.lib.pantax_hip_profile_step_collect(self.ctx, self.db, *ptrs[turn]))
        return keep, absolute, met, info, passed[: self.H], s_all, s_pass

    def drain_steps(self):
        """collect (and drop) whatever enqueued steps are still in flight, e.g. after an exception between enqueue and collect"""
        while self._inflight:
            try:
                self.profile_step_collect()
            except PantaxHipError:
                pass

    def pao_solve(self, node_len, node_abundance, node_base_cov, path_off, path_nodes, cand, fixed_zero=None):
        node_len = as_c(node_len, np.int64)
        ab = as_c(node_abundance, np.float64)
        cov = as_c(node_base_cov, np.uint64)
        path_off = as_c(path_off, np.uint64)
        path_nodes = as_c(path_nodes, np.uint32)
        cand = as_c(cand, np.uint32)
        fz = None if fixed_zero is None else as_c(fixed_zero, np.uint8)
        x = np.zeros(len(cand))
        ratio = np.zeros(len(cand), dtype=np.float32)
        obj = C.c_double(0)
        st = C.c_int32(0)
        self._check(self.lib.pantax_hip_pao_solve(self.ctx, C.c_uint32(len(node_len)), p(node_len), p(ab), p(cov),
                                                  C.c_uint32(len(path_off) - 1), p(path_off), p(path_nodes),
                                                  C.c_uint32(len(cand)), p(cand), p(fz), p(x), p(ratio),
                                                  C.byref(obj), C.byref(st)))
        return x, ratio, obj.value, st.value

    def pao_solve_batch(self, species, fixed_zero=None):
        """The solver seam for many species in ONE call (pantax_hip_pao_solve_batch).  species: list of
        (node_len, node_abundance, node_base_cov or None, path_off, path_nodes, cand) per species; fixed_zero: list of
        per-species uint8 arrays or None.  -> list of (x, ratio, obj, status, iters) per species."""
        S = len(species)
        node_off = np.zeros(S + 1, dtype=np.uint64); hap_off = np.zeros(S + 1, dtype=np.uint64); cand_off = np.zeros(S + 1, dtype=np.uint64)
        node_off[1:] = np.cumsum([len(sp[0]) for sp in species])
        hap_off[1:] = np.cumsum([len(sp[3]) - 1 for sp in species])
        cand_off[1:] = np.cumsum([len(sp[5]) for sp in species])
        node_len = as_c(np.concatenate([sp[0] for sp in species]), np.int64)
        ab = as_c(np.concatenate([sp[1] for sp in species]), np.float64)
        cov = as_c(np.concatenate([np.zeros(len(sp[0]), dtype=np.uint64) if sp[2] is None else sp[2] for sp in species]), np.uint64)
        offs, base = [np.zeros(1, dtype=np.uint64)], 0
        for sp in species:
            po = np.asarray(sp[3], dtype=np.uint64)
            offs.append(po[1:] + np.uint64(base)); base += int(po[-1])
        path_off = as_c(np.concatenate(offs), np.uint64)
        path_nodes = as_c(np.concatenate([sp[4] for sp in species]), np.uint32)
        cand = as_c(np.concatenate([np.asarray(sp[5], dtype=np.uint32) for sp in species]), np.uint32)
        fz = None
        if fixed_zero is not None:
            fz = as_c(np.concatenate([np.zeros(len(sp[5]), dtype=np.uint8) if f is None else f for sp, f in zip(species, fixed_zero)]), np.uint8)
        Cn = len(cand)
        x = np.zeros(max(Cn, 1)); ratio = np.zeros(max(Cn, 1), dtype=np.float32)
        obj = np.zeros(S); st = np.zeros(S, dtype=np.int32); it = np.zeros(S, dtype=np.int32)
        bi = _ffi.SpeciesBatch(S, p(node_off), p(node_len), p(ab), p(cov), p(hap_off), p(path_off), p(path_nodes), p(cand_off), p(cand), p(fz))
        bo = _ffi.SolutionBatch(p(x), p(ratio), p(obj), p(st), p(it))
        self._check(self.lib.pantax_hip_pao_solve_batch(self.ctx, C.byref(bi), C.byref(bo)))
        co = cand_off.astype(np.int64)
        return [(x[co[s]:co[s + 1]].copy(), ratio[co[s]:co[s + 1]].copy(), float(obj[s]), int(st[s]), int(it[s])) for s in range(S)]

    # ------------------------------------------------------------------ pipeline seam
    def profile(self, db, wd, gaf, species=True, strain=True, output_dir=None, fr=0.3, fc=0.46, sr=0.85, sd=0.2,
                min_species_abundance=1e-4, min_cov=0, min_depth=0, shift=False, filtered=True, full=True, force=False,
                mode=2, sample_nodes=0, designated_species=None, zip="serialize", out_binning_file=None,
                reads_binning_file=None, range_file=None, species_len_file=None, image_cache=0, rank=0, world_size=1,
                allreduce=None, alltoallv=None, sample_test=False, solver_semantics=0, minimization_min_cov=0.0):
        """profile::profile(ProfilingConfig) (profile.rs:3325): files in, files out.  allreduce(float64 array) sums in place over
        the ranks; alltoallv(send uint8 array, send_off [W+1], recv uint8 array, recv_off [W+1]) moves bytes between the ranks
        (host buffers) and switches on the sharded ingest."""
        enc = lambda x: None if x is None else str(x).encode()
        cfg = _ffi.ProfilingConfig(
            db=enc(db), wd=enc(wd), output_dir=enc(output_dir or wd), genomes_metadata=None, range_file=enc(range_file),
            input_aln_file=enc(gaf), species_len_file=enc(species_len_file), out_binning_file=enc(out_binning_file),
            reads_binning_file=enc(reads_binning_file), min_species_abundance=min_species_abundance,
            unique_trio_nodes_fraction=fr, unique_trio_nodes_mean_count_f=fc, single_cov_ratio=sr, single_cov_diff=sd,
            min_cov=min_cov, min_depth=min_depth, species=int(species), strain=int(strain), shift=int(shift),
            filtered=int(filtered), full=int(full), force=int(force), mode=mode, sample_nodes=sample_nodes,
            designated_species=enc(designated_species), zip=enc(zip), rank=int(rank), world_size=int(world_size),
            image_cache=int(image_cache), sample_test=int(sample_test), solver_semantics=int(solver_semantics),
            minimization_min_cov=float(minimization_min_cov))
        cb = None
        if allreduce is not None:   # allreduce(np.ndarray float64) sums it in place over the ranks
            def _cb(_user, buf, n):
                try:
                    allreduce(np.ctypeslib.as_array(buf, shape=(int(n),)))
                    return 0
                except Exception:   # noqa: BLE001 -- reported through the return code
                    return 1
            cb = _ffi.ALLREDUCE_FN(_cb)
            cfg.allreduce_sum = C.cast(cb, C.c_void_p)
        cb2 = None
        if alltoallv is not None:
            W = int(world_size)

            def _cb2(_user, send, send_off, recv, recv_off):
                try:
                    so = np.ctypeslib.as_array(send_off, shape=(W + 1,)).astype(np.int64)
                    ro = np.ctypeslib.as_array(recv_off, shape=(W + 1,)).astype(np.int64)
                    sb = (np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(int(so[-1]),)) if so[-1] else np.zeros(0, dtype=np.uint8))
                    rb = (np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(int(ro[-1]),)) if ro[-1] else np.zeros(0, dtype=np.uint8))
                    alltoallv(sb, so, rb, ro)
                    return 0
                except Exception:   # noqa: BLE001 -- reported through the return code
                    return 1
            cb2 = _ffi.ALLTOALLV_FN(_cb2)
            cfg.alltoallv = C.cast(cb2, C.c_void_p)
            cfg.comm_device_buffers = 0
        self._check(self.lib.pantax_hip_profile(self.ctx, C.byref(cfg)))

    def save_images(self, paths, hap_names):
        """one device-ready image per species of the resident db (graph + unique-trio index)."""
        ps = (C.c_char_p * self.S)(*[x.encode() for x in paths])
        hn = (C.c_char_p * max(self.H, 1))(*[x.encode() for x in hap_names])
        self._check(self.lib.pantax_hip_db_save_images(self.ctx, self.db, ps, hn))

    def load_images(self, paths, range_start, range_end, species):
        """Resident db from images; `species` (the same graphs as objects) only supplies the host-side shapes the
        wrapper keeps (node_off, hap_off) -- nothing of it is uploaded."""
        if self.db:
            self.lib.pantax_hip_db_free(self.ctx, self.db)
            self.db = None
            self._inflight = 0
        S = len(paths)
        self.S = S
        self.range_start = as_c(range_start, np.int64)
        self.range_end = as_c(range_end, np.int64)
        self.node_off = np.zeros(S + 1, dtype=np.uint64)
        self.node_off[1:] = np.cumsum([len(g.node_len) for g in species])
        self.hap_off = np.zeros(S + 1, dtype=np.uint64)
        self.hap_off[1:] = np.cumsum([len(g.path_off) - 1 for g in species])
        self.V, self.H = int(self.node_off[-1]), int(self.hap_off[-1])
        ps = (C.c_char_p * S)(*[x.encode() for x in paths])
        db = C.c_void_p()
        self._check(self.lib.pantax_hip_db_load_images(self.ctx, C.c_uint32(S), ps, p(self.range_start), p(self.range_end), C.byref(db)))
        self.db = db
        self.U = None

    def gaf_filter(self, gaf_path, out_path=None):
        """filter_max_alignment_mt (gaf_filter.rs:44-97) on the device -> (lines, records, lines written)."""
        nl, nr, nw = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.pantax_hip_gaf_filter(self.ctx, gaf_path.encode(), out_path.encode() if out_path else None,
                                                   C.byref(nl), C.byref(nr), C.byref(nw)))
        return nl.value, nr.value, nw.value

    @staticmethod
    def sample_ranks(n_valid, sample_nodes, seed=42):
        """a11: bool [n_valid], True where sample_sorted (profile.rs:1287-1295) keeps the row of that rank.  Host only."""
        bits = np.zeros((n_valid + 31) // 32, dtype=np.uint32)
        rc = _ffi.load().pantax_hip_sample_ranks(C.c_uint64(n_valid), C.c_uint64(sample_nodes), C.c_uint64(seed), p(bits))
        if rc != 0:
            raise ValueError("sample_ranks(%d, %d)" % (n_valid, sample_nodes))
        return np.unpackbits(bits.view(np.uint8), bitorder="little")[:n_valid].astype(bool)

    @staticmethod
    def chacha_block(key8, counter, rounds):
        key = np.ascontiguousarray(key8, dtype=np.uint32)
        out = np.zeros(16, dtype=np.uint32)
        if _ffi.load().pantax_hip_chacha_block(p(key), C.c_uint64(counter), int(rounds), p(out)) != 0:
            raise ValueError("chacha_block")
        return out

    # ------------------------------------------------------------------ timing
    def sort_rows(self, k0, k1, k2, algo=0):
        """Sort the rows (k0[i], k1[i], k2[i]) ascending as tuples on the device (0 auto, 1 radix, 2 sample sort)."""
        k = [np.ascontiguousarray(x, dtype=np.uint64).copy() for x in (k0, k1, k2)]
        self._check(self.lib.pantax_hip_sort_rows(self.ctx, C.c_uint64(len(k[0])), p(k[0]), p(k[1]), p(k[2]), int(algo)))
        return k

    def timing_enable(self, on=True):
        self._check(self.lib.pantax_hip_timing_enable(self.ctx, int(on)))

    def timing_filter(self, name=None):
        self._check(self.lib.pantax_hip_timing_filter(self.ctx, name.encode() if name else None))

    def timing_reset(self):
        self._check(self.lib.pantax_hip_timing_reset(self.ctx))

    def timing_get(self):
        cap = 64
        names = (C.c_char_p * cap)()
        launches = (C.c_uint64 * cap)()
        ms = (C.c_double * cap)()
        n = self.lib.pantax_hip_timing_get(self.ctx, cap, names, launches, ms)
        if n < 0:
            self._check(n)
        return {names[i].decode(): (int(launches[i]), float(ms[i])) for i in range(min(n, cap))}


def metrics_to_dicts(met, n=None):
    out = []
    for i, m in enumerate(met):
        if n is not None and i >= n:
            break
        d = {}
        for name, bit, attr in [("unique_trio_fraction", 1, "unique_trio_nodes_fraction"),
                                ("uniq_trio_cov_mean", 2, "frequencies_mean"), ("path_base_cov", 4, "path_cov_ratio"),
                                ("first_sol", 8, "first_sol"), ("strain_cov_diff", 16, "divergence"),
                                ("predicted_coverage", 32, "second_sol"), ("total_cov_diff", 128, "total_cov_diff")]:
            d[name] = getattr(m, attr) if m.has & bit else None
        d["is_rescue"] = bool(m.is_rescue) if m.has & 64 else None
        out.append(d)
    return out
