"""One pass of the profiling hot path over device-resident inputs (the bench "step").

Mirrors profile::profile (profile.rs:3325-3364) from the point where the GAF is in memory:
rcls_profile -> species_profiling -> strain_profiling (load_species_range filter, trio index,
node coverage, PAO solves, abundace_constraint) -> abundance_est.  Species are sharded across
ranks; the only cross-rank exchange is ONE all-reduce per step: a small fixed-shape slab per rank with
three f64 per species (its predicted_coverage and its two strain-level sums: the normalisers of
profile.rs:341, :3198, :3243 are sums of those) and the candidate result rows; names are static
metadata gathered once.
"""
from dataclasses import dataclass

import os
import time
from concurrent.futures import ThreadPoolExecutor   # at import time: a first import inside a timed stream of steps costs milliseconds

import numpy as np


@dataclass
class StepConfig:
    fr: float = 0.3                       # --fr (short reads; 0.5 long), main.rs:108-114
    fc: float = 0.46                      # --fc
    sr: float = 0.85                      # --sr
    sd: float = 0.2                       # --sd
    min_species_abundance: float = 1e-4   # -a
    min_cov: int = 0
    min_depth: int = 0
    shift: bool = False
    filtered: bool = True
    sample_nodes: int = 0                 # --sample (cli.rs:227: 500000 by default); 0 = never sub-sample the LP rows
    rebuild_trio: bool = True             # the reference rebuilds trio_nodes_info every run (profile.rs:2936)
    solver_semantics: int = 0             # 0: Gurobi's handling of the second solve's solution (profile.rs:1500-1508), 1: highs_opt's (profile.rs:2865-2879)


def partition_species(weights, world):
    """Which rank takes which species: longest-processing-time packing -- heaviest first onto the least loaded
    rank, ties to the lower rank; the C file seam uses the same rule with weight = 8 * reads binned + graph nodes.
    -> list of rank per species."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0.0] * world
    owner = [0] * len(weights)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += weights[i]
    return owner


class LocalComm:
    """world_size == 1: the exchange is the identity."""
    rank, world = 0, 1

    def exchange(self, slab):
        return np.asarray(slab, dtype=np.float64)[None]

    def exchange_begin(self, slab):
        return np.asarray(slab, dtype=np.float64)[None]

    def exchange_end(self, handle):
        return handle

    def names(self, species_names, hap_names):
        return [list(species_names)], [list(hap_names)]

    def allreduce_sum(self, a):
        return np.asarray(a, dtype=np.float64).copy()

    def alltoall_counts(self, to):
        return np.asarray(to, dtype=np.int64).copy()

    def alltoallv_words(self, send, send_words, recv_words, on_device=False):
        return send


class _DeviceWords:
    """A window of device memory owned by libpantax_hip (u32 words at `addr`) as an object torch.as_tensor understands."""

    def __init__(self, addr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i4", "data": (int(addr), False), "version": 2}


class TorchComm:
    """torch.distributed plumbing (backend nccl == RCCL on ROCm; gloo in the CPU tests).  ONE collective per step."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self._names = None
        self._buf = None

    def exchange_begin(self, slab):
        """slab [rows, k] float64, same shape on every rank.  Starts ONE all-reduce(sum) of a zero-padded
        [world, rows, k] buffer: each rank fills only its own slice, so the sum IS the gather and a single small
        collective suffices (xGMI is point-to-point: the cost of a step's exchange is one latency-bound ring pass over
        a few KB).  Returns a handle for exchange_end; two buffers alternate, so one exchange may be in flight while
        the next step's kernels run (the collective executes on RCCL's own stream)."""
        slab = np.asarray(slab, dtype=np.float64)
        if self._buf is None or tuple(self._buf[0].shape[1:]) != slab.shape:
            dev = self.device if self.device is not None else "cpu"
            self._buf = [self.torch.zeros((self.world,) + slab.shape, dtype=self.torch.float64, device=dev) for _ in range(2)]
            self._turn = 0
        buf = self._buf[self._turn]
        self._turn ^= 1
        buf.zero_()
        buf[self.rank].copy_(self.torch.from_numpy(slab))
        work = self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, async_op=True)
        return (work, buf)

    def exchange_end(self, handle):
        """-> [world, rows, k] on the host."""
        work, buf = handle
        work.wait()
        return buf.cpu().numpy()

    def exchange(self, slab):
        return self.exchange_end(self.exchange_begin(slab))

    def thread_init(self):
        """a helper thread that issues collectives on this communicator binds to the same device first"""
        if self.device is not None:
            self.torch.cuda.set_device(self.device)

    # ---- the packed reads travel to the owner of their species (ingest time, once per input) ----------------
    def allreduce_sum(self, a):
        """host array -> its sum over the ranks (float64)"""
        dev = self.device if self.device is not None else "cpu"
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def alltoall_counts(self, to):
        """to [k, world] int64: what this rank sends to every rank (k quantities) -> [k, world]: what every rank sends here"""
        to = np.ascontiguousarray(to, dtype=np.int64)
        dev = self.device if self.device is not None else "cpu"
        k = to.shape[0]
        send = self.torch.from_numpy(np.ascontiguousarray(to.T)).to(dev)          # [world, k]: row j goes to rank j
        recv = self.torch.empty_like(send)
        self.dist.all_to_all_single(recv, send)
        return recv.cpu().numpy().T.reshape(k, self.world)

    def alltoallv_words(self, send, send_words, recv_words, on_device=False):
        """All-to-all(v) of 32-bit words: send_words[j] words go to rank j, recv_words[i] arrive from rank i.
        on_device (RCCL): `send` is the address of device memory, the result a CUDA int32 tensor (HBM to HBM over xGMI,
        nothing staged); otherwise `send` is a uint32 numpy array and so is the result (gloo)."""
        sw = [int(x) for x in send_words]
        rw = [int(x) for x in recv_words]
        if on_device:
            src = (self.torch.as_tensor(_DeviceWords(send, sum(sw)), device=self.device) if sum(sw)
                   else self.torch.empty(0, dtype=self.torch.int32, device=self.device))
            dst = self.torch.empty(sum(rw), dtype=self.torch.int32, device=self.device)
            self.dist.all_to_all_single(dst, src, output_split_sizes=rw, input_split_sizes=sw)
            self.torch.cuda.synchronize()
            return dst
        src = self.torch.from_numpy(np.ascontiguousarray(send, dtype=np.uint32).view(np.int32))
        dst = self.torch.empty(sum(rw), dtype=self.torch.int32)
        if self.device is not None:   # RCCL with host staging
            src, dst = src.to(self.device), dst.to(self.device)
        self.dist.all_to_all_single(dst, src, output_split_sizes=rw, input_split_sizes=sw)
        return dst.cpu().numpy().view(np.uint32)

    def names(self, species_names, hap_names):
        """Static metadata: gathered once, not per step."""
        if self._names is None:
            out = [None] * self.world
            self.dist.all_gather_object(out, (list(species_names), list(hap_names)))
            self._names = ([o[0] for o in out], [o[1] for o in out])
        return self._names


def route_reads(eng, owner_of_species, comm, on_device=None):
    """`eng` holds a SLICE of the reads, binned against the ranges of ALL species (eng.rcls_profile on a db of
    every species' range); owner_of_species[s] = rank that owns species s.  The packed records go to their owners
    (pantax_hip_reads_route_pack -> one all-to-all of the sizes, one all-to-all(v) of the messages -> pantax_hip_reads_from_
    routed) and become this rank's resident reads, in the one-process read order restricted to its species.
    -> dict(sent_reads, sent_words, recv_reads, recv_words)"""
    W = comm.world
    rt, nr_to, nt_to = eng.route_pack(owner_of_species, W)
    try:
        words_to = 5 * nr_to.astype(np.int64) + nt_to.astype(np.int64)
        got = comm.alltoall_counts(np.stack([nr_to.astype(np.int64), nt_to.astype(np.int64)]))
        nr_from, nt_from = got[0], got[1]
        words_from = 5 * nr_from + nt_from
        local = W == 1 and on_device is None          # nothing to exchange: hand the one message over in place
        if on_device is None:
            on_device = getattr(comm, "device", None) is not None
        if local:
            addr, _ = eng.route_buffer(rt, W, on_device=True)
            eng.reads_from_routed(addr, nr_from, nt_from, on_device=True)
        elif on_device:
            addr, _ = eng.route_buffer(rt, W, on_device=True)
            recv = comm.alltoallv_words(addr, words_to, words_from, on_device=True)
            eng.reads_from_routed(recv.data_ptr(), nr_from, nt_from, on_device=True)
            del recv
        else:
            send = np.concatenate(eng.route_messages(rt, W))
            recv = comm.alltoallv_words(send, words_to, words_from, on_device=False)
            eng.reads_from_routed(recv, nr_from, nt_from, on_device=False)
    finally:
        eng.route_free(rt)
    return dict(sent_reads=int(nr_to.sum()), sent_words=int(words_to.sum()), recv_reads=int(nr_from.sum()), recv_words=int(words_from.sum()))


# numpy layouts of pantax_hip_hap_metrics / pantax_hip_solve_info (include/pantax_hip.h; sizes checked against ctypes at import)
_MET_DT = np.dtype([("has", "<u4"), ("is_rescue", "<i4"), ("unique_trio_nodes_fraction", "<f8"), ("frequencies_mean", "<f8"), ("path_cov_ratio", "<f8"),
                    ("first_sol", "<f8"), ("divergence", "<f8"), ("second_sol", "<f8"), ("total_cov_diff", "<f8")])
_INFO_DT = np.dtype({"names": ["n_candidates", "status1", "status2", "iters1", "iters2", "n_rows", "n_patterns", "obj1", "obj2"],
                     "formats": ["<i4", "<i4", "<i4", "<i4", "<i4", "<u4", "<u4", "<f8", "<f8"], "offsets": [0, 4, 8, 12, 16, 20, 24, 32, 40], "itemsize": 48})


def _check_struct_layouts():
    import ctypes as C
    from . import _ffi
    assert C.sizeof(_ffi.HapMetrics) == _MET_DT.itemsize and C.sizeof(_ffi.SolveInfo) == _INFO_DT.itemsize
    assert _ffi.SolveInfo.obj1.offset == 32 and _ffi.HapMetrics.second_sol.offset == _MET_DT.fields["second_sol"][1]


_check_struct_layouts()


def local_enqueue(eng, avg_len, cfg):
    """The device part of a step, enqueued and not waited for (pantax_hip_profile_step_enqueue); local_stage(..., "collect")
    takes its results."""
    eng.profile_step_enqueue(avg_len, fr=cfg.fr, fc=cfg.fc, sr=cfg.sr, sd=cfg.sd, min_cov=cfg.min_cov, min_depth=cfg.min_depth, shift=cfg.shift,
                             filtered=cfg.filtered, rebuild_trio=cfg.rebuild_trio, sample_nodes=cfg.sample_nodes, solver_semantics=cfg.solver_semantics)


def local_stage(eng, avg_len, cfg, single_call=True):
    """Everything a rank computes on its own species shard (device stages + host filters).
    single_call: the whole pass through pantax_hip_profile_step (one host wait); False drives the same stages
    one C call at a time (rcls_profile, species_profiling, ... as the reference names them) -- identical results."""
    if single_call:
        if single_call == "collect":    # second half of a step enqueued earlier (local_enqueue)
            keep, absolute, met, info, passed, s_all, s_pass = eng.profile_step_collect()
        else:
            keep, absolute, met, info, passed, s_all, s_pass = eng.profile_step(
                avg_len, fr=cfg.fr, fc=cfg.fc, sr=cfg.sr, sd=cfg.sd, min_cov=cfg.min_cov, min_depth=cfg.min_depth, shift=cfg.shift,
                filtered=cfg.filtered, rebuild_trio=cfg.rebuild_trio, sample_nodes=cfg.sample_nodes, solver_semantics=cfg.solver_semantics)
        keep, absolute, s_all, s_pass = keep.copy(), absolute.copy(), s_all.copy(), s_pass.copy()
        if eng.S < 16:
            solved = np.array([1 if (keep[s] and info[s].status1 == 0 and info[s].status2 == 0) else 0 for s in range(eng.S)], dtype=np.uint8)
        else:
            inf0 = np.frombuffer(info, dtype=_INFO_DT, count=eng.S)
            solved = ((keep != 0) & (inf0["status1"] == 0) & (inf0["status2"] == 0)).astype(np.uint8)
    else:
        # a2 + a3 counters on device, a3 finishing on host
        _, rc, bs, lm, uq = eng.rcls_profile(want_species=False)
        keep, absolute, _ = eng.species_profiling((rc, bs, lm, uq), avg_len, filtered=cfg.filtered)
        # a7 (rebuilt per run like the reference), a8, a9..a14 for every species that survived the MAPQ
        # filter.  The -a abundance cut (profile.rs:602) needs the GLOBAL normaliser, so it is applied
        # after the single exchange in finalize_stage; species are independent, so computing the few
        # low-abundance ones too changes nothing else.
        if cfg.rebuild_trio:
            eng.db_reset()
        eng.trio_nodes_info(fetch=False)
        eng.get_node_abundances(species_active=keep, fetch=False)
        met, info = eng.strain_profiling(absolute, species_active=keep, fr=cfg.fr, fc=cfg.fc, sr=cfg.sr,
                                         min_depth=cfg.min_depth, shift=cfg.shift, sample_nodes=cfg.sample_nodes, solver_semantics=cfg.solver_semantics)
        solved = np.array([1 if (keep[s] and info[s].status1 == 0 and info[s].status2 == 0) else 0
                           for s in range(eng.S)], dtype=np.uint8)
        passed, s_all, s_pass = eng.abundance_filter(met, solved, cfg.sd, cfg.min_cov)
    # candidate strain rows of this rank, before the global cut / normalisation.  The C structs are read through numpy views
    # (a thousand strains per step: field-by-field ctypes access was 0.2 ms of host time between two steps' kernels)
    H = int(eng.hap_off[-1])
    if H < 64:   # a handful of strains: plain field access is cheaper than setting up the views
        rows = []
        for s in range(eng.S):
            if not solved[s]:
                continue
            for h in range(int(eng.hap_off[s]), int(eng.hap_off[s + 1])):
                if not passed[h]:
                    continue
                mm = met[h]
                o = lambda bit, v: v if mm.has & bit else None
                rows.append((s, h, mm.second_sol, o(4, mm.path_cov_ratio), o(1, mm.unique_trio_nodes_fraction), o(2, mm.frequencies_mean),
                             o(8, mm.first_sol), o(16, mm.divergence), o(128, mm.total_cov_diff)))
        stats = dict(iters=[(info[s].iters1, info[s].iters2) for s in range(eng.S)], n_cand=[info[s].n_candidates for s in range(eng.S)],
                     n_rows=[info[s].n_rows for s in range(eng.S)], n_patterns=[info[s].n_patterns for s in range(eng.S)],
                     obj=[(info[s].obj1, info[s].obj2) for s in range(eng.S)])
        return dict(keep=keep, absolute=absolute, s_all=s_all, s_pass=s_pass, rows=rows, stats=stats)
    m = np.frombuffer(met, dtype=_MET_DT, count=H) if H else np.zeros(0, dtype=_MET_DT)
    inf = np.frombuffer(info, dtype=_INFO_DT, count=eng.S)
    sp_of_hap = np.repeat(np.arange(eng.S), np.diff(eng.hap_off.astype(np.int64)))
    sel = np.nonzero((np.asarray(passed[:H]) != 0) & (solved[sp_of_hap] != 0))[0]
    has = m["has"][sel]
    # the candidate rows as ONE array in the layout of the exchanged slab (finalize_begin): no Python object per strain between a step's collect and its
    # tables (round 6: the tuples of 1e4 strains were 6.8 ms a step on the helper thread -- and the last step's tables are inside every timed region)
    rows_np = np.zeros((len(sel), _ROW_K))
    rows_np[:, 0], rows_np[:, 1], rows_np[:, 2] = sp_of_hap[sel], sel, m["second_sol"][sel]
    for i, (bit, name) in enumerate(_OPT_COLS):
        f = (has & bit) != 0
        rows_np[:, 3] += f * float(1 << i)
        rows_np[:, 4 + i] = np.where(f, m[name][sel], 0.0)
    stats = dict(iters=list(zip(inf["iters1"].tolist(), inf["iters2"].tolist())), n_cand=inf["n_candidates"].tolist(),
                 n_rows=inf["n_rows"].tolist(), n_patterns=inf["n_patterns"].tolist(), obj=list(zip(inf["obj1"].tolist(), inf["obj2"].tolist())))
    return dict(keep=keep, absolute=absolute, s_all=s_all, s_pass=s_pass, rows_np=rows_np, stats=stats)


_ROW_K = 10   # columns of the exchanged slab
# the Option<f64> columns of a strain row, in the slab's order: (bit of pantax_hip_hap_metrics.has, field)
_OPT_COLS = [(4, "path_cov_ratio"), (1, "unique_trio_nodes_fraction"), (2, "frequencies_mean"), (8, "first_sol"), (16, "divergence"), (128, "total_cov_diff")]


def rows_to_array(rows):
    """candidate strain rows as tuples (species, hap, coverage, six Option values) -> the slab's layout [n, _ROW_K]: species, hap, coverage,
    bit i = option i is Some, the six values (0.0 for None)"""
    out = np.zeros((len(rows), _ROW_K))
    for j, (s_, h_, cov, *opt) in enumerate(rows):
        r = out[j]
        r[0], r[1], r[2] = s_, h_, cov
        r[3] = sum(1 << i for i, v in enumerate(opt) if v is not None)
        r[4:4 + len(opt)] = [0.0 if v is None else v for v in opt]
    return out
PIPELINE_THREAD_MIN_HAPS = 128   # profile_steps_pipelined: from this many strains per rank the tables are built on a helper thread


def finalize_begin(local, hap_names, comm, shard_max=None, rows_max=None):
    """First half of finalize_stage: pack this rank's slab and START the exchange; returns a handle for finalize_end.
    The collective runs while the caller enqueues its next step."""
    keep, absolute = local["keep"], local["absolute"]
    rows = local["rows_np"] if "rows_np" in local else rows_to_array(local["rows"])
    S_loc = len(keep)
    S_max = S_loc if shard_max is None else shard_max
    R_max = max(len(hap_names), 1) if rows_max is None else rows_max
    slab = np.zeros((1 + S_max + R_max, _ROW_K))
    slab[0, 0], slab[0, 1] = S_loc, len(rows)
    slab[1:1 + S_loc, 0] = keep
    slab[1:1 + S_loc, 1] = np.where(keep == 1, absolute, 0.0)
    slab[1:1 + S_loc, 2] = local["s_all"]
    slab[1:1 + S_loc, 3] = local["s_pass"]
    slab[1 + S_max:1 + S_max + len(rows)] = rows
    return comm.exchange_begin(slab), S_max


def finalize_stage(local, species_names, hap_names, cfg, comm, shard_max=None, rows_max=None):
    """The one cross-rank exchange + the final tables (pure host code: no device, testable under gloo).
    Every rank contributes one fixed-shape slab: [n_species, n_rows | species: keep, predicted_coverage, sum of all
    strain coverages, sum of passing | candidate strain rows: species, hap, coverage, Option bits, six metrics].
    shard_max / rows_max: upper bounds of species / strain rows per rank (identical on all ranks)."""
    return finalize_end(finalize_begin(local, hap_names, comm, shard_max, rows_max), species_names, hap_names, cfg, comm)


def finalize_end(pending, species_names, hap_names, cfg, comm):
    """Second half: wait for the exchange, derive the global normalisers, build the tables on rank 0."""
    handle, S_max = pending
    glob = comm.exchange_end(handle)                                 # [world, rows, K]
    W = glob.shape[0]
    n_sp = [int(round(glob[r, 0, 0])) for r in range(W)]
    n_rw = [int(round(glob[r, 0, 1])) for r in range(W)]
    sp_blk = [glob[r, 1:1 + n_sp[r]] for r in range(W)]
    total_abs = sum(float(b[:, 1].sum()) for b in sp_blk)            # profile.rs:341
    act = [(b[:, 0] == 1) & (b[:, 1] > 0) & (b[:, 1] / total_abs > cfg.min_species_abundance) for b in sp_blk]   # profile.rs:602
    g_pass = sum(float(b[a, 3].sum()) for b, a in zip(sp_blk, act))  # profile.rs:3243
    # a rank counts its own species; the K dbs of one GPU (finalize_many) are one rank's species: all of them
    n_active = int(sum(a.sum() for a in act)) if getattr(comm, "counts_every_block", False) else int(act[comm.rank].sum())
    all_sn, all_hn = comm.names(species_names, hap_names)          # collective on its first call only (cached)
    if comm.rank != 0:
        return [], [], n_active
    # the tables column by column (numpy), one tuple per row only at the very end; both sorts are stable and descending, like the list sorts they replace
    species_rows, strain_rows, sp_key, st_key = [], [], [], []
    for r in range(W):
        b = sp_blk[r]
        kept = np.nonzero(b[:, 0] == 1)[0]
        ab = b[kept, 1] / total_abs
        sn_r, hn_r = all_sn[r], all_hn[r]
        species_rows += list(zip([sn_r[s] for s in kept.tolist()], ab.tolist(), b[kept, 1].tolist()))
        sp_key.append(ab)
        blk = glob[r, 1 + S_max:1 + S_max + n_rw[r]]
        s_idx = np.rint(blk[:, 0]).astype(np.int64)
        on = act[r][s_idx] if len(s_idx) else np.zeros(0, dtype=bool)
        blk, s_idx = blk[on], s_idx[on]
        h_idx = np.rint(blk[:, 1]).astype(np.int64)
        has = np.rint(blk[:, 3]).astype(np.int64)
        st_ab = blk[:, 2] / g_pass if len(blk) else np.zeros(0)
        opt = [[v if f else None for v, f in zip(blk[:, 4 + i].tolist(), ((has >> i) & 1).astype(bool).tolist())] for i in range(6)]
        strain_rows += list(zip([sn_r[s] for s in s_idx.tolist()], [hn_r[h] for h in h_idx.tolist()], blk[:, 2].tolist(), st_ab.tolist(), *opt))
        st_key.append(st_ab)
    if species_rows:
        species_rows = [species_rows[i] for i in np.argsort(-np.concatenate(sp_key), kind="stable").tolist()]     # profile.rs:344
    if strain_rows:
        strain_rows = [strain_rows[i] for i in np.argsort(-np.concatenate(st_key), kind="stable").tolist()]       # profile.rs:3247-3248
    return species_rows, strain_rows, n_active


def profile_steps_pipelined(eng, species_names, hap_names, avg_len, n_steps, cfg=None, comm=None, shard_max=None, rows_max=None,
                            next_input=None, threaded=None):
    """n_steps passes back to back (a stream of samples) without the device ever waiting for the host between two of them:
    * step i+1 is ENQUEUED before step i is collected (pantax_hip_profile_step_enqueue / _collect, two result slots): the
      device runs the main-stream work of the steps one after the other -- only the unique-trio rebuild of step i+1 starts
      earlier, behind step i's first filter (the index's last reader) -- and the ~60 launches of a step, the host wait, the parsing of the result arena and the Python around the
      call no longer sit between two steps' kernels;
    * from PIPELINE_THREAD_MIN_HAPS strains per rank on, everything that follows a step's collect -- packing the slab, the
      one all-reduce, the normalisers and the tables (~0.6 ms of host code for 1000 strains) -- runs on a helper thread;
      below that the exchange of step i is started inline and completed after step i+1's collect.
    Collectives are issued in step order on every rank.  next_input(i), if given, is called before step i is enqueued to
    swap in that step's reads (the swap itself waits for the step that still uses the old ones).  threaded: helper thread or
    inline; with more than one rank pass a value that is the SAME on every rank (rows_max >= PIPELINE_THREAD_MIN_HAPS is what
    bench.py passes): the default looks at this rank's own strains, and ranks of a strong-scaling run own different numbers.
    Returns the list of (species_rows, strain_rows, stats), same as n_steps calls of profile_step."""
    cfg = cfg or StepConfig()
    comm = comm or LocalComm()
    if n_steps <= 0:
        return []

    def finish(local):
        sr, tr, n_active = finalize_stage(local, species_names, hap_names, cfg, comm, shard_max, rows_max)
        return sr, tr, dict(local["stats"], n_active=n_active)

    _enq_ms = []

    def enqueue(i):
        if next_input is not None:
            next_input(i)
        t_e = time.perf_counter()
        local_enqueue(eng, avg_len, cfg)
        _enq_ms.append((time.perf_counter() - t_e) * 1e3)

    if threaded is None:
        threaded = (rows_max if rows_max is not None else len(hap_names)) >= PIPELINE_THREAD_MIN_HAPS
    out, fut, pending = [], None, None
    ex = ThreadPoolExecutor(1, initializer=getattr(comm, "thread_init", None) or (lambda: None)) if threaded else None
    try:
        _trace = [time.perf_counter()] if os.environ.get("PANTAX_PIPE_TRACE") else None   # debug: when every step was collected
        lookahead = os.environ.get("PANTAX_STEP_LOOKAHEAD", "1") != "0"    # 0: enqueue step i+1 only after step i's collect (measurements)
        enqueue(0)
        for i in range(n_steps):
            if lookahead and i + 1 < n_steps:
                enqueue(i + 1)                                  # the device goes straight on after step i
            local = local_stage(eng, avg_len, cfg, "collect")   # step i's one host wait
            if _trace is not None:
                _trace.append(time.perf_counter())
            if not lookahead and i + 1 < n_steps:
                enqueue(i + 1)
            if threaded:
                if fut is not None:
                    out.append(fut.result())                    # step i-1's tables: built while step i ran
                fut = ex.submit(finish, local)
            else:
                nxt = (finalize_begin(local, hap_names, comm, shard_max, rows_max), local["stats"])
                if pending is not None:
                    sr, tr, n_active = finalize_end(pending[0], species_names, hap_names, cfg, comm)
                    out.append((sr, tr, dict(pending[1], n_active=n_active)))
                pending = nxt
        if fut is not None:
            out.append(fut.result())
        if pending is not None:
            sr, tr, n_active = finalize_end(pending[0], species_names, hap_names, cfg, comm)
            out.append((sr, tr, dict(pending[1], n_active=n_active)))
        if _trace is not None:
            _trace.append(time.perf_counter())                  # (the last step's tables)
    finally:
        if ex is not None:
            ex.shutdown()
        if _trace is not None:
            _trace.append(time.perf_counter())                  # (the helper thread joined)
        if _trace is not None:
            import sys
            print("[pipelined] ms between collects:", " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip(_trace, _trace[1:])), file=sys.stderr)
            print("[pipelined] ms per enqueue call:", " ".join("%.2f" % x for x in _enq_ms), file=sys.stderr)
        eng.drain_steps()       # only non-empty after an exception between an enqueue and its collect
    return out


def profile_step(eng, species_names, hap_names, avg_len, cfg=None, comm=None, shard_max=None, single_call=True, rows_max=None):
    """Returns (species_rows, strain_rows, stats) on rank 0 (empty lists elsewhere).
    species_rows: (species_taxid, predicted_abundance, predicted_coverage) sorted descending.
    strain_rows : (species_taxid, hap_id, predicted_coverage, predicted_abundance, path_base_cov,
                   unique_trio_fraction, uniq_trio_cov_mean, first_sol, strain_cov_diff, total_cov_diff)."""
    cfg = cfg or StepConfig()
    comm = comm or LocalComm()
    local = local_stage(eng, avg_len, cfg, single_call)
    species_rows, strain_rows, n_active = finalize_stage(local, species_names, hap_names, cfg, comm, shard_max, rows_max)
    stats = dict(local["stats"], n_active=n_active)
    return species_rows, strain_rows, stats


# ---------------------------------------------------------------------------------------------- several dbs on ONE GPU
# A resident db addresses its path steps with 32 bits (include/pantax_hip.h: PANTAX_HIP_E_LIMIT beyond 2^32 - 1 positions).  A database
# of more path steps than that -- BASELINE configs[4]: 1 000 species x 50 strains = 1.1e10 -- is cut BY SPECIES into several dbs that share
# the GPU: species are independent from a4 on (profile.rs:3297-3319), so every db is stepped on its own ctx (its own streams: the device
# interleaves them) with the reads of its species, and the dbs' results meet exactly like those of ranks -- the slabs of finalize_begin,
# the global normalisers of profile.rs:341, :3198, :3243 in finalize_end -- only without a collective: they are in one process.
class _SlabComm:
    rank, world = 0, 1

    def exchange_begin(self, slab):
        return np.asarray(slab, dtype=np.float64)


class _ManyComm:
    rank = 0
    counts_every_block = True

    def __init__(self, species_names_list, hap_names_list):
        self.world = len(species_names_list)
        self._names = ([list(x) for x in species_names_list], [list(x) for x in hap_names_list])

    def exchange_end(self, handle):
        return handle

    def names(self, species_names, hap_names):
        return self._names


def split_species_by_path_steps(path_steps, limit=3_000_000_000):
    """contiguous groups of species whose path steps sum to at most `limit` (the C seam's cap, api_profile.cpp `db_path_steps_max`: the visit
    table's pads -- about 20 % at 50 strains -- share the 32-bit slots with the steps), as few groups as
    that takes and about equally heavy -> list of (first species, end species)"""
    ps = [int(p) for p in path_steps]
    for i, p in enumerate(ps):
        if p > limit:
            raise ValueError("species %d alone has %d path steps" % (i, p))
    total = sum(ps)
    K = max(1, -(-total // limit))
    while True:
        target = -(-total // K)
        groups, a, acc = [], 0, 0
        for i, p in enumerate(ps):
            if acc and (acc + p > limit or (acc + p > target and len(groups) < K - 1)):
                groups.append((a, i))
                a, acc = i, 0
            acc += p
        groups.append((a, len(ps)))
        if all(sum(ps[x:y]) <= limit for x, y in groups):
            return groups
        K += 1


def finalize_many(locals_, species_names_list, hap_names_list, cfg):
    """the tables of one sample over K dbs: every db's local stage -> one slab each -> the same finalisation K ranks get"""
    S_max = max(len(l["keep"]) for l in locals_)
    R_max = max(max(len(h), 1) for h in hap_names_list)
    slabs = [finalize_begin(l, h, _SlabComm(), S_max, R_max)[0] for l, h in zip(locals_, hap_names_list)]
    sr, tr, n_active = finalize_end((np.stack(slabs), S_max), None, None, cfg, _ManyComm(species_names_list, hap_names_list))
    stats = {k: [x for l in locals_ for x in l["stats"][k]] for k in locals_[0]["stats"]}
    return sr, tr, dict(stats, n_active=n_active)


def profile_steps_many(engs, species_names_list, hap_names_list, avg_len_list, n_steps, cfg=None, one_after_the_other=False):
    """n_steps samples over K dbs that share the GPU, one step enqueued ahead on every db (see profile_steps_pipelined) -> list of
    (species_rows, strain_rows, stats).  one_after_the_other: every db's step is collected before the next db's is enqueued -- nothing
    overlaps, so a per-kernel clock sees every kernel with the GPU to itself (measurements; slower)."""
    cfg = cfg or StepConfig()
    K = len(engs)
    out = []
    if one_after_the_other:
        for i in range(n_steps):
            locals_ = []
            for k in range(K):
                local_enqueue(engs[k], avg_len_list[k], cfg)
                locals_.append(local_stage(engs[k], avg_len_list[k], cfg, "collect"))
            out.append(finalize_many(locals_, species_names_list, hap_names_list, cfg))
        return out
    try:
        for k in range(K):
            local_enqueue(engs[k], avg_len_list[k], cfg)
        for i in range(n_steps):
            if i + 1 < n_steps:
                for k in range(K):
                    local_enqueue(engs[k], avg_len_list[k], cfg)
            locals_ = [local_stage(engs[k], avg_len_list[k], cfg, "collect") for k in range(K)]
            out.append(finalize_many(locals_, species_names_list, hap_names_list, cfg))
    finally:
        for e in engs:
            e.drain_steps()
    return out
