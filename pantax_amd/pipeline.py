"""One pass of the profiling hot path over device-resident inputs (the bench "step").

Mirrors profile::profile (profile.rs:3325-3364) from the point where the GAF is in memory:
rcls_profile -> species_profiling -> strain_profiling (load_species_range filter, trio index,
node coverage, PAO solves, abundace_constraint) -> abundance_est.  Species are sharded across
ranks; the only cross-rank exchange is ONE all-reduce carrying three f64 per species (its
predicted_coverage and its two strain-level sums: the normalisers of profile.rs:341, :3198, :3243
are sums of those) followed by a gather of the result rows to rank 0.
"""
from dataclasses import dataclass

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
    rebuild_trio: bool = True             # the reference rebuilds trio_nodes_info every run (profile.rs:2936)


class LocalComm:
    """world_size == 1: the exchange is the identity."""
    rank, world = 0, 1

    def all_gather(self, arr, nmax=None):
        return np.asarray(arr, dtype=np.float64)

    def gather_rows(self, rows):
        return rows


class TorchComm:
    """torch.distributed plumbing (backend nccl == RCCL on ROCm; gloo in the CPU tests)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device

    def all_gather(self, arr, nmax=None):
        """arr [n, k] float64 (n <= nmax on every rank) -> concatenation over ranks in rank order.
        ONE all-reduce(sum) of a zero-padded [world, nmax+1, k] slab: each rank fills only its own
        slice (row 0 carries its n), so the sum IS the gather and a single collective suffices."""
        arr = np.asarray(arr, dtype=np.float64)
        n, k = arr.shape
        nmax = n if nmax is None else nmax
        slab = np.zeros((self.world, nmax + 1, k), dtype=np.float64)
        slab[self.rank, 0, 0] = float(n)
        slab[self.rank, 1:n + 1] = arr
        t = self.torch.from_numpy(slab).to(self.device) if self.device is not None else self.torch.from_numpy(slab)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        slab = t.cpu().numpy()
        return np.concatenate([slab[r, 1:int(round(slab[r, 0, 0])) + 1] for r in range(self.world)], axis=0)

    def gather_rows(self, rows):
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(rows, out, dst=0)
        if self.rank != 0:
            return []
        return [r for part in out for r in part]


def local_stage(eng, avg_len, cfg, single_call=True):
    """Everything a rank computes on its own species shard (device stages + host filters).
    single_call: the whole pass through pantax_hip_profile_step (one host wait); False drives the same stages
    one C call at a time (rcls_profile, species_profiling, ... as the reference names them) -- identical results."""
    if single_call:
        keep, absolute, met, info, passed, s_all, s_pass = eng.profile_step(
            avg_len, fr=cfg.fr, fc=cfg.fc, sr=cfg.sr, sd=cfg.sd, min_cov=cfg.min_cov, min_depth=cfg.min_depth, shift=cfg.shift,
            filtered=cfg.filtered, rebuild_trio=cfg.rebuild_trio)
        keep, absolute, s_all, s_pass = keep.copy(), absolute.copy(), s_all.copy(), s_pass.copy()
        solved = np.array([1 if (keep[s] and info[s].status1 == 0 and info[s].status2 == 0) else 0
                           for s in range(eng.S)], dtype=np.uint8)
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
                                         min_depth=cfg.min_depth, shift=cfg.shift)
        solved = np.array([1 if (keep[s] and info[s].status1 == 0 and info[s].status2 == 0) else 0
                           for s in range(eng.S)], dtype=np.uint8)
        passed, s_all, s_pass = eng.abundance_filter(met, solved, cfg.sd, cfg.min_cov)
    rows = []   # candidate strain rows of this rank, before the global cut / normalisation
    for s in range(eng.S):
        if not solved[s]:
            continue
        for h in range(int(eng.hap_off[s]), int(eng.hap_off[s + 1])):
            if not passed[h]:
                continue
            m = met[h]
            opt = lambda bit, v: v if m.has & bit else None
            rows.append((s, h, m.second_sol, opt(4, m.path_cov_ratio), opt(1, m.unique_trio_nodes_fraction),
                         opt(2, m.frequencies_mean), opt(8, m.first_sol), opt(16, m.divergence), opt(128, m.total_cov_diff)))
    stats = dict(iters=[(info[s].iters1, info[s].iters2) for s in range(eng.S)],
                 n_rows=[info[s].n_rows for s in range(eng.S)], n_patterns=[info[s].n_patterns for s in range(eng.S)],
                 obj=[(info[s].obj1, info[s].obj2) for s in range(eng.S)])
    return dict(keep=keep, absolute=absolute, s_all=s_all, s_pass=s_pass, rows=rows, stats=stats)


def finalize_stage(local, species_names, hap_names, cfg, comm, shard_max=None):
    """The one cross-rank exchange + the final tables (pure host code: no device, testable under gloo)."""
    keep, absolute = local["keep"], local["absolute"]
    loc = np.stack([np.where(keep == 1, absolute, 0.0), local["s_all"], local["s_pass"]], axis=1)
    glob = comm.all_gather(loc, shard_max)
    total_abs = glob[:, 0].sum()                                    # profile.rs:341
    g_active = (glob[:, 0] > 0) & (glob[:, 0] / total_abs > cfg.min_species_abundance)   # profile.rs:602
    g_pass = glob[g_active, 2].sum()                                # profile.rs:3243
    abundance = np.where(keep == 1, absolute / total_abs if total_abs > 0 else 0.0, 0.0)
    active = (keep == 1) & (abundance > cfg.min_species_abundance)
    strain_rows = [(species_names[s], hap_names[h], cov, cov / g_pass) + tuple(rest)
                   for (s, h, cov, *rest) in local["rows"] if active[s]]
    species_rows = [(species_names[s], float(abundance[s]), float(absolute[s])) for s in range(len(keep)) if keep[s]]
    species_rows = comm.gather_rows(species_rows)
    strain_rows = comm.gather_rows(strain_rows)
    species_rows.sort(key=lambda r: -r[1])     # profile.rs:344
    strain_rows.sort(key=lambda r: -r[3])      # profile.rs:3247-3248
    return species_rows, strain_rows, int(active.sum())


def profile_step(eng, species_names, hap_names, avg_len, cfg=None, comm=None, shard_max=None, single_call=True):
    """Returns (species_rows, strain_rows, stats) on rank 0 (empty lists elsewhere).
    species_rows: (species_taxid, predicted_abundance, predicted_coverage) sorted descending.
    strain_rows : (species_taxid, hap_id, predicted_coverage, predicted_abundance, path_base_cov,
                   unique_trio_fraction, uniq_trio_cov_mean, first_sol, strain_cov_diff, total_cov_diff)."""
    cfg = cfg or StepConfig()
    comm = comm or LocalComm()
    local = local_stage(eng, avg_len, cfg, single_call)
    species_rows, strain_rows, n_active = finalize_stage(local, species_names, hap_names, cfg, comm, shard_max)
    stats = dict(local["stats"], n_active=n_active)
    return species_rows, strain_rows, stats
