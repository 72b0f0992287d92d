#!/usr/bin/env python3
"""bench.py -- PAO wall time / Mreads/s of the profiling hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM: read
binning + species counters, species profile, unique-trio index (rebuilt per step like the reference does per run,
profile.rs:2936), node-coverage histogram, LP row grouping, the two PAO solves, filters and the abundance table.

Workload at N=1 (default) = BASELINE.json configs[2], the largest configuration BASELINE.json writes for one GPU:
"100 species / 1k strains, 10M Illumina GAF, 1 MI355X" (10 strains per species, 5 Mbp genomes, 150 bp
reads, seed 20260501 + 3).  `--workload cfg2` selects configs[1] (1 species, 1M reads); `--workload cfg4` selects
configs[3], the 8-GPU configuration the metric is quoted on (1k species / 10k strains, 100M reads): it fits ONE MI355X
(V = 3.2e8 nodes, P = 2.2e9 path steps, T = 7.6e8 walk steps; measured once at 74 ms per step = 1.35 Greads/s,
profiles/r02_bench_cfg4_one_gpu_quick.json) but stays opt-in: the default run must be quick and safe on any box (a second
cfg4 run with the CPU-baseline legs lost its GPU box before it reported, cause undetermined).  With N
ranks every rank owns its own shard of the default shape (weak scaling: N x 100 species / N x 10M reads, i.e. cfg4's 1k
species / 100M reads at N = 8 give or take 25 %; species are independent sub-problems) and one RCCL all-reduce per step
carries the normalisers.  `--scaling strong` instead cuts ONE set of the given size over the ranks the way the file seam does
(SURVEY 8e): every rank takes a 1/N slice of the READS, bins it against all species ranges, the species are packed onto
the ranks by weight (longest processing time first, pipeline.partition_species) and the packed records travel to the
owner of their species in one RCCL all-to-all(v) over xGMI (pipeline.route_reads) -- once, before the timed steps
(`ingest_route`); at N = 1 it is the default workload.  Larger workloads keep the extra legs bounded: the CPU baseline runs
on the first 25M reads (all species, all host cores), the from-GAF-text leg on the first 10M reads.

    python bench.py                                   # cfg3, 1 GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Beside `value` the line carries: `roofline` (dominant kernel, HIP events on the library's stream), `cpu_baseline` (the
plain-C oracle on ALL host cores, one species per worker like profile.rs:3297-3319, plus SciPy-HiGHS legs on row samples
of the same LP), `from_gaf_text` (GAF text on disk -> tables, PCIe + device tokenizer included), `pao_hard` (a
synthetic variant whose first filter keeps all ten strains: the LP regime BASELINE.md section 2 flags).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {   # name: (BASELINE.json config, seed offset, species, haps, reads, genome_len)
    "cfg2": ("configs[1]: single-species E. coli-like, 10 strains, 1M short reads", 2, 1, 10, 1_000_000, 5_000_000),
    "cfg3": ("configs[2]: 100 species / 1k strains, 10M short reads", 3, 100, 10, 10_000_000, 5_000_000),
    "cfg4_share": ("configs[3] per-GPU share: 125 species / 1250 strains, 12.5M short reads", 4, 125, 10, 12_500_000, 5_000_000),
    # the configuration BASELINE.json's metric is quoted on ("at 10k strains"): it fits ONE MI355X (~100 GB of the 288 GB)
    "cfg4": ("configs[3]: 1k species / 10k strains, 100M short reads -- the whole 8-GPU configuration", 5, 1000, 10, 100_000_000, 5_000_000),
}
CPU_SAMPLE_READS = 25_000_000    # the CPU baseline runs on the first reads of a larger workload (a bounded sample, ~20-30 s of all cores)
GAF_SAMPLE_READS = 10_000_000    # the from-GAF-text leg writes / loads at most this many reads (1.4 GB of text)


def algorithmic_bytes(sset, n_lp_rows, U, R, T):
    """SURVEY.md section 8d per-stage compulsory traffic for ONE step of this rank's workload (U = unique trios, R reads,
    T walk steps resident on this rank)."""
    V = sum(g.n_nodes for g in sset.species)
    L = int(sum(int(g.node_len.sum()) for g in sset.species))
    P = int(sum(int(g.path_off[-1]) for g in sset.species))
    H = sum(g.n_paths for g in sset.species)
    Wr = max(T - 2 * R, 0)
    return {
        # a2: 4T + 4R(offsets) + 4R(out)
        "bin_reads_kernel": 4 * T + 4 * R + 4 * R,
        # a8 minus the popcount pass: 4T + 12R + 4V(node_len) + 8V(bases) + L/8(bitmap) + 12*Wr(trio probes)
        "coverage_step_kernel": 4 * T + 12 * R + 4 * V + 8 * V + L // 8 + 12 * Wr,
        # a8 popcount: L/8 bitmap in + 8V cov out
        "popcount_kernel": L // 8 + 8 * V,
        # a7: SURVEY 8d "2 x 12 x (P - 2H) (write keys, read sorted) + 12U" is the whole index; the bucket scatter alone
        # reads the walks (4P) and writes one 16-B record per window
        # node-block path of the index (the default): SURVEY 8d's whole-index figure 2 x 12 x (P - 2H) + 12U split over its
        # two passes over the walks -- the block kernel forms every window's 12-byte key and decides count == 1 (the "write
        # keys" half, plus the 4P of walk it reads, which 8d leaves out), the lookup kernel reads the decided keys back in
        # walk order and writes the U unique rows (the "read sorted" half + 12U)
        "trio_block_kernel": 4 * P + 12 * max(P - 2 * H, 0),
        "trio_lookup_kernel": 12 * max(P - 2 * H, 0) + 12 * U,
        "trio_fill_kernel": 4 * P + 16 * max(P - 2 * H, 0),
        "trio_count_kernel": 4 * P + 4 * V,
        "trio_uniq_kernel": 16 * max(P - 2 * H, 0),
        # a10: 4P in + 8V mask out
        "mask_kernel": 4 * P + 8 * V,
        "sort_hist_kernel": 8 * n_lp_rows,
        "sort_scatter_kernel": 2 * 24 * n_lp_rows,
    }, dict(R=R, T=T, V=V, L=L, P=P, H=H)


# ------------------------------------------------------------------------------------------------ CPU baseline
_CPU = {}   # state shared with forked workers (copy-on-write)


def _cpu_bin_task(rng):
    from oracle import oracle as orc
    a, b = rng
    sset = _CPU["sset"]
    rd = sset.reads
    return orc.bin_reads(rd.step_off[a:b + 1], rd.node_id, _CPU["rs"], _CPU["re"])


def _species_lp(orc, g, G, b, c, md):
    """The LP optimize_species solved for this species: candidate columns = haplotypes with a first_sol."""
    cand = np.array([i for i, m in enumerate(md) if m["first_sol"] is not None], dtype=np.uint32)
    if len(cand) == 0:
        return None
    mask, _ = orc.path_masks(G, cand, c)
    ab = b / g.node_len
    return mask, ab, len(cand)


def _cpu_species_task(si):
    from oracle import oracle as orc
    from tests.helpers import select_reads
    sset, cfg = _CPU["sset"], _CPU["cfg"]
    g = sset.species[si]
    t0 = time.perf_counter()
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
    T = orc.TrioTable(G)
    t1 = time.perf_counter()
    lo, n = _CPU["first"][si], _CPU["cnt"][si]
    sel = np.sort(_CPU["order"][lo:lo + n])
    so, nid, ps, pe = select_reads(sset.reads, sel)
    b, c, tb, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
    t2 = time.perf_counter()
    rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=cfg["fr"], fc=cfg["fc"], sr=cfg["sr"])
    orc.abundance_constraint(_CPU["absolute"][si], met)
    t3 = time.perf_counter()
    n_rows = int(((b > 0)).sum())
    return si, t1 - t0, t2 - t1, t3 - t2, nc, n_rows


def _highs_task(job):
    """SciPy's bundled HiGHS on the first `rows` (random order, seed 0) valid rows of one species' LP (the reference's
    open backend, highs_opt profile.rs:2689-2882: x in [0, 1.05 max a], y_v >= +-(A x - a)_v, min (1/n) sum y), next
    to the oracle's exact LAD on the same rows."""
    from oracle import oracle as orc
    from scipy import sparse
    from scipy.optimize import linprog
    import scipy
    mask, ab, p, rows, tlimit = job
    valid = np.nonzero((ab > 0))[0]
    ub = 1.05 * float(ab.max())
    take = valid[np.random.default_rng(0).permutation(len(valid))[:rows]]
    take.sort()
    m, a = mask[take], ab[take]
    n = len(take)
    A = sparse.csr_matrix(np.stack([((m >> np.uint64(k)) & np.uint64(1)).astype(float) for k in range(p)], 1))
    I = sparse.identity(n, format="csr")
    Aub = sparse.vstack([sparse.hstack([A, -I]), sparse.hstack([-A, -I])]).tocsr()
    bub = np.concatenate([a, -a])
    c = np.concatenate([np.zeros(p), np.ones(n) / n])
    bounds = [(0, ub)] * p + [(0, None)] * n
    t0 = time.perf_counter()
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=bounds, method="highs", options={"time_limit": float(tlimit)})
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    x, obj, it, st = orc.lad_solve(m, a, p, np.full(p, ub))
    dt_lad = time.perf_counter() - t1
    return dict(rows=n, columns=p, patterns=int(len(np.unique(m))), highs_seconds=dt, highs_status=int(r.status),
                highs_objective=(float(r.fun) if r.status == 0 else None), exact_lad_objective=float(obj), exact_lad_seconds=dt_lad,
                time_limit_s=tlimit, scipy=scipy.__version__)


def highs_legs(pool, lp, sizes, tlimit):
    """-> list of per-size results, or a note when SciPy is not importable on this box."""
    try:
        import scipy.optimize  # noqa: F401
    except Exception as e:   # noqa: BLE001
        return {"note": "SciPy not importable here (%s): no HiGHS leg" % type(e).__name__}
    mask, ab, p = lp
    n_valid = int((ab > 0).sum())
    jobs = [(mask, ab, p, min(s, n_valid), tlimit) for s in sizes]
    return pool.map(_highs_task, jobs)


def cpu_baseline(sset, cfg, cores, highs_sizes, highs_tlimit, n_limit=None):
    """The oracle (plain-C port of the reference algorithm) on ALL host cores over the whole workload: reads binned in
    `cores` slices, then one species per worker (the reference's rayon par_iter over species, profile.rs:3297-3319)
    through trio index, coverage, filters and both LP solves (the oracle's own exact LAD solver); the reference's open
    solver, HiGHS, is timed on row samples of the largest species' LP (the full LP does not finish in minutes,
    BASELINE.md section 2).  Runs BEFORE the GPU is initialised (forked workers)."""
    import multiprocessing as mp
    from oracle import oracle as orc
    rd = sset.reads
    S = len(sset.species)
    n = min(rd.n_reads, n_limit) if n_limit else rd.n_reads     # a read prefix of a larger workload, all species
    _CPU.update(sset=sset, cfg=dict(fr=cfg.fr, fc=cfg.fc, sr=cfg.sr),
                rs=[g.range_start for g in sset.species], re=[g.range_end for g in sset.species])
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    cuts = np.linspace(0, n, cores + 1).astype(np.int64)
    with ctx.Pool(cores) as pool:
        parts = pool.map(_cpu_bin_task, [(int(cuts[i]), int(cuts[i + 1])) for i in range(cores)])
    sp = np.concatenate(parts)
    counts = orc.species_counts(sp, rd.qlen[:n], rd.mapq[:n], S)
    keep, absolute, _ = orc.species_profile(sp, rd.qlen[:n], counts, sset.avg_len())
    t_bin = time.perf_counter() - t0
    order = np.argsort(sp, kind="stable")
    cnt = np.bincount(sp[sp >= 0], minlength=S)
    first = np.searchsorted(sp[order], np.arange(S))
    t_group = time.perf_counter() - t0 - t_bin
    _CPU.update(order=order, cnt=cnt, first=first, absolute=absolute)
    todo = [si for si in range(S) if keep[si]]
    todo.sort(key=lambda si: -int(cnt[si]))                     # heaviest first
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_species_task, todo, chunksize=1)
        dt = time.perf_counter() - t0
        # HiGHS legs: the LP of the species with the most rows, on bounded row samples
        highs = None
        if res and highs_sizes:
            si = max(res, key=lambda r: r[5])[0]
            g = sset.species[si]
            from tests.helpers import select_reads
            G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
            T = orc.TrioTable(G)
            sel = np.sort(order[first[si]:first[si] + cnt[si]])
            so, nid, ps, pe = select_reads(rd, sel)
            b, c, tb, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
            rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=cfg.fr, fc=cfg.fc, sr=cfg.sr)
            lp = _species_lp(orc, g, G, b, c, orc.metrics_to_dicts(met))
            if lp is not None:
                highs = highs_legs(pool, lp, highs_sizes, highs_tlimit)
                if isinstance(highs, list):
                    highs = {"species": g.name, "full_lp_rows": int((lp[1] > 0).sum()), "legs": highs}
    t_trio = sum(r[1] for r in res)
    t_cov = sum(r[2] for r in res)
    t_lp = sum(r[3] for r in res)
    return dict(value=n / dt / 1e6, unit="Mreads/s", cores=cores, kind="port",
                solver="oracle's exact active-set LAD (oracle/pantax_oracle.c, same optimum as HiGHS: tests/golden/lp_cases.npz); "
                       "HiGHS itself timed in `highs` on row samples of one species' LP",
                sample=("the whole workload: %d reads, %d species, one species per worker on %d cores (fork pool); "
                        "binning in %d read slices" % (n, S, cores, cores)) if n == rd.n_reads else
                       ("the first %d of the %d reads over ALL %d species (the per-species index build does not shrink with the reads: "
                        "Mreads/s of the sample is a lower bound of the CPU's rate on the whole workload), one species per worker on %d "
                        "cores (fork pool); binning in %d read slices" % (n, rd.n_reads, S, cores, cores)),
                seconds=dt,
                phases_wall_s={"binning+species_profile": t_bin, "group_reads_by_species": t_group,
                               "per-species (trio index, coverage, filters, 2 LP solves)": dt - t_bin - t_group},
                phases_cpu_s_summed_over_workers={"trio_index": t_trio, "node_coverage (incl. read selection)": t_cov,
                                                  "filters+LP solves": t_lp},
                highs=highs)


def pmc_traffic(kernel, wl, corrected=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, separate
    passes, KB -> bytes; profiles/r02_pmc_<workload>.json, collected by tools/pmc_step.sh on the same workload).  None
    when no file matches this workload.  corrected: 2 x FETCH + WRITE -- the guide's gfx950 correction (FETCH_SIZE tallies
    128-byte requests at 64 B for wide coalesced reads) applied to the whole fetch, i.e. an upper bound; calibration on this
    path's own kernels: WRITE_SIZE is exact, FETCH_SIZE reads 0.5x on coalesced streams and 1.0x on the
    node-block kernel's scattered 4-byte loads."""
    try:
        for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
            if not (fn.startswith("r02_pmc_") and fn.endswith(".json")):
                continue
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
            w = d["workload"]
            if (w["reads"], w["species"], w["haps"], w["genome_len"], w.get("seed")) != (wl["reads"], wl["species"], wl["haps"], wl["genome_len"], wl["seed"]):
                continue
            k = d["kernels"].get(kernel)
            if k:
                return k["hbm_bytes_fetch_x2"] if corrected else k["hbm_bytes_per_launch"]
    except Exception:   # noqa: BLE001
        pass
    return None


def pao_hard_cpu(synth, seed, n_species):
    """A benchmark LP that is not trivial (profile.rs:1428-1451 at the shape BASELINE.md section 2 flags): every strain of
    a species is present (10 columns after the first filter, >= 30 membership patterns, ~3e5 rows per species).  CPU side:
    the set, species 0 through the oracle (its objective is compared with the device's), and that species' LP for the
    HiGHS legs."""
    from oracle import oracle as orc
    from tests.helpers import select_reads
    sset = synth.make_set(seed, n_species, 10, 1_000_000 * n_species, 5_000_000, present_frac=1.0)
    g = sset.species[0]
    rd = sset.reads
    sp = orc.bin_reads(rd.step_off, rd.node_id, [x.range_start for x in sset.species], [x.range_end for x in sset.species])
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
    T = orc.TrioTable(G)
    so, nid, ps, pe = select_reads(rd, np.nonzero(sp == 0)[0])
    b, c, tb, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
    t0 = time.perf_counter()
    rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb)
    t_orc = time.perf_counter() - t0
    lp = _species_lp(orc, g, G, b, c, orc.metrics_to_dicts(met))
    info = dict(workload="%d species x 10 strains, ALL strains present (present_frac 1.0), %d reads, seed %d" % (n_species, rd.n_reads, seed),
                oracle_optimize_species_s_species0=t_orc, oracle_obj1_species0=o1, n_candidates_oracle_species0=nc)
    return sset, info, lp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS) + ["custom"])
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("--species", type=int, default=None)
    ap.add_argument("--haps", type=int, default=None)
    ap.add_argument("--genome-len", type=int, default=None)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-cores", type=int, default=0, help="workers of the CPU baseline (0 = all host cores)")
    ap.add_argument("--highs-rows", default="2000,5000,10000", help="row samples of the HiGHS legs ('' = none)")
    ap.add_argument("--highs-time-limit", type=float, default=25.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gaf", action="store_true", help="skip the from-GAF-text measurement")
    ap.add_argument("--no-hard", action="store_true", help="skip the pao_hard leg")
    ap.add_argument("--hard-species", type=int, default=8)
    args = ap.parse_args()

    base = WORKLOADS.get(args.workload, WORKLOADS["cfg3"])
    label, seed_off = base[0], base[1]
    n_species = args.species if args.species is not None else base[2]
    n_haps = args.haps if args.haps is not None else base[3]
    n_reads = args.reads if args.reads is not None else base[4]
    genome_len = args.genome_len if args.genome_len is not None else base[5]
    if (n_species, n_haps, n_reads, genome_len) != tuple(base[2:6]):
        label = "custom"
    highs_sizes = [int(x) for x in args.highs_rows.split(",") if x.strip()]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from pantax_amd import synth
    from pantax_amd.pipeline import LocalComm, StepConfig, TorchComm, partition_species, profile_step, profile_steps_pipelined
    cfg = StepConfig()

    # deterministic synthetic input (seed = 20260501 + cfg index): weak scaling = one such set per rank
    # (+ 1000 x rank), strong scaling = ONE set, species packed onto the ranks by weight, reads follow their species
    t_gen = time.perf_counter()
    strong = args.scaling == "strong" and world > 1
    full = None
    if not strong:
        seed = 20260501 + seed_off + 1000 * rank
        sset = synth.make_set(seed, n_species, n_haps, n_reads, genome_len)
        for i, g in enumerate(sset.species):
            g.name = "%d" % (100000 * rank + 1000 + i)
        total_reads = n_reads * world
    else:
        seed = 20260501 + seed_off
        full = synth.make_set(seed, n_species, n_haps, n_reads, genome_len)   # every rank generates the same set; it keeps a slice of the reads
        sset = None
        total_reads = n_reads
    gen_s = time.perf_counter() - t_gen
    wl = dict(reads=n_reads, species=n_species, haps=n_haps, genome_len=genome_len, seed=seed)

    # ---- CPU baseline first: forked workers, before anything initialises the GPU in this process (rank 0, N = 1 only)
    cpu = None
    cores = args.cpu_cores or (os.cpu_count() or 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sset, cfg, cores, highs_sizes, args.highs_time_limit, n_limit=CPU_SAMPLE_READS)
    hard = None
    if rank == 0 and world == 1 and not args.no_hard:
        import multiprocessing as mp
        hard_set, hard, hard_lp = pao_hard_cpu(synth, 20260601, args.hard_species)
        hard_names = [g.name for g in hard_set.species]
        hard_haps = [h for g in hard_set.species for h in g.hap_names]
        if hard_lp is not None and highs_sizes:
            with mp.get_context("fork").Pool(min(cores, len(highs_sizes))) as pool:
                hard["highs"] = highs_legs(pool, hard_lp, highs_sizes, args.highs_time_limit)
            hard["lp_rows_species0"] = int((hard_lp[1] > 0).sum())
            hard["lp_columns_species0"] = hard_lp[2]

    import torch
    from pantax_amd.engine import Engine
    # PANTAX_BENCH_BACKEND=gloo: dry run of the N > 1 flow on a box with fewer GPUs than ranks (ranks share devices, the
    # exchange goes over gloo); the driver's runs use the default, RCCL with one GPU per rank
    backend = os.environ.get("PANTAX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    comm = LocalComm()
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            comm = TorchComm(device=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            comm = TorchComm(device=None)
    eng = Engine(local_rank)
    t_up = time.perf_counter()
    ingest_route = None
    if strong:
        # this rank's slice of the reads (file order) is binned against ALL species ranges, the species are packed
        # onto the ranks by the summed counts, and the packed records travel to their owners -- once, before the timed steps
        from pantax_amd.pipeline import route_reads
        rd = full.reads
        a, b = n_reads * rank // world, n_reads * (rank + 1) // world
        so = rd.step_off.astype(np.int64)
        mapq = np.where((rd.mapq < 0) | (rd.mapq > 254), 255, rd.mapq)
        eng.upload_ranges([g.range_start for g in full.species], [g.range_end for g in full.species])
        eng.upload_reads(so[a:b + 1] - so[a], rd.node_id[so[a]:so[b]], rd.pstart[a:b], rd.pend[a:b], rd.qlen[a:b], mapq[a:b])
        eng.sync()
        t_r = time.perf_counter()
        _, rc_loc, *_ = eng.rcls_profile(want_species=False)
        rc_all = comm.allreduce_sum(rc_loc)
        owner = partition_species([8.0 * float(rc_all[i]) + g.n_nodes for i, g in enumerate(full.species)], world)
        rstats = route_reads(eng, owner, comm)
        eng.sync()
        ingest_route = dict(ms=(time.perf_counter() - t_r) * 1e3, **rstats,
                            what="bin the slice against all ranges + one all-reduce of the counts + pack + all-to-all(v) + rebuild resident reads")
        sset = synth.SyntheticSet([g for i, g in enumerate(full.species) if owner[i] == rank], None)
        del full
        eng.upload_db(sset.species)
    else:
        eng.upload_db(sset.species)
        eng.upload_packed(sset.reads)          # inputs resident in HBM before the timed region
    eng.sync()
    upload_ms = (time.perf_counter() - t_up) * 1e3   # host->device of packed reads + graph (pageable memory, incl. numpy packing)
    species_names = [g.name for g in sset.species]
    hap_names = [hn for g in sset.species for hn in g.hap_names]
    avg_len = sset.avg_len()
    S_loc = len(sset.species)
    S_max, H_max = S_loc, len(hap_names)
    if strong:
        import torch.distributed as dist
        t = torch.tensor([S_loc, len(hap_names)], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        S_max, H_max = int(t[0].item()), int(t[1].item())

    # one-time set-up of the step path (not the W warm-up steps of the contract, which follow): the first stream of steps that
    # keeps one step enqueued ahead makes the HIP runtime grow its pools (a ~6 ms stall at the third enqueue, measured); it
    # happens here, with the uploads, not inside the timed region
    t_pr = time.perf_counter()
    profile_steps_pipelined(eng, species_names, hap_names, avg_len, 3, cfg, comm, shard_max=S_max, rows_max=H_max)
    eng.sync()
    prime_ms = (time.perf_counter() - t_pr) * 1e3

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    # Warm-up steps bracket EVERY launch with HIP events (per-kernel table, dominant kernel); the timed steps
    # bracket only that dominant kernel: ~200 event records per step cost ~0.15 ms, which is not part of the path.
    out = None
    eng.timing_enable(True)
    eng.timing_reset()
    n_warm_timed = 0
    if args.warmup:
        # the very first step also allocates: its launches are not representative
        out = profile_step(eng, species_names, hap_names, avg_len, cfg, comm, shard_max=S_max, rows_max=H_max)
        n_warm_timed = 1
    if args.warmup > 1:
        # the remaining warm-up steps go through the path that is timed below (one step enqueued ahead): whatever the runtime
        # sets up the first time two steps are in flight happens here, not in the timed region
        eng.timing_reset()
        n_warm_timed = args.warmup - 1
        out = profile_steps_pipelined(eng, species_names, hap_names, avg_len, n_warm_timed, cfg, comm, shard_max=S_max, rows_max=H_max)[-1]
    warm = eng.timing_get() if args.warmup else {}
    # the two largest of the warm-up table are bracketed in the timed steps; the dominant kernel is the one with the larger
    # average THERE (at cfg3 the node-block index kernel and the coverage kernel are within a few per cent of each other, and
    # the warm-up figures carry the cost of the other ~60 event pairs)
    top2 = [k for k, _ in sorted(warm.items(), key=lambda kv: -kv[1][1])[:2]] if warm else ["coverage_step_kernel"]
    # ... and the coverage kernel always (the round-1 review's named kernel; since the next step's index rebuild runs beside the
    # previous step's tail the warm-up table ranks the index kernels above it)
    eng.timing_filter("|".join(top2 + [k for k in ["coverage_step_kernel"] if k not in top2]))
    eng.timing_reset()
    # host hygiene before the timed region: with torch imported the interpreter holds ~1e6 long-lived objects, and a full
    # collection of the cyclic garbage collector (triggered by the tables' tuples every few dozen steps) stops the thread that
    # enqueues the next step for 30-50 ms; frozen objects are not scanned again
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    # K steps back to back; with N > 1 the all-reduce of step i is in flight while step i+1 computes (every step's tables
    # are complete before the closing barrier)
    t0 = time.perf_counter()
    out = profile_steps_pipelined(eng, species_names, hap_names, avg_len, args.steps, cfg, comm, shard_max=S_max, rows_max=H_max)[-1]
    barrier()
    dt = time.perf_counter() - t0
    timings = eng.timing_get()
    eng.timing_enable(False)
    eng.timing_filter(None)
    # extra (not `value`): the same step when the unique-trio index, which depends on the DB only, stays
    # resident between steps instead of being rebuilt like the reference does on every run
    cfg_cached = StepConfig(rebuild_trio=False)
    profile_step(eng, species_names, hap_names, avg_len, cfg_cached, comm, shard_max=S_max, rows_max=H_max)
    barrier()
    t1 = time.perf_counter()
    profile_steps_pipelined(eng, species_names, hap_names, avg_len, args.steps, cfg_cached, comm, shard_max=S_max, rows_max=H_max)
    barrier()
    dt_cached = time.perf_counter() - t1
    # first-class extra: the same workload from GAF TEXT on disk -- pread + PCIe + device tokenizer (a1) -> resident reads -> one
    # step -> tables.  Never `value` (the contract's value has its inputs resident in HBM).
    gaf_extra = None
    if rank == 0 and world == 1 and not args.no_gaf:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            gp = os.path.join(td, "reads.gaf")
            t_w = time.perf_counter()
            n_gaf = min(n_reads, GAF_SAMPLE_READS)
            synth.write_gaf(synth.head_reads(sset.reads, n_gaf), gp)
            write_s = time.perf_counter() - t_w
            eng.load_reads_from_gaf(gp)                      # warm (allocations, page cache)
            eng.sync()
            t2 = time.perf_counter()
            eng.load_reads_from_gaf(gp)
            eng.sync()
            t_load = time.perf_counter() - t2
            out_gaf = profile_step(eng, species_names, hap_names, avg_len, cfg, LocalComm(), shard_max=S_max, rows_max=H_max)
            eng.sync()
            t_e2e = time.perf_counter() - t2
            same = (out is not None and out_gaf[0] == out[0] and out_gaf[1] == out[1]) if n_gaf == n_reads else None
            gaf_extra = {"gaf_bytes": os.path.getsize(gp), "reads": n_gaf, "tokenize_to_resident_ms": t_load * 1e3, "end_to_end_ms": t_e2e * 1e3,
                         "end_to_end_mreads_per_s": n_gaf / t_e2e / 1e6, "gaf_gb_per_s": os.path.getsize(gp) / t_load / 1e9,
                         "tables_equal_to_packed_input_run": same, "gaf_written_in_s": write_s,
                         "note": None if n_gaf == n_reads else "the first %d reads of the workload as GAF text, against the whole resident db" % n_gaf}
    # extra: the non-trivial LP (pao_hard), timed with every launch bracketed
    if hard is not None:
        eng_h = Engine(local_rank)
        eng_h.upload_db(hard_set.species)
        eng_h.upload_packed(hard_set.reads)
        h_avg = hard_set.avg_len()
        profile_step(eng_h, hard_names, hard_haps, h_avg, cfg)
        eng_h.timing_enable(True)
        eng_h.timing_reset()
        n_h = 3
        t3 = time.perf_counter()
        for _ in range(n_h):
            out_h = profile_step(eng_h, hard_names, hard_haps, h_avg, cfg)
        eng_h.sync()
        t_h = (time.perf_counter() - t3) / n_h
        kt_h = eng_h.timing_get()
        eng_h.timing_enable(False)
        st_h = out_h[2]
        lad_ms = sum(v[1] for k, v in kt_h.items() if k.startswith("lad_")) / n_h
        hard.update(reads=hard_set.reads.n_reads, ms_per_step_all_launches_bracketed=t_h * 1e3,
                    lad_kernels_ms_per_step=lad_ms, lad_kernels_ms_per_species=lad_ms / max(len(hard_names), 1),
                    n_candidates=st_h["n_cand"][:8],
                    n_rows=st_h["n_rows"][:8], n_patterns=st_h["n_patterns"][:8], iters=st_h["iters"][:8],
                    objective=st_h["obj"][:4], gpu_obj1_species0=st_h["obj"][0][0],
                    kernels_ms_per_step={k: v[1] / n_h for k, v in sorted(kt_h.items(), key=lambda kv: -kv[1][1])[:8]})
        eng_h.close()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        species_rows, strain_rows, stats = out
        ms_per_step = dt / args.steps * 1e3
        value = total_reads / (dt / args.steps) / 1e6
        n_lp_rows = int(sum(stats["n_rows"]))
        n_unique = int(eng.trio_nodes_info(fetch=False))      # after the timed region: only its size is wanted
        ab, dims = algorithmic_bytes(sset, n_lp_rows, n_unique, eng.R, eng.T)
        dims["U"] = n_unique
        # dominant kernel by HIP-event time on the library's stream
        roofline = None
        dom = max((k for k in top2 if k in timings), key=lambda k: timings[k][1] / max(timings[k][0], 1), default=None)
        if dom and dom in timings:
            launches, tot_ms = timings[dom]
            avg_ms = tot_ms / max(launches, 1)
            per = ab.get(dom)
            if per is not None:
                ach = per / (avg_ms * 1e-3) / 1e9
                roofline = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                traffic=pmc_traffic(dom, wl), traffic_fetch_x2=pmc_traffic(dom, wl, True), avg_ms=avg_ms, launches_timed=launches,
                                algorithmic_bytes=per)
            else:
                roofline = dict(bound="hbm", kernel=dom, achieved=0.0, peak=HBM_PEAK_GBS, unit="GB/s", frac=0.0, traffic=None,
                                avg_ms=avg_ms, algorithmic_bytes=0,
                                note="no streaming-traffic model for this launch (latency-bound: the small-LP solver does "
                                     "O(#patterns*log n) searches per pivot)")
        # the runner-up of the timed steps, same ruler (the two are within a few per cent of each other at cfg3)
        if roofline is not None:
            roofline["note"] = ("HIP-event durations of kernels that share the device: in a stream of steps the index rebuild of step i+1 "
                                "(trio_block / trio_lookup, side stream, LOW priority) runs beside the tail of step i and is stretched by it; "
                                "stand-alone times: DESIGN.md section 4")
            for k2 in top2:
                if k2 != dom and k2 in timings and k2 in ab:
                    l2, t2 = timings[k2]
                    a2 = t2 / max(l2, 1)
                    roofline["runner_up"] = dict(kernel=k2, avg_ms=a2, algorithmic_bytes=ab[k2], frac=ab[k2] / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 traffic=pmc_traffic(k2, wl), traffic_fetch_x2=pmc_traffic(k2, wl, True))
            if "coverage_step_kernel" in timings and "coverage_step_kernel" in ab and roofline["kernel"] != "coverage_step_kernel" \
                    and roofline.get("runner_up", {}).get("kernel") != "coverage_step_kernel":
                l3, t3 = timings["coverage_step_kernel"]
                a3 = t3 / max(l3, 1)
                roofline["coverage_step_kernel"] = dict(avg_ms=a3, algorithmic_bytes=ab["coverage_step_kernel"], launches_timed=l3,
                                                        frac=ab["coverage_step_kernel"] / (a3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                        traffic=pmc_traffic("coverage_step_kernel", wl),
                                                        traffic_fetch_x2=pmc_traffic("coverage_step_kernel", wl, True))
        # the other large kernels against the same ruler (warm-up table; one launch per step each unless noted)
        others = {}
        for k, (launches, tot_ms) in warm.items():
            if k in ab and launches:
                a_ms = tot_ms / launches
                others[k] = dict(avg_ms=a_ms, algorithmic_bytes=ab[k], frac=ab[k] / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        line = {
            "metric": "PAO wall-time (s) + Mreads/s GAF->abundance (packed reads resident in HBM)",
            "value": value, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "pao_wall_s": ms_per_step / 1e3,
            "from_gaf_text_mreads_per_s": gaf_extra["end_to_end_mreads_per_s"] if gaf_extra else None,
            "upload_ms_once": upload_ms, "step_path_primed_ms_once": prime_ms, "synthetic_set_generated_in_s": gen_s, "ingest_route": ingest_route,
            "ms_per_step_trio_index_resident": dt_cached / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": "%s: %s -- %d species x %d strains, %d short reads (150 bp), genome %d bp %s; this rank: V=%d nodes, "
                                   "P=%d path steps, T=%d walk steps, seed %d"
                                   % (args.workload if label != "custom" else "custom", label, n_species, n_haps, n_reads, genome_len,
                                      "per GPU" if args.scaling == "weak" else "in all (cut over the ranks)", dims["V"], dims["P"], dims["T"], seed),
                       "species_per_gpu": S_loc, "strains_total": n_species * n_haps * (world if args.scaling == "weak" else 1),
                       "reads_total": total_reads, "parallelism": "species-shard x%d" % world, "sample_nodes": 0,
                       "exchange": "none" if world == 1 else ("one rccl all_reduce per step, in flight during the next step" if backend == "nccl" else backend + " all_reduce (dry run)")},
            "from_gaf_text": gaf_extra,
            "roofline": roofline,
            "roofline_other_kernels": others,
            "kernels_ms_per_step": {k: v[1] / max(n_warm_timed, 1) for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])},
            # timer scopes, not dispatches: a scope brackets one stage (a sort = several launches).  Dispatches per step, counted in
            # the rocprofv3 trace of this path: 43 at cfg3 (profiles/r02_final_cfg3_timeline.txt), fills and copies included
            "kernel_timer_scopes_per_step": int(sum(v[0] for v in warm.values()) / max(n_warm_timed, 1)),
            "kernels_ms_per_step_source": "warm-up steps (every launch bracketed by HIP events); the timed steps bracket roofline.kernel only",
            "solver": {"iters": stats["iters"][:4], "n_rows": stats["n_rows"][:4], "n_patterns": stats["n_patterns"][:4],
                       "objective": stats["obj"][:4], "lp_rows_total": n_lp_rows},
            "pao_hard": hard,
            "result": {"n_species_rows": len(species_rows), "n_strain_rows": len(strain_rows),
                       "top_strains": [(r[0], r[1], round(r[2], 4), round(r[3], 6)) for r in strain_rows[:3]]},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
