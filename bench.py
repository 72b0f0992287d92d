#!/usr/bin/env python3
"""bench.py -- PAO wall time / Mreads/s of the profiling hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM: read
binning + species counters, species profile, unique-trio index (rebuilt per step like the reference does per run,
profile.rs:2936), node-coverage histogram, LP row grouping, the two PAO solves, filters and the abundance table.

Workload (default, every N) = BASELINE.json configs[3], the configuration the metric is quoted on ("at 10k strains"):
1k species / 10k strains, 100M short reads (10 strains per species, 5 Mbp genomes, 150 bp reads, seed
20260501 + 5).  It fits ONE MI355X (V = 3.2e8 nodes, P = 2.2e9 path steps, T = 7.6e8 walk steps, ~100 GB of the 288 GB),
so N = 1 runs all of it and N > 1 cuts the SAME set over the ranks (`scaling: strong`, the curve of one workload): every
rank generates only its 1/N slice of the reads (tools/native/synth_set.c: every species and every chunk of reads is a pure
function of the seed), bins it against all species ranges, the species are packed onto the ranks by weight (longest
processing time first, pipeline.partition_species) and the packed records travel to the owner of their species in one RCCL
all-to-all(v) over xGMI (pipeline.route_reads) -- once, before the timed steps (`ingest_route`); one RCCL all-reduce per step
carries the normalisers.  `--scaling weak` gives every rank its own set of the given size instead.
`--workload cfg3` = configs[2] (100 species, 10M reads), `cfg2` = configs[1] (1 species, 1M reads).

    python bench.py                                   # cfg4, 1 GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Beside `value` (inputs resident in HBM) the line carries `roofline` (dominant kernel, HIP events on the library's stream),
`from_gaf_text` (the same workload from GAF text on disk -> tables, PCIe + device tokenizer included), `pao_hard` (a
synthetic variant whose first filter keeps all ten strains: the LP regime BASELINE.md section 2 flags) and
`cpu_baseline`: the plain-C oracle on all host cores, one species per worker thread like profile.rs:3297-3319, on a bounded
sample (the first quarter of the reads over ALL species), plus SciPy-HiGHS -- the reference's open solver -- on row samples of
one species' LP and on its FULL LP under a stated time limit.  The CPU legs run in a CHILD process that never touches the
GPU (started before this process initialises it); the parent waits for the oracle leg (all cores), then goes on while the
child finishes the single-threaded HiGHS legs; the GPU line is printed even when the child fails (`cpu_baseline.error`).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
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
    "cfg4": ("configs[3]: 1k species / 10k strains, 100M short reads", 5, 1000, 10, 100_000_000, 5_000_000),
    # configs[4] (HiFi gut mock, 50k strains, 8 GPUs) per-GPU share: 125 species x 50 strains, long reads N(15000, 3000^2) with the bases of
    # cfg4's share (the long-read kernels: coverage_step_kernel, walk_sum_kernel; 50 candidate columns per species)
    "cfg5_share": ("configs[4] per-GPU share: 125 species / 6250 strains, 125k HiFi-shaped reads", 6, 125, 50, 125_000, 5_000_000),
    # configs[4] at its size on ONE GPU: 1 000 species x 50 strains = 1.1e10 path steps, more than one resident db addresses -> the species are
    # cut into several dbs that share the GPU (run_many_dbs)
    "cfg5": ("configs[4]: 1k species / 50k strains, 1M HiFi-shaped reads", 7, 1000, 50, 1_000_000, 5_000_000),
    # round 6: the shape of the database PanTax ships -- 8 778 species with the strains-per-species histogram of the reference's genomes_info.txt
    # (7 465 single-genome species as 1024-bp chunk graphs, build_eq1.rs:26-36; the rest 2 .. 10 strains), 5e7 short reads (synthdata.RefDbSet)
    "refdb": ("reference-DB shape: 8778 species / 13404 strains (7465 single-strain chunk graphs), 50M short reads", 7, 8778, 10, 50_000_000, 5_000_000),
}
LONG_READ_WORKLOADS = ("cfg5_share", "cfg5")
DEFAULT_WORKLOAD = "cfg4"
CPU_SAMPLE_READS = 25_000_000    # the CPU baseline runs on the first chunks of a larger workload (a bounded sample, ~10-30 s of all cores)
GENERATOR = "native-v1"          # tools/native/synth_set.c; part of the workload key of the committed PMC files


def workload_spec(name, species=None, haps=None, reads=None, genome_len=None):
    base = WORKLOADS[name]
    S = species if species is not None else base[2]
    H = haps if haps is not None else base[3]
    R = reads if reads is not None else base[4]
    L = genome_len if genome_len is not None else base[5]
    if name == "refdb":     # --species N scales every count of the histogram (tests, quick runs); --reads as given
        return dict(name="refdb", label=base[0], seed=20260501 + base[1], species=S, haps=H, reads=R, genome_len=L, long_reads=False, refdb=True,
                    scale=S / float(base[2]))
    custom = (S, H, R, L) != tuple(base[2:6])
    return dict(name="custom" if custom else name, label="custom" if custom else base[0], seed=20260501 + base[1], species=S, haps=H, reads=R, genome_len=L,
                long_reads=name in LONG_READ_WORKLOADS)


def native_set(spec, threads=None, seed_shift=0):
    import synthdata as synth
    if spec["name"] == "refdb" or spec.get("refdb"):
        return synth.RefDbSet(spec["seed"] + seed_shift, spec["reads"], spec["genome_len"], scale=spec.get("scale", 1.0), threads=threads)
    return synth.NativeSet(spec["seed"] + seed_shift, spec["species"], spec["haps"], spec["reads"], spec["genome_len"], long_reads=spec.get("long_reads", False),
                           threads=threads)


def workload_key(spec):
    k = dict(reads=spec["reads"], species=spec["species"], haps=spec["haps"], genome_len=spec["genome_len"], seed=spec["seed"], generator=GENERATOR)
    if spec.get("long_reads"):
        k["long_reads"] = True
    return k


def algorithmic_bytes(species, n_lp_rows, U, R, T, kept=None, with_columns=None):
    """SURVEY.md section 8d per-stage compulsory traffic for ONE step of this rank's workload (U = unique trios, R reads,
    T walk steps resident on this rank).  kept: names of the species the species level kept (None: all) -- the statistics and histogram passes
    of the resident step do not read the others (round 6): their rulers count the kept species' nodes, plus the 12 bytes of zeros a dropped node is given;
    with_columns: names of the species with at least one LP column (the histogram pass of the row sort reads those alone)."""
    V = sum(g.n_nodes for g in species)
    L = int(sum(int(g.node_len.sum()) for g in species))
    V_k = V if kept is None else sum(g.n_nodes for g in species if g.name in kept)
    L_k = L if kept is None else int(sum(int(g.node_len.sum()) for g in species if g.name in kept))
    V_c = V_k if with_columns is None else sum(g.n_nodes for g in species if g.name in with_columns)
    P = int(sum(int(g.path_off[-1]) for g in species))
    H = sum(g.n_paths for g in species)
    Wr = max(T - 2 * R, 0)
    cov = 4 * T + 12 * R + 4 * V + 8 * V + L // 8 + 12 * Wr
    win = max(P - 2 * H, 0)
    return {
        # keys = the kernel names rocprofv3 shows (the library's timer labels are those names)
        # a2: 4T + 4R(offsets) + 4R(out)
        "bin_slots_kernel": 4 * T + 4 * R + 4 * R,
        "bin_reads_kernel": 4 * T + 4 * R + 4 * R,
        # a8 minus the popcount pass: 4T + 12R + 4V(node_len) + 8V(bases) + L/8(bitmap) + 12*Wr(trio probes); the short-read kernel and
        # the general one (groups that hold steps of longer walks) share the ruler
        "coverage_fast_kernel": cov,
        "coverage_step_kernel": cov,
        "coverage_long_kernel": cov,     # round 6: the select-only instantiation for the groups of longer walks
        # the coverage arena zeroed in front of every pass: bases 8V + trio_bases 8U + bit vector L/8 + full-node flags V/8, written once
        "zero_fill_kernel": 8 * V + 8 * U + L // 8 + V // 8,
        # the walk sums of long reads: step code 1 + node id 4 + one 4-byte node length per step, 8 bytes per long walk out
        "walk_sum_kernel": 9 * T + 8 * R,
        # a8 popcount: L/8 bitmap in + 8V cov out
        "popcount_kernel": L // 8 + 8 * V,
        # the resident step (round 4): popcount folded into the node statistics pass -- lengths 4V + bases 8V in, counts 4V + abundances 8V out, bitmap L/8
        "node_cov_stats_kernel": 24 * V_k + L_k // 8 + 12 * (V - V_k),
        # a7 (round 5: bytes the launch MUST move, not a split of SURVEY 8d's whole-index figure, which counts 12-byte keys this design never moves).
        # trio_visit_kernel: the visit table (4 bytes per visit slot, ~64/60 of the interior positions: pads) + every walk entry once (4P) +
        # per group of 64 visits the head mask, node base and ballot (20 B) + one 16-byte record per unique window.
        # trio_rows_kernel: the records (16U) + per group ballot, first row, species (16 B) + three node lengths per row (12U) + the rows it
        # files: entry 8 + length 4 + owner 2 bytes (14U); the 16-byte read-modify-write of the heads' node records is left out (the number
        # of nodes that head rows is not known to the harness: ~0.14 U at ten strains per species).
        "trio_visit_kernel": 4 * (P * 64 // 60) + 4 * P + 20 * (P // 60) + 16 * U,
        "trio_rows_kernel": 16 * U + 16 * (P // 60) + 12 * U + 14 * U,
        # trio_file_kernel (every build of a db but its first: decision and filing in one pass): the visit table (4 bytes per visit slot; first-fit
        # packing: ~1 % pads) + every walk entry once (4P) + per group head mask, node base, species, first row (24 B) + three node lengths per
        # row (12U) + the rows: entry 8 + length 4 + owner 2 bytes (14U).  The heads' 16-byte node records (read, rewritten only where they differ)
        # are left out as above.
        "trio_file_kernel": 4 * (P * 64 // 63) + 4 * P + 24 * (P // 63) + 12 * U + 14 * U,
        # the node-block / bucket / pass-over-the-walks kernels (species the visit table does not cover; forced paths): SURVEY 8d's halves
        "trio_block_kernel": 4 * P + 12 * win,
        "trio_lookup_kernel": 12 * win + 12 * U,
        "trio_fill_kernel": 4 * P + 16 * win,
        "trio_count_kernel": 4 * P + 4 * V,
        "trio_uniq_kernel": 16 * win,
        # a9: per-haplotype statistics by key.  Pass 0 reads the abundance of EVERY row (8U) and compacts the non-zero ones; length, owner and the two later
        # passes touch those only -- how many is not known to the harness, so the ruler is the compulsory 8U (rounds 4-5 counted 3 x 14 U: three full passes)
        "hap_rows_pass_kernel": 8 * U,
        # a10: the membership masks by node (8V node -> haplotypes in, 8V masks out) / by walk (4P in + 8V out)
        "mask_nodes_kernel": 24 * V,      # ... + 8V of counts and lengths since the path_cov_ratio sums ride on this pass (round 4; 16V before)
        "mask_kernel": 4 * P + 8 * V,
        # a12's row compaction: abundance + mask of every node in (16V), the valid rows out (16 n)
        "scan_chained_kernel<Row>": 16 * V + 16 * n_lp_rows,
        # a12's rows sorted straight from the node arrays (sample_sort_nodes.hip, round 4): the histogram pass reads abundance + mask of every
        # node (16V; it also stages the rows that have to travel, a data-dependent third of them: not counted); the scatter and the tie
        # fills write every row once between them (16 n)
        "ssn_hist_kernel": 24 * V_c,     # (the masks are formed in this pass since the end of round 4: haplotype words 8V + abundance 8V + covered bases 4V + lengths 4V)
        # the LP objective summed over the sorted rows (8 n) instead of over abundance + mask of every node
        "objective_rows_kernel": 8 * n_lp_rows,
        "sort_hist_kernel": 8 * n_lp_rows,
        "sort_scatter_kernel": 2 * 24 * n_lp_rows,
    }, dict(R=R, T=T, V=V, L=L, P=P, H=H)


# ------------------------------------------------------------------------------------------------ CPU legs (child process)
def _mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 1048576.0
    except OSError:
        pass
    return None


def _cgroup_limit_gb():
    for fn in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            t = open(fn).read().strip()
            return None if t == "max" else int(t) / 2**30
        except (OSError, ValueError):
            continue
    return None


def _species_lp(orc, G, b, c, met):
    """The LP optimize_species solved for this species: candidate columns = haplotypes with a first_sol."""
    cand = np.array([i for i, m in enumerate(met) if m["first_sol"] is not None], dtype=np.uint32)
    if len(cand) == 0:
        return None
    mask, _ = orc.path_masks(G, cand, c)
    return mask, b / G.node_len, len(cand)


def _highs_leg(lp, rows, tlimit):
    """SciPy's bundled HiGHS on the first `rows` (random order, seed 0; None = ALL) valid rows of one species' LP (the
    reference's open backend, highs_opt profile.rs:2689-2882: x in [0, 1.05 max a], y_v >= +-(A x - a)_v, min (1/n) sum y),
    next to the oracle's exact LAD on the same rows."""
    from oracle import oracle as orc
    from scipy import sparse
    from scipy.optimize import linprog
    import scipy
    mask, ab, p = lp
    valid = np.nonzero((ab > 0))[0]
    ub = 1.05 * float(ab.max())
    if rows is not None and rows < len(valid):
        take = np.sort(valid[np.random.default_rng(0).permutation(len(valid))[:rows]])
    else:
        take = valid
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
    return dict(rows=n, columns=p, patterns=int(len(np.unique(m))), highs_seconds=dt, highs_status=int(r.status), highs_message=str(r.message)[:120],
                highs_objective=(float(r.fun) if r.status == 0 else None), exact_lad_objective=float(obj), exact_lad_seconds=dt_lad,
                time_limit_s=tlimit, finished=bool(r.status == 0), scipy=scipy.__version__)


def _oracle_lp_of_species(orc, ns, sset_species, rd, sp, first, order, s, cfgd):
    from tests.helpers import select_reads
    g = sset_species[s]
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
    T = orc.TrioTable(G)
    sel = np.sort(order[int(first[s]):int(first[s + 1])]).astype(np.int64)
    so, nid, ps, pe = select_reads(rd, sel)
    b, c, tb, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
    t0 = time.perf_counter()
    rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, **cfgd)
    md = orc.metrics_to_dicts(met)
    return _species_lp(orc, G, b, c, md), (time.perf_counter() - t0, nc, o1, [m["first_sol"] for m in md])


def cpu_leg_child(args):
    """Runs in a child process that never initialises the GPU.  Writes its result file twice: after the oracle leg
    (stage "oracle_done": the parent goes on) and at the end (stage "done")."""
    out_path = args.cpu_leg_child
    res = {"stage": "start", "mem_available_gb_start": _mem_available_gb(), "cgroup_memory_limit_gb": _cgroup_limit_gb()}

    def publish():
        tmp = out_path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(res, f)
        os.replace(tmp, out_path)
    try:
        from oracle import oracle as orc
        import synthdata as synth
        from pantax_amd.pipeline import StepConfig
        spec = workload_spec(args.workload, args.species, args.haps, args.reads, args.genome_len)
        cfg = StepConfig(fr=0.5) if spec.get("long_reads") else StepConfig()
        cfgd = dict(fr=cfg.fr, fc=cfg.fc, sr=cfg.sr)
        cores = args.cpu_cores or (os.cpu_count() or 1)
        t_g = time.perf_counter()
        ns = native_set(spec, threads=min(cores, 64))
        n_chunks = synth.N_CHUNKS if spec["reads"] <= CPU_SAMPLE_READS else max(1, (synth.N_CHUNKS * CPU_SAMPLE_READS) // spec["reads"])
        sset = ns.make(0, n_chunks)
        rd = sset.reads
        n, S = rd.n_reads, spec["species"]
        res["sample_generated_in_s"] = time.perf_counter() - t_g
        res["mem_available_gb_after_generation"] = _mem_available_gb()
        # ---- the oracle on all cores: binning in slices, then one species per worker thread
        t0 = time.perf_counter()
        sp = orc.par_bin_reads(rd.step_off, rd.node_id, ns.range_start, ns.range_end, cores)
        counts = orc.species_counts(sp, rd.qlen, rd.mapq, S)
        keep, absolute, _ = orc.species_profile(sp, rd.qlen, counts, ns.avg_len())
        t_bin = time.perf_counter() - t0
        first, order = orc.group_reads(sp, S)
        t_group = time.perf_counter() - t0 - t_bin
        Gs = [orc.Graph(g.node_len, g.path_off, g.path_nodes) for g in sset.species]
        cnt = np.diff(first.astype(np.int64))
        todo = np.argsort(-cnt, kind="stable")                      # heaviest first
        pr = orc.par_profile_species(Gs, ns.range_start, rd.step_off, rd.node_id, rd.pstart, rd.pend, first, order, keep, absolute, todo, cores, **cfgd)
        dt = time.perf_counter() - t0
        whole = n == spec["reads"]
        res["cpu_baseline"] = dict(
            value=n / dt / 1e6, unit="Mreads/s", cores=cores, kind="port",
            solver="oracle's exact active-set LAD (oracle/pantax_oracle.c, same optimum as HiGHS: tests/golden/lp_cases.npz); "
                   "HiGHS itself timed in `highs` on row samples and on the FULL LP of one species",
            sample=("the whole workload: %d reads, %d species; binning in %d read slices, then one species per worker thread on %d threads "
                    "(oracle/oracle_parallel.c)" % (n, S, cores, cores)) if whole else
                   ("the first %d of the %d reads (chunks 0..%d of %d of the generator) over ALL %d species (the per-species index build does not "
                    "shrink with the reads: Mreads/s of the sample is a lower bound of the CPU's rate on the whole workload); binning in %d read "
                    "slices, then one species per worker thread on %d threads (oracle/oracle_parallel.c)"
                    % (n, spec["reads"], n_chunks - 1, synth.N_CHUNKS, S, cores, cores)),
            seconds=dt, reads=n, species_failed=int((pr["rc"] != 0).sum()),
            phases_wall_s={"binning+species_profile": t_bin, "group_reads_by_species": t_group,
                           "per-species (trio index, coverage, filters, 2 LP solves)": dt - t_bin - t_group},
            phases_cpu_s_summed_over_workers={"trio_index": float(pr["t_trio"].sum()), "node_coverage (incl. read selection)": float(pr["t_cov"].sum()),
                                              "filters+LP solves": float(pr["t_lp"].sum())},
            mem_available_gb={"start": res["mem_available_gb_start"], "after_generation": res["mem_available_gb_after_generation"],
                              "after_oracle": _mem_available_gb()}, cgroup_memory_limit_gb=res["cgroup_memory_limit_gb"])
        res["stage"] = "oracle_done"
        publish()
        # the HiGHS legs wait for the parent's timed steps to be over (round-3/4 advisor finding: they ran beside them): the parent drops a file
        t_go = time.time()
        while not os.path.exists(out_path + ".go") and time.time() - t_go < 900:
            if os.getppid() == 1:                  # the parent is gone: nobody will read the HiGHS legs
                res["stage"] = "done"; publish()
                return
            time.sleep(0.1)
        res["waited_for_the_timed_steps_s"] = time.time() - t_go
        # ---- HiGHS legs (single thread each, sequential): row samples + the FULL LP of the species with the most LP rows
        highs_sizes = [int(x) for x in args.highs_rows.split(",") if x.strip()]
        try:
            import scipy.optimize  # noqa: F401
            have_scipy = True
        except Exception as e:   # noqa: BLE001
            have_scipy = False
            res["cpu_baseline"]["highs"] = {"note": "SciPy not importable here (%s): no HiGHS leg" % type(e).__name__}
        if have_scipy and (highs_sizes or args.highs_full_time_limit > 0):
            s_big = int(np.argmax(np.where(keep != 0, pr["n_rows"], 0)))
            lp, _ = _oracle_lp_of_species(orc, ns, sset.species, rd, sp, first, order, s_big, cfgd)
            if lp is not None:
                n_valid = int((lp[1] > 0).sum())
                legs = [_highs_leg(lp, min(sz, n_valid), args.highs_time_limit) for sz in highs_sizes]
                hg = {"species": sset.species[s_big].name, "full_lp_rows": n_valid, "columns": lp[2], "legs": legs}
                res["cpu_baseline"]["highs"] = hg
                publish()
                if args.highs_full_time_limit > 0:
                    full = _highs_leg(lp, None, args.highs_full_time_limit)
                    full["what"] = ("the reference's own solve at size: ALL %d valid rows of this species' LP (highs_opt, profile.rs:2689-2882) under a "
                                    "time limit of %.0f s; `finished` false = HiGHS was still running at the limit, i.e. highs_seconds is a LOWER bound"
                                    % (n_valid, args.highs_full_time_limit))
                    hg["full_lp"] = full
                    publish()
        # ---- pao_hard, CPU side: species 0 of the all-strains-present variant through the oracle, HiGHS on its LP
        if not args.no_hard:
            hs = hard_spec(args.hard_species)
            hns = synth.NativeSet(hs["seed"], hs["species"], hs["haps"], hs["reads"], hs["genome_len"], present_frac=1.0, threads=min(cores, 64))
            hset = hns.make()
            hrd = hset.reads
            hsp = orc.par_bin_reads(hrd.step_off, hrd.node_id, hns.range_start, hns.range_end, min(cores, 32))
            hfirst, horder = orc.group_reads(hsp, hs["species"])
            lp, (t_orc, nc, o1, fs) = _oracle_lp_of_species(orc, hns, hset.species, hrd, hsp, hfirst, horder, 0, {})
            hard = dict(oracle_optimize_species_s_species0=t_orc, oracle_obj1_species0=o1, n_candidates_oracle_species0=nc,
                        oracle_first_sol_species0=fs, hap_names_species0=list(hset.species[0].hap_names))
            if lp is not None and have_scipy and highs_sizes:
                hard["highs"] = [_highs_leg(lp, min(sz, int((lp[1] > 0).sum())), args.highs_time_limit) for sz in highs_sizes]
                hard["lp_rows_species0"] = int((lp[1] > 0).sum())
                hard["lp_columns_species0"] = lp[2]
            res["pao_hard_cpu"] = hard
        res["stage"] = "done"
        publish()
    except BaseException as e:   # noqa: BLE001 -- the parent prints the GPU line regardless
        import traceback
        res["error"] = "%s: %s" % (type(e).__name__, e)
        res["traceback"] = traceback.format_exc()[-2000:]
        res["stage"] = "failed"
        publish()
        return 1
    return 0


def hard_spec(n_species):
    return dict(name="pao_hard", label="pao_hard", seed=20260601, species=n_species, haps=10, reads=1_000_000 * n_species, genome_len=5_000_000)


class CpuLeg:
    """The child process of the CPU legs, seen from the parent."""

    def __init__(self, args, rank, world):
        self.proc, self.path, self.err = None, None, None
        if rank != 0 or world != 1 or args.no_cpu_baseline:
            return
        fd, self.path = tempfile.mkstemp(prefix="pantax_cpu_leg_", suffix=".json")
        os.close(fd)
        os.unlink(self.path)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg-child", self.path, "--workload", args.workload,
               "--cpu-cores", str(args.cpu_cores), "--highs-rows", args.highs_rows, "--highs-time-limit", str(args.highs_time_limit),
               "--highs-full-time-limit", str(args.highs_full_time_limit), "--hard-species", str(args.hard_species)]
        for k in ("species", "haps", "reads", "genome_len"):
            v = getattr(args, k)
            if v is not None:
                cmd += ["--" + k.replace("_", "-"), str(v)]
        if args.no_hard:
            cmd.append("--no-hard")
        self.errfile = tempfile.TemporaryFile()
        self.proc = subprocess.Popen(cmd, stdout=self.errfile, stderr=subprocess.STDOUT, cwd=ROOT)   # a CHILD, started before this process touches the GPU

    def _read(self):
        try:
            with open(self.path) as f:
                return json.load(f)
        except (OSError, ValueError):
            return None

    def wait_stage(self, stages, timeout):
        """block until the child has published one of `stages` (or exited / timed out) -> its result dict or None"""
        if self.proc is None:
            return None
        t_end = time.time() + timeout
        while time.time() < t_end:
            d = self._read()
            if d and d.get("stage") in stages:
                return d
            if self.proc.poll() is not None:
                return self._read()
            time.sleep(0.2)
        return self._read()

    def go(self):
        """the timed steps are over: the child may start its HiGHS legs"""
        if self.path is not None:
            try:
                open(self.path + ".go", "w").close()
            except OSError:
                pass

    def finish(self, timeout):
        if self.proc is None:
            return None
        self.go()
        try:
            self.proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            self.proc.kill()          # the exact process this object started
            self.proc.wait()
            self.err = "child still running after %.0f s: killed" % timeout
        d = self._read() or {}
        if self.proc.returncode not in (0, None) and "error" not in d:
            self.errfile.seek(0)
            d["error"] = (self.err or "child exit code %s" % self.proc.returncode) + ": " + self.errfile.read()[-1500:].decode(errors="replace")
        for f in (self.path, self.path + ".go"):
            try:
                os.unlink(f)
            except OSError:
                pass
        return d


PMC_ROUND = "r06"   # profiles/<PMC_ROUND>_pmc_<workload>.json: the counter passes of THIS round's code; older files are never read


def pmc_traffic(kernel, key, wl_name, corrected=False):
    """HBM bytes per launch of `kernel` from THIS round's committed rocprofv3 --pmc passes of the same workload (FETCH_SIZE +
    WRITE_SIZE, separate passes, KB -> bytes; profiles/<PMC_ROUND>_pmc_<workload>.json, collected by tools/pmc_step.sh).  None when
    the file is missing, is of another workload, or does not hold the kernel -- never a number of an older round.  corrected:
    2 x FETCH + WRITE -- the guide's gfx950 correction (FETCH_SIZE tallies 128-byte requests at 64 B for wide coalesced reads) applied
    to the whole fetch, i.e. an upper bound."""
    fn = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (PMC_ROUND, wl_name))
    try:
        d = json.load(open(fn))
        w = d.get("workload", {})
        if any(w.get(k) != v for k, v in key.items()):
            return None
        # (the long-walk coverage kernel is an instantiation of coverage_fast_kernel: the profiler lists it under that name -- in a long-read workload,
        # where the short-read instantiation is not launched, the entry is the long-walk kernel's)
        k = d["kernels"].get(kernel) or (d["kernels"].get("coverage_fast_kernel") if kernel == "coverage_long_kernel" and key.get("long_reads") else None)
        if k and "hbm_bytes_per_launch" in k:
            return k["hbm_bytes_fetch_x2"] if corrected else k["hbm_bytes_per_launch"]
    except Exception:   # noqa: BLE001
        pass
    return None


def gaf_tmp_dir(need_bytes):
    """where the from-GAF-text leg writes its file: the first of $TMPDIR, /tmp (a file system with a page cache, like a real
    input), /dev/shm with room for it"""
    for d in (tempfile.gettempdir(), "/tmp", "/dev/shm"):
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize > 1.25 * need_bytes + (1 << 30):
                return d
        except OSError:
            continue
    return None


def abundance_l1_leg(eng, ns, species, rd, out, cfg, threads, n_sample=50, sp=None, rc=None, extra_pick=()):
    """north_star: strain abundances within L1 1e-4 of the solver-backed PAO.  For a sample of species (the one with the most and the
    fewest reads + evenly spaced ones) the CHECKER (oracle/: trio index, coverage, both exact LAD solves, constraint -- whose LP
    optimum equals SciPy-HiGHS on the golden fixtures) runs on the species' reads of THIS workload at full size, and the strain rows of
    the timed GPU step are compared with it: relative L1 = sum |x_gpu - x_oracle| / sum |x_oracle| over the species' reported strains,
    for the LP solution (first_sol) and for the final predicted coverage.  A species whose LP optimum is a face rather than a point is
    listed (any point of the face is an optimum; the objectives are compared instead)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from tests.helpers import select_reads
    t0 = time.perf_counter()
    species_rows, strain_rows, stats = out
    S = len(species)
    if sp is None:
        sp, rc, *_ = eng.rcls_profile()
    cnt = np.asarray(rc)
    pick = sorted({int(np.argmax(cnt)), int(np.argmin(np.where(cnt > 0, cnt, cnt.max() + 1)))} | {int(i) for i in np.linspace(0, S - 1, max(n_sample - 2, 1)).astype(int)})
    pick = sorted(set(pick) | {int(x) for x in extra_pick})
    kept_names = {r[0] for r in species_rows}
    pick = [s for s in pick if species[s].name in kept_names]      # species the step kept
    cov_of = {r[0]: r[2] for r in species_rows}
    by_sp = {}
    for r in strain_rows:
        by_sp.setdefault(r[0], {})[r[1]] = r
    first_, order = orc.group_reads(sp, S)                          # one pass over the reads (a scan per picked species was 50 ms each at 1e8 reads)
    first_ = first_.astype(np.int64)
    sel_all = {s: np.sort(order[first_[s]:first_[s + 1]].astype(np.int64)) for s in pick}
    cfgd = dict(fr=cfg.fr, fc=cfg.fc, sr=cfg.sr)

    def one(s):
        g = species[s]
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        so, nid, ps, pe = select_reads(rd, sel_all[s])
        b, c, t, _ = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
        _, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, t, **cfgd)
        orc.abundance_constraint(cov_of[g.name], met)
        om = orc.metrics_to_dicts(met)
        rows = by_sp.get(g.name, {})
        d1 = n1 = d2 = n2 = 0.0
        for h, hn in enumerate(g.hap_names):
            if hn not in rows:
                continue
            e1, e2 = om[h]["first_sol"] or 0.0, om[h]["predicted_coverage"] or 0.0
            d1 += abs(rows[hn][7] - e1); n1 += abs(e1)
            d2 += abs(rows[hn][2] - e2); n2 += abs(e2)
        return dict(species=g.name, reads=int(len(sel_all[s])), strains_reported=len(rows), candidates_oracle=int(nc),
                    l1_first_sol=d1 / n1 if n1 else 0.0, l1_predicted_coverage=d2 / n2 if n2 else 0.0, oracle_obj1=o1)
    with ThreadPoolExecutor(min(threads, max(len(pick), 1))) as ex:
        per = list(ex.map(one, pick))
    worst = max([max(p_["l1_first_sol"], p_["l1_predicted_coverage"]) for p_ in per], default=0.0)
    return dict(abundance_l1_vs_oracle=worst, tolerance=1e-4, species_checked=len(per), per_species=per, seconds=time.perf_counter() - t0,
                what="relative L1 of the timed step's strain rows against the oracle (exact LAD == HiGHS optimum on the fixtures) on the same reads, full size")


class _StderrCapture:
    """fd 2 into a temporary file for the duration of the block (the library's phase times go to the C-level stderr)."""

    def __enter__(self):
        self.tmp = tempfile.TemporaryFile()
        sys.stderr.flush()
        self.saved = os.dup(2)
        os.dup2(self.tmp.fileno(), 2)
        return self

    def __exit__(self, *a):
        os.dup2(self.saved, 2)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode(errors="replace")
        self.tmp.close()


def _seam_phases(text):
    """'[pantax_hip_profile r0] phase   12.345 ms' lines -> {phase: ms} (the last value of a repeated phase wins)"""
    out = {}
    for ln in text.splitlines():
        if ln.startswith("[pantax_hip_profile r0]") and ln.rstrip().endswith("ms"):
            body = ln[len("[pantax_hip_profile r0]"):].rsplit(None, 2)
            try:
                out[body[0].strip()] = out.get(body[0].strip(), 0.0) + float(body[1])      # (a phase repeats once per group of species: summed)
            except (IndexError, ValueError):
                pass
    return out


def _read_table(path):
    with open(path) as f:
        f.readline()
        return [ln.rstrip("\n").split("\t") for ln in f]


def seam_child(spec_path):
    """`bench.py --seam-child <json>`: the pantax_hip_profile calls of the files -> tables leg in a process of their own -- what a `pantax-hip` run is: a
    fresh process, nothing resident, no torch (the parent's interpreter with torch's thread pools loaded ran the same call 0.1 s slower and noisier:
    the crew that fills the pinned ring competes for the host).  Writes its result (times, phases, traces, work directories) to <json>.out."""
    with open(spec_path) as f:
        a = json.load(f)
    from pantax_amd.engine import Engine
    db, td, gaf_path, fr = a["db"], a["td"], a["gaf"], a["fr"]
    gi = os.path.join(db, "species_graph_info")
    res = {}
    eng = Engine(int(a.get("device", 0)))
    eng.set_option("hip_trace", "1")
    cwd = os.getcwd()

    def call(name, image_cache):
        wd = os.path.join(td, name)
        os.mkdir(wd)
        os.chdir(wd)
        try:
            with _StderrCapture() as cap:
                t = time.perf_counter()
                eng.profile(db, wd, gaf_path, zip="serialize", sample_nodes=0, image_cache=image_cache, fr=fr)   # fr: the resident step's (0.5 for long reads, as the reference's wrapper sets it)
                dt = time.perf_counter() - t
        finally:
            os.chdir(cwd)
        # (the per-piece lines of the tokenizer and the per-pipeline lines of the graph loader are dropped: a hundred of them per run)
        res.setdefault("trace", {})[name] = "\n".join(l for l in cap.text.splitlines() if "[gaf_tokenize]   piece" not in l and "[upload_segments]" not in l)[-9000:]
        return wd, dt, _seam_phases(cap.text)
    try:
        call("wd_prime", 0)                                  # allocations, page cache
        wd_c, t_cold, ph_c = call("wd_cold", 0)
        _, t_img, _ = call("wd_images", 2)                   # leaves the images behind (not timed as a result)
        res["db_image_gb"] = sum(os.path.getsize(os.path.join(gi, f)) for f in os.listdir(gi) if f.endswith(".hipdb")) / 1e9
        warm = [call("wd_warm%d" % i, 1) for i in range(2)]
        wd_w, t_warm, ph_w = min(warm, key=lambda r: r[1])
        db_keys = ("graph headers", "db upload", "db upload (a group of the species)")   # (with groups: what this thread waited for the loader + the tables it built)
        n_reads = a["n_reads"]
        res.update(files_to_tables_cold_s=t_cold, files_to_tables_warm_s=t_warm, files_to_tables_warm_s_both=[r[1] for r in warm],
                   image_writing_run_s=t_img, db_load_cold_s=sum(ph_c.get(k, 0.0) for k in db_keys) / 1e3, db_load_warm_s=sum(ph_w.get(k, 0.0) for k in db_keys) / 1e3,
                   gaf_load_s=ph_w.get("ranges + GAF tokenise", 0.0) / 1e3, strain_step_s=ph_w.get("strain step", 0.0) / 1e3,
                   phases_ms_cold=ph_c, phases_ms_warm=ph_w, mreads_per_s_warm=n_reads / t_warm / 1e6, mreads_per_s_cold=n_reads / t_cold / 1e6,
                   wd_cold=wd_c, wd_warm=wd_w, process="a child process of bench.py (fresh, no torch): what a pantax-hip run is")
    except Exception as e:   # noqa: BLE001 -- reported to the parent
        res["error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        eng.close()
    with open(spec_path + ".out", "w") as f:
        json.dump(res, f)
    return 0


def file_seam_leg(local_rank, species, gaf_path, td, threads, out, n_reads, fr=0.3):
    """The wall time the reference itself logs (profile.rs:3326-3327, :3429-3433): profile::profile, FILES to FILES, graph loading
    included -- here pantax_hip_profile on a real DB directory of this workload (species_range.txt, species_genomes_stats.txt,
    genomes_info.txt, one bincode `.bin` per species, zip.rs:171-190) + the GAF text.  cold: graphs from the `.bin` files (64-bit
    values narrowed on their way into the pinned ring); warm: from the device-ready images a run with image_cache = 2 leaves behind.
    Page cache warm in both (the files were just written).  The calls run in a CHILD process (seam_child); this process writes the DB
    directory before and compares the tables after.  Never `value`."""
    import subprocess
    import synthdata as synth
    res = {}
    db = os.path.join(td, "db")
    os.mkdir(db)
    t0 = time.perf_counter()
    synth.write_db(synth.SyntheticSet(species, None), db, write_gfa=False, threads=threads)
    res["db_written_in_s"] = time.perf_counter() - t0
    gi = os.path.join(db, "species_graph_info")
    res["db_bin_gb"] = sum(os.path.getsize(os.path.join(gi, f)) for f in os.listdir(gi)) / 1e9
    spec_path = os.path.join(td, "seam_child.json")
    with open(spec_path, "w") as f:
        json.dump(dict(db=db, td=td, gaf=gaf_path, fr=fr, n_reads=n_reads, device=local_rank), f)
    env = dict(os.environ)
    env.pop("PANTAX_HIP_TRACE", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--seam-child", spec_path], env=env, timeout=1500)
    if r.returncode != 0 or not os.path.exists(spec_path + ".out"):
        raise RuntimeError("the seam child exited with %d" % r.returncode)
    with open(spec_path + ".out") as f:
        res.update(json.load(f))
    if res.get("error"):
        raise RuntimeError(res["error"])
    wd_c, wd_w = res.pop("wd_cold"), res.pop("wd_warm")
    if True:
        # the tables of the seam against the resident step's (same species order, values to 1e-9) and cold against warm (same bytes)
        same_bytes = all(open(os.path.join(wd_c, f)).read() == open(os.path.join(wd_w, f)).read() for f in ("species_abundance.txt", "strain_abundance.txt"))
        tsp = _read_table(os.path.join(wd_w, "species_abundance.txt"))
        tst = _read_table(os.path.join(wd_w, "strain_abundance.txt"))
        eq = None
        if out is not None:
            sp_rows, st_rows, _ = out
            close = lambda a, b: abs(a - b) <= 1e-9 * max(1.0, abs(b))
            # row by row where the sort key differs; rows with EQUAL abundance may stand in either order (the file writer keeps the order of the species
            # table among ties -- what the oracle comparison of the file-seam test pins --, the Python tables of the resident step their arrival order):
            # the rows are matched by (species, strain) and both tables must be sorted
            eq = len(tsp) == len(sp_rows) and len(tst) == len(st_rows)
            a_sp = {t[0]: (float(t[1]), float(t[2])) for t in tsp}
            eq = eq and all(r[0] in a_sp and close(a_sp[r[0]][0], r[1]) and close(a_sp[r[0]][1], r[2]) for r in sp_rows)
            by_sp = {}
            for t in tst:
                by_sp.setdefault(t[0], []).append(t)

            def strain_matches(r):
                hit = [t for t in by_sp.get(r[0], []) if t[2].startswith(r[1])]
                return len(hit) == 1 and close(float(hit[0][3]), r[2]) and close(float(hit[0][4]), r[3])
            eq = eq and all(strain_matches(r) for r in st_rows)
            same_order = all(t[0] == r[0] for t, r in zip(tsp, sp_rows)) and all(t[0] == r[0] and t[2].startswith(r[1]) for t, r in zip(tst, st_rows))
            res["rows_in_the_same_order_as_resident_step"] = same_order
        res.update(cold_and_warm_tables_same_bytes=same_bytes, tables_equal_to_resident_step=eq, n_strain_rows=len(tst))
    return res


def run_many_dbs(args, spec, local_rank):
    """A workload of more path steps than one resident db addresses (32-bit positions: BASELINE configs[4] at its size, 1 000 species x 50
    strains = 1.1e10 path steps) on ONE GPU: the species are cut into contiguous groups under the limit, every group is a db on a ctx of its
    own with the reads of its species (species are independent from a4 on, profile.rs:3297-3319), the dbs are stepped side by side and their
    results meet like those of ranks (pipeline.profile_steps_many) -- the global normalisers of profile.rs:341, :3198, :3243 over all of them."""
    import synthdata as synth
    from pantax_amd.engine import Engine
    from pantax_amd.pipeline import StepConfig, profile_steps_many, split_species_by_path_steps
    import torch
    cfg = StepConfig(fr=0.5) if spec.get("long_reads") else StepConfig()
    n_species, n_haps, n_reads, genome_len = spec["species"], spec["haps"], spec["reads"], spec["genome_len"]
    host_threads = max(1, min(64, os.cpu_count() or 1))
    torch.cuda.set_device(local_rank)
    t_gen = time.perf_counter()
    ns = native_set(spec, threads=host_threads)
    rd = ns.reads()
    species = ns.graphs()
    gen_s = time.perf_counter() - t_gen
    groups = split_species_by_path_steps(ns.P)
    K = len(groups)
    # the reads of every db: by the species that holds their first node (a walk that leaves its species is "U" wherever it is binned)
    so = rd.step_off.astype(np.int64)
    klen = np.diff(so)
    first = np.where(klen > 0, rd.node_id[np.minimum(so[:-1], max(len(rd.node_id) - 1, 0))].astype(np.int64), 0)
    sp_of = np.clip(np.searchsorted(ns.range_start, first, side="right") - 1, 0, n_species - 1)
    t_up = time.perf_counter()
    engs, names_l, haps_l, avg_l, R_l, T_l = [], [], [], [], [], []
    avg_all = ns.avg_len()
    mapq = np.where((rd.mapq < 0) | (rd.mapq > 254), 255, rd.mapq)
    for gi, (a, b) in enumerate(groups):
        sel = np.nonzero((sp_of >= a) & (sp_of < b) & ((klen > 0) | (gi == 0)))[0]
        nsteps = klen[sel]
        off = np.zeros(len(sel) + 1, dtype=np.int64)
        np.cumsum(nsteps, out=off[1:])
        idx = np.repeat(so[:-1][sel] - off[:-1], nsteps) + np.arange(int(off[-1]), dtype=np.int64)
        eng = Engine(local_rank)
        eng.upload_db(species[a:b])
        eng.upload_reads(off, rd.node_id[idx], rd.pstart[sel], rd.pend[sel], rd.qlen[sel], mapq[sel])
        eng.sync()
        engs.append(eng)
        names_l.append([g.name for g in species[a:b]])
        haps_l.append([hn for g in species[a:b] for hn in g.hap_names])
        avg_l.append(avg_all[a:b])
        R_l.append(len(sel)); T_l.append(int(off[-1]))
        del idx
    upload_ms = (time.perf_counter() - t_up) * 1e3
    # `value` = the step over resident dbs (graphs + their unique-trio index), as in main(); the same step with the index rebuilt inside it beside it
    import dataclasses
    cfg_main = dataclasses.replace(cfg, rebuild_trio=False)
    run = lambda n, c=cfg_main, serial=False: profile_steps_many(engs, names_l, haps_l, avg_l, n, c, one_after_the_other=serial)

    def barrier():
        torch.cuda.synchronize()
        for e in engs:
            e.sync()

    def timings():
        tot = {}
        for e in engs:
            for k, (n, ms) in e.timing_get().items():
                a = tot.setdefault(k, [0, 0.0])
                a[0] += n; a[1] += ms
        return tot
    run(2)
    barrier()
    for e in engs:
        e.timing_enable(True); e.timing_reset()
    # the per-kernel clocks (kernel table, roofline) come from warm-up steps in which the dbs are stepped ONE AFTER THE OTHER: side by side, a
    # kernel's events also span the other dbs' kernels it shares the GPU with, and the table would add up to several times the step
    n_warm = max(args.warmup, 1)
    barrier()
    t_ser = time.perf_counter()
    out = run(n_warm, serial=True)[-1]
    barrier()
    ms_serial = (time.perf_counter() - t_ser) / n_warm * 1e3
    warm = timings()
    cov_kernel = "coverage_fast_kernel" if "coverage_fast_kernel" in warm else ("coverage_long_kernel" if "coverage_long_kernel" in warm else "coverage_step_kernel")
    top2 = [k for k, _ in sorted(warm.items(), key=lambda kv: -kv[1][1])[:6]]
    for e in engs:
        e.timing_enable(False)
    run(1)
    import gc
    gc.collect(); gc.freeze()
    barrier()
    t0 = time.perf_counter()
    out = run(args.steps)[-1]
    barrier()
    dt = time.perf_counter() - t0
    tm = warm
    run(1, cfg)
    barrier()
    t1 = time.perf_counter()
    run(args.steps, cfg)
    barrier()
    dt_rebuild = time.perf_counter() - t1
    # abundance L1 against the oracle for a sample of species (species binned on the host by the oracle's own rule)
    l1 = None
    if not args.no_l1:
        try:
            from oracle import oracle as orc
            sp = orc.par_bin_reads(rd.step_off, rd.node_id, ns.range_start, ns.range_end, host_threads)
            cnt = np.bincount(sp[sp >= 0], minlength=n_species)
            l1 = abundance_l1_leg(None, ns, species, rd, out, cfg, host_threads, n_sample=7, sp=sp, rc=cnt)
        except Exception as e:   # noqa: BLE001 -- the line is printed regardless
            l1 = {"error": "%s: %s" % (type(e).__name__, e)}
    species_rows, strain_rows, stats = out
    ms_per_step = dt / args.steps * 1e3
    n_lp_rows = int(sum(stats["n_rows"]))
    n_unique = int(sum(e.trio_nodes_info(fetch=False) for e in engs))
    ab, dims = algorithmic_bytes(species, n_lp_rows, n_unique, sum(R_l), sum(T_l))

    def ruler(k):
        launches, tot_ms = tm[k]
        per_step_ms = tot_ms / n_warm
        return dict(kernel=k, ms_per_step_summed_over_the_dbs=round(per_step_ms, 4), launches_timed=launches, algorithmic_bytes=ab.get(k, 0),
                    achieved=ab.get(k, 0) / (per_step_ms * 1e-3) / 1e9, frac=ab.get(k, 0) / (per_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
    roofline = None
    ruled = sorted((k for k in top2 if k in tm and ab.get(k)), key=lambda k: -tm[k][1])
    dom = ruled[0] if ruled else None
    if dom:
        r0 = ruler(dom)
        roofline = dict(bound="hbm", kernel=dom, achieved=r0["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=r0["frac"], traffic=None,
                        avg_ms=r0["ms_per_step_summed_over_the_dbs"], launches_timed=r0["launches_timed"], algorithmic_bytes=r0["algorithmic_bytes"],
                        note="bytes of the whole workload over the kernel's time per step summed over the %d dbs' launches, clocked in the warm-up steps "
                             "(the dbs one after the other: every kernel has the GPU to itself); the timed steps run the dbs side by side" % K,
                        traffic_source="no PMC record for this workload: traffic null")
        roofline["kernels"] = [{k: ruler(k2)[k] for k in ("kernel", "ms_per_step_summed_over_the_dbs", "algorithmic_bytes", "achieved", "frac")} for k2 in ruled[:5]]
        if cov_kernel in tm and cov_kernel not in ruled[:5]:
            roofline["coverage"] = {k: ruler(cov_kernel)[k] for k in ("kernel", "ms_per_step_summed_over_the_dbs", "algorithmic_bytes", "frac")}
    line = {
        "metric": "PAO wall-time (s) + Mreads/s: packed reads + DBs (graphs and their unique-trio index) resident in HBM -> abundance tables",
        "value": n_reads / (dt / args.steps) / 1e6, "unit": "Mreads/s", "n_gpus": 1, "steps": args.steps, "warmup": n_warm,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
        "config": {"workload": "%s: %d species x %d strains, %d %s, genome %d bp, seed %d, generator %s"
                               % (spec["name"], n_species, n_haps, n_reads, "long reads N(15000, 3000^2) bp" if spec.get("long_reads") else "short reads (150 bp)",
                                  genome_len, spec["seed"], GENERATOR),
                   "baseline_config": spec["label"][:60], "dbs_on_the_gpu": K, "species_per_db": [b - a for a, b in groups],
                   "why_several_dbs": "a resident db addresses its path steps with 32 bits; %d path steps are cut by species (species are independent)" % dims["P"],
                   "gsteps_per_s": sum(T_l) / (dt / args.steps) / 1e9, "V": dims["V"], "P": dims["P"], "T": dims["T"], "U": n_unique,
                   "strains_total": n_species * n_haps, "reads_total": n_reads, "parallelism": "species-shard x1 (%d dbs side by side)" % K,
                   "pao_wall_s": ms_per_step / 1e3, "ms_per_step_with_index_rebuild": dt_rebuild / args.steps * 1e3,
                   "ms_per_step_dbs_one_after_the_other": ms_serial,
                   "abundance_l1_vs_oracle": (l1 or {}).get("abundance_l1_vs_oracle"), "abundance_l1_species_checked": (l1 or {}).get("species_checked"),
                   "abundance_l1_tolerance": 1e-4, "abundance_l1_error": (l1 or {}).get("error"),
                   "lp_rows_total": n_lp_rows, "n_species_rows": len(species_rows), "n_strain_rows": len(strain_rows), "upload_ms_once": upload_ms,
                   "synthetic_set_generated_in_s": gen_s, "sample_nodes": 0},
        "roofline": roofline,
        "kernels_ms_per_step": {k: round(v[1] / n_warm, 3) for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])[:14]},
        "result": {"n_species_rows": len(species_rows), "n_strain_rows": len(strain_rows),
                   "top_strains": [(r[0], r[1], round(r[2], 4), round(r[3], 6)) for r in strain_rows[:3]]},
    }
    detail = {"line": line, "abundance_l1": l1, "kernels_ms_per_step_all": {k: v[1] / n_warm for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])},
              "reads_per_db": R_l, "steps_per_db": T_l, "host": {"cores": os.cpu_count(), "mem_available_gb": _mem_available_gb()}}
    dpath = args.detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail_%s_n1.json" % spec["name"])
    try:
        os.makedirs(os.path.dirname(dpath), exist_ok=True)
        with open(dpath, "w") as f:
            json.dump(detail, f, indent=1)
        line["detail_file"] = os.path.relpath(dpath, ROOT)
    except OSError as e:
        line["detail_file"] = "not written: %s" % e
    print(json.dumps(line), flush=True)
    for e in engs:
        e.close()


def launch_ranks(n):
    """One node, n ranks: python -m torch.distributed.run ... bench.py <the same arguments>, as a child process."""
    import socket
    # a free port for the rendezvous.  Bind-then-close leaves a window in which a second bench started at the same moment may be given the same port
    # (round-4 advisor finding): the port is drawn from this process's own slice of the dynamic range first (pid-keyed), so two launchers only collide
    # if the kernel hands both the same fallback
    port = None
    for k in range(16):
        cand = 20000 + (os.getpid() * 16 + k) % 40000
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", cand)); port = cand; break
            except OSError:
                continue
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def launcher_selftest(rank, world):
    """The launcher path without a GPU (CPU test): the ranks meet over gloo, rank 0 reports how many it saw."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": int(t.item()), "world_size": dist.get_world_size()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); without a launcher around it, --gpus N > 1 starts the N ranks itself")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("--species", type=int, default=None)
    ap.add_argument("--haps", type=int, default=None)
    ap.add_argument("--genome-len", type=int, default=None)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): ONE set of the given size cut over the ranks; weak: one such set per rank")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker threads of the CPU baseline (0 = all host cores)")
    ap.add_argument("--highs-rows", default="2000,5000,10000", help="row samples of the HiGHS legs ('' = none)")
    ap.add_argument("--highs-time-limit", type=float, default=25.0)
    ap.add_argument("--highs-full-time-limit", type=float, default=120.0, help="time limit of HiGHS on the FULL LP of one species (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gaf", action="store_true", help="skip the from-GAF-text measurement")
    ap.add_argument("--no-seam", action="store_true", help="skip the files-to-files leg (pantax_hip_profile on a DB directory of the workload)")
    ap.add_argument("--gaf-reads", type=int, default=0, help="reads of the from-GAF-text leg (0 = the whole workload)")
    ap.add_argument("--no-hard", action="store_true", help="skip the pao_hard leg")
    ap.add_argument("--no-l1", action="store_true", help="skip the abundance-L1-vs-oracle leg")
    ap.add_argument("--detail-file", default=None, help="where the verbose side record goes (default gpurun_out/bench_detail_<workload>_n<N>.json)")
    ap.add_argument("--hard-species", type=int, default=8)
    ap.add_argument("--cpu-leg-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--seam-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_leg_child:
        sys.exit(cpu_leg_child(args))
    if args.seam_child:
        sys.exit(seam_child(args.seam_child))
    # ---- `python3 bench.py --gpus N` without a launcher around it: this process -- which has touched neither torch nor the GPU --
    # starts the N ranks as a CHILD (never an exec) and relays rank 0's line and the exit code
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus != world:
        print("bench.py: --gpus %d under a launcher with WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.launcher_selftest:
        sys.exit(launcher_selftest(rank, world))
    spec = workload_spec(args.workload, args.species, args.haps, args.reads, args.genome_len)
    n_species, n_haps, n_reads, genome_len = spec["species"], spec["haps"], spec["reads"], spec["genome_len"]
    if spec["name"] == "cfg5":          # more path steps than one resident db addresses: several dbs on the one GPU (N = 1 only)
        if world != 1:
            print("bench.py: --workload cfg5 runs on one GPU (over N GPUs every rank holds its share: --workload cfg5_share --scaling weak)", file=sys.stderr)
            sys.exit(2)
        return run_many_dbs(args, spec, local_rank)

    # ---- CPU legs: a child process, started before anything initialises the GPU here (rank 0, N = 1 only).  The parent waits
    # for the oracle leg -- it uses every core -- and then goes on beside the child's single-threaded HiGHS legs.
    if spec.get("refdb"):
        args.no_cpu_baseline = True      # the CPU baseline belongs to the headline workload (cfg4); 8 778 oracle species would not fit the bounded sample
        args.no_seam = True              # (the file seam is exercised at cfg4; a DB directory of 8 778 species is a test of the file system)
        args.no_hard = True
    leg = CpuLeg(args, rank, world)
    t_cpu_wait = time.perf_counter()
    leg.wait_stage(("oracle_done", "done", "failed"), timeout=900)
    cpu_wait_s = time.perf_counter() - t_cpu_wait

    import synthdata as synth
    from pantax_amd.pipeline import LocalComm, StepConfig, TorchComm, partition_species, profile_step, profile_steps_pipelined
    cfg = StepConfig(fr=0.5) if spec.get("long_reads") else StepConfig()   # long reads: --fr 0.5 (main.rs:108-114)
    import torch
    # torch is here for torch.distributed only; its intra-op pool (one thread per core by default) is not needed and competes with the library's own
    # host threads (the crew that fills the pinned upload ring)
    try:
        torch.set_num_threads(max(1, min(4, os.cpu_count() or 1)))
    except Exception:   # noqa: BLE001
        pass
    from pantax_amd.engine import Engine
    # PANTAX_BENCH_BACKEND=gloo: dry run of the N > 1 flow on a box with fewer GPUs than ranks (ranks share devices, the
    # exchange goes over gloo); the driver's runs use the default, RCCL with one GPU per rank
    backend = os.environ.get("PANTAX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    comm = LocalComm()
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            comm = TorchComm(device=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            comm = TorchComm(device=None)
        assert dist.get_world_size() == world
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ones)                          # every rank is there (over RCCL with the default backend)
        ranks_seen = int(ones.item())
        assert ranks_seen == world

    # ---- deterministic synthetic input (seed = 20260501 + cfg index).  strong: ONE set; this rank generates its
    # slice of the reads (chunks of the generator) and, later, the graphs of the species it owns.  weak: one set per rank.
    host_threads = max(1, min(64, (os.cpu_count() or 1) // max(world, 1)))
    strong = args.scaling == "strong"
    t_gen = time.perf_counter()
    ns = native_set(spec, threads=host_threads, seed_shift=0 if strong else 1000 * rank)
    if strong:
        c_lo, c_hi = synth.N_CHUNKS * rank // world, synth.N_CHUNKS * (rank + 1) // world
        total_reads = n_reads
    else:
        c_lo, c_hi = 0, synth.N_CHUNKS
        ns.names = ["%d" % (100000 * rank + 1000 + i) for i in range(n_species)]
        total_reads = n_reads * world
    if spec.get("refdb"):
        if world != 1:
            print("bench.py: --workload refdb runs on one GPU", file=sys.stderr)
            sys.exit(2)
        n_species, n_haps = ns.S, "1..10"
    rd = ns.reads(c_lo, c_hi)                       # generates (and caches) all graphs: the walks of every present strain are needed
    gen_s = time.perf_counter() - t_gen
    wkey = workload_key(spec)

    eng = Engine(local_rank)
    t_up = time.perf_counter()
    ingest_route = None
    if strong and world > 1:
        # this rank's slice of the reads (file order) is binned against ALL species ranges, the species are packed
        # onto the ranks by the summed counts, and the packed records travel to their owners -- once, before the timed steps
        from pantax_amd.pipeline import route_reads
        eng.upload_ranges(ns.range_start, ns.range_end)
        eng.upload_packed(rd)
        eng.sync()
        rd = None
        t_r = time.perf_counter()
        _, rc_loc, *_ = eng.rcls_profile(want_species=False)
        rc_all = comm.allreduce_sum(rc_loc)
        owner = partition_species([8.0 * float(rc_all[i]) + float(ns.V[i]) for i in range(n_species)], world)
        rstats = route_reads(eng, owner, comm)
        eng.sync()
        ingest_route = dict(ms=(time.perf_counter() - t_r) * 1e3, **rstats,
                            what="bin the slice against all ranges + one all-reduce of the counts + pack + all-to-all(v) + rebuild resident reads")
        mine = [i for i in range(n_species) if owner[i] == rank]
        ns.drop_graphs(mine)
        species = ns.graphs(mine)
        avg_len = ns.avg_len()[mine]
        eng.upload_db(species)
    else:
        species = ns.graphs()
        avg_len = ns.avg_len()
        eng.upload_db(species)
        eng.upload_packed(rd)          # inputs resident in HBM before the timed region
    eng.sync()
    R_res, T_res = eng.R, eng.T                      # reads / walk steps resident on this rank
    upload_ms = (time.perf_counter() - t_up) * 1e3   # host->device of packed reads + graph (pageable memory, incl. numpy packing)
    species_names = [g.name for g in species]
    hap_names = [hn for g in species for hn in g.hap_names]
    S_loc = len(species)
    S_max, H_max = S_loc, len(hap_names)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([S_loc, len(hap_names)], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        S_max, H_max = int(t[0].item()), int(t[1].item())
    # tables on a helper thread or inline: decided from a value that is the same on every rank (the ranks of a strong-scaling
    # run own different numbers of strains; all collectives of the communicator stay on one thread per rank either way)
    from pantax_amd.pipeline import PIPELINE_THREAD_MIN_HAPS
    threaded = H_max >= PIPELINE_THREAD_MIN_HAPS
    # WHAT `value` TIMES (round 6, settled): the unique-trio index is a function of the DB alone (SURVEY 8f-2 makes it a DB artefact), so the step of the
    # headline runs over a resident DB = graphs + that index, like every sample after a deployment's first one: cfg_main.  The same step with the index
    # REBUILT inside it (what a one-sample process pays, as the reference does, profile.rs:2936) is reported beside it, one-pass and two-pass, and so is
    # the db's very first index build.
    import dataclasses
    cfg_main = dataclasses.replace(cfg, rebuild_trio=False)
    run_steps = lambda n, c=cfg_main: profile_steps_pipelined(eng, species_names, hap_names, avg_len, n, c, comm, shard_max=S_max, rows_max=H_max, threaded=threaded)
    t_ix = time.perf_counter()
    eng.trio_nodes_info(fetch=False)                 # the db's FIRST index build: visit kernel -> records -> prefix -> rows kernel, with the host waits a first build has
    eng.sync()
    index_first_build_ms = (time.perf_counter() - t_ix) * 1e3

    # one-time set-up of the step path (not the W warm-up steps of the contract, which follow): the first stream of steps that
    # keeps one step enqueued ahead makes the HIP runtime grow its pools (a ~6 ms stall at the third enqueue, measured); it
    # happens here, with the uploads, not inside the timed region
    t_pr = time.perf_counter()
    run_steps(3)
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
    # host hygiene, ahead of the warm-up steps so that the device goes from them straight into the timed region (round 6: between the two, the 0.1 s of this
    # collection let the clocks drop -- the first timed step took 2-3 ms longer than the rest): with torch imported the interpreter holds ~1e6 long-lived
    # objects, and a full collection of the cyclic garbage collector (triggered by the tables' tuples every few dozen steps) stops the thread that
    # enqueues the next step for 30-50 ms; frozen objects are not scanned again
    import gc
    gc.collect()
    gc.freeze()
    eng.timing_enable(True)
    eng.timing_reset()
    n_warm_timed = 0
    if args.warmup:
        # the very first step also allocates: its launches are not representative
        out = profile_step(eng, species_names, hap_names, avg_len, cfg_main, comm, shard_max=S_max, rows_max=H_max)
        n_warm_timed = 1
    if args.warmup > 1:
        # the remaining warm-up steps go through the path that is timed below (one step enqueued ahead): whatever the runtime
        # sets up the first time two steps are in flight happens here, not in the timed region
        eng.timing_reset()
        n_warm_timed = args.warmup - 1
        out = run_steps(n_warm_timed)[-1]
    warm = eng.timing_get() if args.warmup else {}
    # the two largest of the warm-up table are bracketed in the timed steps; the dominant kernel is the one with the larger
    # average THERE
    cov_kernel = "coverage_fast_kernel" if (not warm or "coverage_fast_kernel" in warm) else ("coverage_long_kernel" if "coverage_long_kernel" in warm else "coverage_step_kernel")   # short reads / long reads
    # the six largest of the warm-up table are bracketed in the timed steps as well (a dozen event records per step)
    top2 = [k for k, _ in sorted(warm.items(), key=lambda kv: -kv[1][1])[:6]] if warm else [cov_kernel]
    # ... and the coverage kernel always (the histogram is the path's named kernel)
    eng.timing_filter("|".join(top2 + [k for k in [cov_kernel] if k not in top2]))
    eng.timing_reset()
    barrier()
    # K steps back to back; with N > 1 the all-reduce of step i is in flight while step i+1 computes (every step's tables
    # are complete before the closing barrier)
    t0 = time.perf_counter()
    out = run_steps(args.steps)[-1]
    t_steps = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("PANTAX_PIPE_TRACE"):
        print("[bench] timed region: %.2f ms of steps + %.2f ms in the closing barrier" % (t_steps * 1e3, (dt - t_steps) * 1e3), file=sys.stderr)
    cpu_child_alive = False                         # the child's HiGHS legs wait (it sleeps on a file between its oracle leg and them) ...
    timings = eng.timing_get()
    eng.timing_enable(False)
    eng.timing_filter(None)
    # extra (not `value`): the same step with the unique-trio index REBUILT inside it, on the side stream beside the binning (one-pass rebuild: filed by
    # the groups' row offsets the db's first build learnt) -- the headline of rounds 1-5
    profile_step(eng, species_names, hap_names, avg_len, cfg, comm, shard_max=S_max, rows_max=H_max)
    barrier()
    t1 = time.perf_counter()
    run_steps(args.steps, cfg)
    barrier()
    dt_rebuild = time.perf_counter() - t1
    # extra (not `value`): the rebuild-everything step with the index filed the way a db's FIRST build files it -- visit kernel -> records -> prefix of
    # the groups' counts -> rows kernel (option trio_two_pass) -- instead of the one-pass rebuild, which files by the groups' row offsets the first
    # build learnt (a function of the graphs alone, verified on every build): what that reuse is worth
    dt_two_pass = None
    try:
        eng.set_option("trio_two_pass", "1")
        profile_step(eng, species_names, hap_names, avg_len, cfg, comm, shard_max=S_max, rows_max=H_max)
        barrier()
        t2 = time.perf_counter()
        run_steps(args.steps, cfg)
        barrier()
        dt_two_pass = time.perf_counter() - t2
    finally:
        eng.set_option("trio_two_pass", None)
    profile_step(eng, species_names, hap_names, avg_len, cfg, comm, shard_max=S_max, rows_max=H_max)   # (back on the one-pass rebuild for the legs below)
    barrier()
    leg.go()                                        # ... for this point: both timed regions of the resident step are over (the child's first seconds after
                                                    # the signal -- the largest species' LP through the oracle, SciPy's start-up -- cost the index-resident
                                                    # steps 5 ms each when the signal came in front of them)
    # first-class extra: the same workload from GAF TEXT on disk -- pread + PCIe + device tokenizer (a1) -> resident reads -> one
    # step -> tables.  Never `value` (the contract's value has its inputs resident in HBM).
    l1 = None
    if rank == 0 and not args.no_l1:
        try:
            if world == 1 and spec.get("refdb"):
                # the reference-DB shape: the sample must hold single-strain species (H = 1: the chunk graphs, no LP beyond one column), the species
                # with the most strains / nodes and the one with the most reads
                kept = {r[0] for r in out[0]}
                singles = [s for s in range(len(species)) if species[s].n_paths == 1 and species[s].name in kept][:3]
                big = int(np.argmax([g.n_nodes for g in species]))
                l1 = abundance_l1_leg(eng, ns, species, rd, out, cfg, host_threads, n_sample=40, extra_pick=singles + [big, len(species) - 1])
            elif world == 1:
                l1 = abundance_l1_leg(eng, ns, species, rd, out, cfg, host_threads)
            else:
                # N > 1: rank 0 holds the tables of ALL ranks' species; it generates the whole set once more on the host (the ranks kept only
                # their shares), bins it with the checker's own rule and checks a species sample across the ranks' shards
                from oracle import oracle as orc
                sp_all = ns.graphs()
                rd_all = ns.reads()
                sp_h = orc.par_bin_reads(rd_all.step_off, rd_all.node_id, ns.range_start, ns.range_end, host_threads)
                l1 = abundance_l1_leg(None, ns, sp_all, rd_all, out, cfg, host_threads, sp=sp_h, rc=np.bincount(sp_h[sp_h >= 0], minlength=n_species))
                del rd_all, sp_h
        except Exception as e:   # noqa: BLE001 -- the line is printed regardless
            l1 = {"error": "%s: %s" % (type(e).__name__, e)}
    gaf_extra = None
    seam = None
    n_unique_early = None
    if rank == 0 and world == 1 and not args.no_gaf:
        n_gaf = min(n_reads, args.gaf_reads) if args.gaf_reads else n_reads
        grd = synth.head_reads(rd, n_gaf)
        est = 145 * n_gaf                                   # ~144 bytes of text per short read
        do_seam = not args.no_seam and n_gaf == n_reads
        if do_seam:                                         # + the DB directory: 8 bytes per node and per path step (.bin), 4 (.hipdb)
            est += 12 * (sum(g.n_nodes for g in species) + int(sum(int(g.path_off[-1]) for g in species)))
        td_root = gaf_tmp_dir(est)
        if td_root is None:
            gaf_extra = {"error": "no temporary directory with %.1f GB free for the GAF text" % (est / 1e9)}
        else:
            with tempfile.TemporaryDirectory(dir=td_root) as td:
                gp = os.path.join(td, "reads.gaf")
                t_w = time.perf_counter()
                gaf_bytes = synth.write_gaf_parallel(grd, gp, threads=host_threads)
                write_s = time.perf_counter() - t_w
                eng.load_reads_from_gaf(gp, columns=False)       # warm (allocations, page cache)
                eng.sync()
                # the leg is bound by pread from the page cache into the pinned ring, which varies by tens of per cent from one run to
                # the next on one box: two timed runs, the faster one is reported, both are listed
                runs = []
                for _rep in range(2):
                    t2 = time.perf_counter()
                    eng.trio_index_prefetch()                    # this run's index build (db only) starts beside the transfer of its reads
                    eng.load_reads_from_gaf(gp, columns=False)   # the walks stay in HBM; no host copy of the per-read columns is asked for
                    eng.sync()
                    t_load_r = time.perf_counter() - t2
                    out_gaf = profile_step(eng, species_names, hap_names, avg_len, cfg, LocalComm(), shard_max=S_max, rows_max=H_max)
                    eng.sync()
                    runs.append((time.perf_counter() - t2, t_load_r))
                t_e2e, t_load = min(runs)
                same = (out is not None and out_gaf[0] == out[0] and out_gaf[1] == out[1]) if n_gaf == n_reads else None
                # the box's ceiling for this leg: pinned host -> device copy rate (1 GiB in 64-MB chunks, second pass)
                h2d = None
                try:
                    pin = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
                    dv = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
                    for rep in range(2):
                        torch.cuda.synchronize()
                        t_h = time.perf_counter()
                        for off in range(0, 1 << 30, 64 << 20):
                            dv[off:off + (64 << 20)].copy_(pin[off:off + (64 << 20)], non_blocking=True)
                        torch.cuda.synchronize()
                        h2d = (1 << 30) / (time.perf_counter() - t_h) / 1e9
                    del pin, dv
                except Exception:   # noqa: BLE001
                    pass
                if do_seam:
                    try:
                        # The files -> tables leg is what a `pantax-hip` process does: a fresh process, nothing resident.  It runs in a child (seam_child);
                        # the resident db and reads of the legs above (~110 GB at cfg4) and the blocks this process keeps cached for them are given back
                        # first (a ctx's destruction trims the cache).  Inside this interpreter (torch and its thread pools loaded) the same call took
                        # 0.58-0.60 s against 0.47-0.50 in a fresh process (tools/seam_bench.py).  The step's index size is noted before.
                        n_unique_early = int(eng.trio_nodes_info(fetch=False))
                        eng.close()
                        seam = file_seam_leg(local_rank, species, gp, td, host_threads, out, n_reads, fr=cfg.fr)
                        eng = Engine(local_rank)
                    except Exception as e:   # noqa: BLE001 -- the line is printed regardless
                        seam = {"error": "%s: %s" % (type(e).__name__, e)}
                gaf_extra = {"gaf_bytes": gaf_bytes, "reads": n_gaf, "tokenize_to_resident_ms": t_load * 1e3, "end_to_end_ms": t_e2e * 1e3,
                             "end_to_end_s": t_e2e, "end_to_end_ms_of_both_runs": [r[0] * 1e3 for r in runs], "end_to_end_mreads_per_s": n_gaf / t_e2e / 1e6, "gaf_gb_per_s": gaf_bytes / t_load / 1e9,
                             "tables_equal_to_packed_input_run": same, "gaf_written_in_s": write_s, "gaf_dir": td_root,
                             "pinned_h2d_ceiling_gb_per_s": h2d, "gaf_gb_per_s_of_ceiling": (gaf_bytes / t_load / 1e9 / h2d) if h2d else None,
                             "what": "GAF text on disk (page cache) -> pread + PCIe + device tokenizer -> resident grouped reads -> one step -> tables; the faster of two timed runs (both listed)",
                             "note": None if n_gaf == n_reads else "the first %d reads of the workload as GAF text, against the whole resident db" % n_gaf}
    rd = None
    # extra: the non-trivial LP (pao_hard), timed with every launch bracketed
    hard = None
    if rank == 0 and world == 1 and not args.no_hard:
        hs = hard_spec(args.hard_species)
        hns = synth.NativeSet(hs["seed"], hs["species"], hs["haps"], hs["reads"], hs["genome_len"], present_frac=1.0, threads=host_threads)
        hard_set = hns.make()
        hard_names = [g.name for g in hard_set.species]
        hard_haps = [h for g in hard_set.species for h in g.hap_names]
        hard = dict(workload="%d species x 10 strains, ALL strains present (present_frac 1.0), %d reads, seed %d" % (hs["species"], hs["reads"], hs["seed"]))
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
                    gpu_first_sol_species0={r[1]: r[7] for r in out_h[1] if r[0] == hard_names[0]},
                    kernels_ms_per_step={k: v[1] / n_h for k, v in sorted(kt_h.items(), key=lambda kv: -kv[1][1])[:8]})
        eng_h.close()
        del hard_set, hns
    dt_min = dt_max = dt
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt, -dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max, dt_min = float(t[0].item()), -float(t[1].item())
        dt = dt_max

    if rank == 0:
        species_rows, strain_rows, stats = out
        ms_per_step = dt / args.steps * 1e3
        value = total_reads / (dt / args.steps) / 1e6
        n_lp_rows = int(sum(stats["n_rows"]))
        n_unique = n_unique_early if n_unique_early is not None else int(eng.trio_nodes_info(fetch=False))      # after the timed region: only its size is wanted
        ab, dims = algorithmic_bytes(species, n_lp_rows, n_unique, R_res, T_res, kept={r[0] for r in species_rows} if world == 1 else None,
                                      with_columns={g.name for g, nc in zip(species, stats["n_cand"]) if nc > 0} if world == 1 and len(stats.get("n_cand", [])) == len(species) else None)
        dims["U"] = n_unique
        tr = lambda k, corrected=False: pmc_traffic(k, wkey, spec["name"], corrected) if world == 1 else None

        def ruler(k, launches, tot_ms):
            """one kernel against the HBM roofline: SURVEY 8d's algorithmic bytes / HIP-event time, and -- the other ruler -- the bytes
            the counters saw (this round's committed PMC passes of the same workload; None when there is none)"""
            avg_ms = tot_ms / max(launches, 1)
            per = ab.get(k, 0)
            t_raw = tr(k)
            d = dict(kernel=k, avg_ms=round(avg_ms, 4), launches_timed=launches, algorithmic_bytes=per,
                     achieved=per / (avg_ms * 1e-3) / 1e9, frac=per / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=t_raw,
                     traffic_fetch_x2=tr(k, True), frac_by_counter_bytes=(t_raw / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_raw else None)
            return d
        # dominant kernel by HIP-event time on the library's stream, in the timed steps (the index-resident step runs on ONE stream: nothing shares the
        # device with a bracketed kernel, so a bracket is the kernel's stand-alone time).  `kernels`: the five largest kernels that have a ruler
        # (algorithmic bytes this harness can state), largest first; kernels without one are listed by time in kernels_ms_per_step only.
        roofline = None
        ruled = sorted((k for k in top2 if k in timings and ab.get(k)), key=lambda k: -timings[k][1])
        dom = ruled[0] if ruled else None
        if dom:
            r0 = ruler(dom, *timings[dom])
            roofline = dict(bound="hbm", kernel=dom, achieved=r0["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=r0["frac"], traffic=r0["traffic"],
                            frac_by_counter_bytes=r0["frac_by_counter_bytes"], traffic_fetch_x2=r0["traffic_fetch_x2"], avg_ms=r0["avg_ms"],
                            launches_timed=r0["launches_timed"], algorithmic_bytes=r0["algorithmic_bytes"],
                            traffic_source="profiles/%s_pmc_%s.json (committed --pmc passes, not this run)" % (PMC_ROUND, spec["name"]) if r0["traffic"] else
                                           "no PMC record of this round for this workload: traffic null")
            roofline["kernels"] = []
            for k2 in ruled[:5]:
                r2 = ruler(k2, *timings[k2])
                launches_per_step = timings[k2][0] / max(args.steps, 1)
                roofline["kernels"].append(dict(kernel=k2, avg_ms=r2["avg_ms"], launches_per_step=round(launches_per_step, 2), algorithmic_bytes=r2["algorithmic_bytes"],
                                                achieved=round(r2["achieved"], 1), frac=round(r2["frac"], 4), traffic=r2["traffic"], frac_by_counter_bytes=r2["frac_by_counter_bytes"]))
            if cov_kernel in timings and all(kk["kernel"] != cov_kernel for kk in roofline["kernels"]):
                r3 = ruler(cov_kernel, *timings[cov_kernel])
                roofline["coverage"] = {k: r3[k] for k in ("kernel", "avg_ms", "algorithmic_bytes", "frac", "traffic", "frac_by_counter_bytes")}
            # the step against the sum of its kernels' stand-alone times (warm-up table: every launch bracketed): what launch gaps and host waits cost
            roofline["sum_of_kernels_ms_per_step"] = round(sum(v[1] for v in warm.values()) / max(n_warm_timed, 1), 3)
        # the other large kernels against the same rulers (warm-up table; stretched by what shares the device with them)
        others = {}
        for k, (launches, tot_ms) in warm.items():
            if k in ab and launches:
                r_ = ruler(k, launches, tot_ms)
                others[k] = {kk: r_[kk] for kk in ("avg_ms", "algorithmic_bytes", "frac", "traffic", "frac_by_counter_bytes")}
        # the CPU legs: whatever the child has by now (it is given the rest of its HiGHS time limit)
        cpu_res = leg.finish(timeout=args.highs_full_time_limit + 3 * args.highs_time_limit + 240) if leg.proc is not None else None
        cpu, cpu_detail = None, None
        if cpu_res is not None:
            cpu_detail = cpu_res.get("cpu_baseline") or {"value": None, "unit": "Mreads/s", "cores": 0, "kind": "port", "sample": None}
            if "error" in cpu_res:
                cpu_detail["error"] = cpu_res["error"]
                cpu_detail["traceback"] = cpu_res.get("traceback")
            cpu_detail["parent_waited_s_for_oracle_leg"] = cpu_wait_s
            if hard is not None and cpu_res.get("pao_hard_cpu"):
                hard.update(cpu_res["pao_hard_cpu"])
                o1h, g1h = hard.get("oracle_obj1_species0"), hard.get("gpu_obj1_species0")
                if o1h is not None and g1h is not None:
                    hard["objective_rel_diff_vs_oracle"] = abs(g1h - o1h) / max(1.0, abs(o1h))
                fs, hn = hard.get("oracle_first_sol_species0"), hard.get("hap_names_species0")
                if fs and hn and hard.get("gpu_first_sol_species0"):
                    gfs = hard["gpu_first_sol_species0"]
                    num = sum(abs(gfs.get(h_, 0.0) - (f_ or 0.0)) for h_, f_ in zip(hn, fs))
                    den = sum(abs(f_ or 0.0) for f_ in fs)
                    hard["abundance_l1_vs_oracle_species0"] = num / den if den else 0.0
            hg = cpu_detail.get("highs") or {}
            full = hg.get("full_lp") or {}
            legs = hg.get("legs") or []
            # the line's cpu_baseline: scalars and short strings only (the verbose record goes to the detail file)
            cpu = dict(value=cpu_detail.get("value"), unit="Mreads/s", cores=cpu_detail.get("cores"), kind="port",
                       sample=("first %d of %d reads over all %d species" % (cpu_detail.get("reads") or 0, n_reads, n_species)) if cpu_detail.get("reads") else None,
                       seconds=cpu_detail.get("seconds"), solver="oracle exact LAD (== HiGHS optimum on the fixtures); HiGHS timed beside it",
                       highs_full_lp_rows=hg.get("full_lp_rows"), highs_full_lp_seconds=full.get("highs_seconds"), highs_full_lp_finished=full.get("finished"),
                       highs_10k_rows_seconds=next((l_["highs_seconds"] for l_ in legs if l_.get("rows") == 10000), None),
                       child_alive_during_timed_steps=cpu_child_alive, species_failed=cpu_detail.get("species_failed"), error=cpu_detail.get("error"))
        gx = gaf_extra or {}
        sx = seam or {}
        line = {
            "metric": "PAO wall-time (s) + Mreads/s: packed reads + DB (graphs and their unique-trio index) resident in HBM -> abundance tables; "
                      "the metric's own GAF -> abundance quantity is value_gaf_to_tables (files -> tables, never `value`)",
            "value": value, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": "%s: %d species x %s strains, %d %s, genome %d bp, seed %d, generator %s"
                                   % (spec["name"], n_species, n_haps, n_reads, "long reads N(15000, 3000^2) bp" if spec.get("long_reads") else "short reads (150 bp)",
                                      genome_len, spec["seed"], GENERATOR),
                       "gsteps_per_s": T_res * world / (dt / args.steps) / 1e9,
                       "baseline_config": spec["label"][:60], "set": "per GPU" if args.scaling == "weak" else "one set cut over the ranks",
                       "V": dims["V"], "P": dims["P"], "T": dims["T"], "U": n_unique, "species_per_gpu": S_loc,
                       "strains_total": (int(ns.n_haps.sum()) if spec.get("refdb") else n_species * n_haps * (world if args.scaling == "weak" else 1)), "reads_total": total_reads,
                       "parallelism": "species-shard x%d" % world, "rccl_ranks": ranks_seen if backend == "nccl" else None, "ranks_seen": ranks_seen,
                       "exchange": "none" if world == 1 else ("one rccl all_reduce per step" if backend == "nccl" else backend + " all_reduce (dry run)"),
                       "pao_wall_s": ms_per_step / 1e3,
                       "value_times": "resident step over a resident DB (graphs + the DB's unique-trio index, SURVEY 8f-2): bin -> species decision -> coverage -> filters -> LPs -> tables",
                       "ms_per_step_with_index_rebuild": dt_rebuild / args.steps * 1e3,      # the index rebuilt inside every step (one-pass; rounds 1-5's headline)
                       "ms_per_step_two_pass_rebuild": dt_two_pass / args.steps * 1e3 if dt_two_pass else None,
                       "index_first_build_ms": index_first_build_ms,                        # once per db (fresh process): from the upload's visit table to the filed rows
                       "ms_per_step_ranks_min_max": [dt_min / args.steps * 1e3, dt_max / args.steps * 1e3],
                       # the metric's own wording, GAF text on disk -> tables (never `value`)
                       "from_gaf_text_s": gx.get("end_to_end_s"), "from_gaf_text_mreads_per_s": gx.get("end_to_end_mreads_per_s"),
                       "from_gaf_text_to_resident_s": (gx.get("tokenize_to_resident_ms") or 0) / 1e3 or None, "gaf_gb": (gx.get("gaf_bytes") or 0) / 1e9 or None,
                       "gaf_gb_per_s": gx.get("gaf_gb_per_s"), "pinned_h2d_ceiling_gb_per_s": gx.get("pinned_h2d_ceiling_gb_per_s"),
                       "gaf_gb_per_s_of_ceiling": gx.get("gaf_gb_per_s_of_ceiling"), "tables_equal_to_packed_input_run": gx.get("tables_equal_to_packed_input_run"),
                       "from_gaf_text_error": gx.get("error"),
                       # the reference's own wall-time definition (profile.rs:3326-3327, :3429-3433): profile::profile files -> files, graph loading
                       # included; cold = graphs from the bincode .bin files, warm = from device-ready images; page cache warm (never `value`)
                       "files_to_tables_s": {"cold": sx.get("files_to_tables_cold_s"), "warm": sx.get("files_to_tables_warm_s")},
                       "files_to_tables_mreads_per_s": {"cold": sx.get("mreads_per_s_cold"), "warm": sx.get("mreads_per_s_warm")},
                       "db_load_s": {"cold": sx.get("db_load_cold_s"), "warm": sx.get("db_load_warm_s")}, "seam_gaf_load_s": sx.get("gaf_load_s"),
                       "seam_strain_step_s": sx.get("strain_step_s"), "db_bin_gb": sx.get("db_bin_gb"), "db_image_gb": sx.get("db_image_gb"),
                       "seam_tables_equal_to_resident_step": sx.get("tables_equal_to_resident_step"), "seam_error": sx.get("error"),
                       # north_star's tolerance: strain abundances against the solver-backed PAO (here: the oracle, == HiGHS on the fixtures)
                       "abundance_l1_vs_oracle": (l1 or {}).get("abundance_l1_vs_oracle"), "abundance_l1_species_checked": (l1 or {}).get("species_checked"),
                       "abundance_l1_tolerance": 1e-4, "abundance_l1_error": (l1 or {}).get("error"),
                       "pao_hard_lad_ms_per_species": (hard or {}).get("lad_kernels_ms_per_species"),
                       "pao_hard_objective_rel_diff_vs_oracle": (hard or {}).get("objective_rel_diff_vs_oracle"),
                       "pao_hard_abundance_l1_vs_oracle": (hard or {}).get("abundance_l1_vs_oracle_species0"),
                       "lp_rows_total": n_lp_rows, "n_species_rows": len(species_rows), "n_strain_rows": len(strain_rows),
                       "ingest_route_ms": (ingest_route or {}).get("ms"), "upload_ms_once": upload_ms, "sample_nodes": 0},
            "roofline": roofline,
        }
        # BASELINE's "Mreads/s GAF -> abundance" in its own words: GAF text + DB files on disk (page cache) -> strain_abundance.txt through the C file seam
        # (pantax_hip_profile), DB from device-ready images (warm) / from the reference's .bin containers (cold); pcie_frac = bytes that must cross PCIe /
        # time / the pinned host->device ceiling measured on this box
        if sx.get("files_to_tables_warm_s"):
            h2d = gx.get("pinned_h2d_ceiling_gb_per_s")
            moved_gb = (gx.get("gaf_bytes") or 0) / 1e9 + (sx.get("db_image_gb") or 0)
            line["value_gaf_to_tables"] = {"mreads_per_s": sx.get("mreads_per_s_warm"), "seconds": sx.get("files_to_tables_warm_s"), "db": "images",
                                           "cold_mreads_per_s": sx.get("mreads_per_s_cold"), "cold_seconds": sx.get("files_to_tables_cold_s"),
                                           "bytes_over_pcie_gb": round(moved_gb, 3), "pcie_frac": (moved_gb / sx["files_to_tables_warm_s"] / h2d) if h2d else None,
                                           "pcie_floor_s": (moved_gb / h2d) if h2d else None}
        if cpu is not None:
            line["cpu_baseline"] = cpu
        line["kernels_ms_per_step"] = {k: round(v[1] / max(n_warm_timed, 1), 3) for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])[:14]}
        line["result"] = {"n_species_rows": len(species_rows), "n_strain_rows": len(strain_rows),
                          "top_strains": [(r[0], r[1], round(r[2], 4), round(r[3], 6)) for r in strain_rows[:3]]}
        # everything verbose goes to the side record: the line itself stays well under 8 KB (the driver keeps its tail)
        detail = {"line": line, "from_gaf_text": gaf_extra, "file_seam": seam, "cpu_baseline": cpu_detail, "pao_hard": hard, "abundance_l1": l1, "roofline_other_kernels": others,
                  "kernels_ms_per_step_all": {k: v[1] / max(n_warm_timed, 1) for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])},
                  "kernel_timer_scopes_per_step": int(sum(v[0] for v in warm.values()) / max(n_warm_timed, 1)),
                  "kernels_ms_per_step_source": "warm-up steps (every launch bracketed by HIP events); the timed steps bracket roofline.kernel, its runner-up and the coverage kernel only",
                  "solver": {"iters": stats["iters"][:8], "n_rows": stats["n_rows"][:8], "n_patterns": stats["n_patterns"][:8], "objective": stats["obj"][:8]},
                  "ingest_route": ingest_route, "step_path_primed_ms_once": prime_ms, "synthetic_set_generated_in_s": gen_s,
                  "host": {"cores": os.cpu_count(), "mem_available_gb": _mem_available_gb()}}
        dpath = args.detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail_%s_n%d.json" % (spec["name"], world))
        try:
            os.makedirs(os.path.dirname(dpath), exist_ok=True)
            with open(dpath, "w") as f:
                json.dump(detail, f, indent=1)
            line["detail_file"] = os.path.relpath(dpath, ROOT)
        except OSError as e:
            line["detail_file"] = "not written: %s" % e
        print(json.dumps(line), flush=True)
    if os.environ.get("PANTAX_BENCH_RSS"):     # tools/strong_dry_run.sh: what a rank needs of the host
        import resource
        print("rank %d: peak host RSS %.1f GB" % (rank, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0), file=sys.stderr, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
