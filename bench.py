#!/usr/bin/env python3
"""bench.py -- PAO wall time / Mreads/s of the profiling hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: read binning + species counters, species profile, unique-trio index (rebuilt per
step like the reference does per run), node-coverage histogram, LP row grouping, the two PAO
solves, filters and the abundance table.  Workload at N=1 = BASELINE.json configs[1]
("Single-species E. coli, 10 strains, 1M synthetic short-read GAF"); with N ranks each rank owns
its own species shard of that shape (weak scaling; species are independent sub-problems) and one
RCCL all-reduce carries the normalisers.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
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


def algorithmic_bytes(sset, U):
    """SURVEY.md section 8d per-stage compulsory traffic for ONE step of this rank's workload."""
    rd = sset.reads
    R, T = rd.n_reads, len(rd.node_id)
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
        # a7 bucket scatter: 4P in + 16 B record per window out
        "trio_fill_kernel": 4 * P + 16 * max(P - 2 * H, 0),
        # a10: 4P in + 8V mask out
        "mask_kernel": 4 * P + 8 * V,
    }, dict(R=R, T=T, V=V, L=L, P=P, H=H, U=U)


def cpu_baseline(sset, sample_reads, cfg):
    """The oracle (plain-C port of the reference algorithm) on ONE host core over a bounded sample:
    the first `sample_reads` reads of the same workload through binning, trio index, coverage,
    filters and both LP solves.  Reported beside the GPU number; it is a baseline, not the target."""
    from oracle import oracle as orc
    from tests.helpers import select_reads
    rd = sset.reads
    n = min(sample_reads, rd.n_reads)
    t0 = time.perf_counter()
    step_off = rd.step_off[: n + 1]
    node_id = rd.node_id[: int(step_off[-1])]
    sp = orc.bin_reads(step_off, node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species])
    counts = orc.species_counts(sp, rd.qlen[:n], rd.mapq[:n], len(sset.species))
    keep, absolute, _ = orc.species_profile(sp, rd.qlen[:n], counts, sset.avg_len())
    t_bin = time.perf_counter() - t0
    t_lp = t_trio = t_cov = 0.0
    for si, g in enumerate(sset.species):
        if not keep[si]:
            continue
        t1 = time.perf_counter()
        G = orc.Graph(g.node_len, g.path_off, g.path_nodes)
        T = orc.TrioTable(G)
        t_trio += time.perf_counter() - t1
        t1 = time.perf_counter()
        sel = np.nonzero(sp == si)[0]
        so = np.zeros(len(sel) + 1, dtype=np.uint64)
        ns = (step_off[1:] - step_off[:-1]).astype(np.int64)[sel]
        so[1:] = np.cumsum(ns)
        starts = step_off[:-1].astype(np.int64)[sel]
        idx = np.repeat(starts, ns) + (np.arange(int(ns.sum())) - np.repeat(so[:-1].astype(np.int64), ns))
        b, c, tb, _ = orc.node_coverage(G, T, g.range_start, so, node_id[idx], rd.pstart[:n][sel], rd.pend[:n][sel])
        t_cov += time.perf_counter() - t1
        t1 = time.perf_counter()
        rc, met, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=cfg.fr, fc=cfg.fc, sr=cfg.sr)
        t_lp += time.perf_counter() - t1
        orc.abundance_constraint(absolute[si], met)
    dt = time.perf_counter() - t0
    return dict(value=n / dt / 1e6, unit="Mreads/s", cores=1, kind="port",
                sample="first %d reads of the same workload (all %d species), oracle bin+trio+coverage+filters+2 LP solves; "
                       "%.2f s total, %.2f s of it in the exact LAD solves" % (n, len(sset.species), dt, t_lp),
                seconds=dt,
                phases_s={"binning+species_profile": t_bin, "trio_index": t_trio, "node_coverage (incl. read selection)": t_cov,
                          "filters+LP solves": t_lp})


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE,
    separate passes, KB units -> bytes; profiles/r01_pmc_coverage.json).  Only valid for the default
    workload the passes were collected on; None otherwise.  See the file for the gfx950 FETCH_SIZE caveat."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_coverage.json")))
        w = d["workload"]
        if (w["reads"], w["species"], w["haps"], w["genome_len"]) != (args.reads, args.species, args.haps, args.genome_len):
            return None
        return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def highs_probe(sset, cfg, max_rows=20000):
    """Optional: time SciPy's bundled HiGHS on the species-0 LP restricted to max_rows covered nodes
    (HiGHS is the reference's open solver, profile.rs:2689-2882; the full LP does not finish in
    minutes)."""
    try:
        from scipy import sparse
        from scipy.optimize import linprog
    except Exception:
        return None
    return None  # filled in by tools/highs_probe.py when run by hand; kept out of the default path (minutes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--species", type=int, default=1)
    ap.add_argument("--haps", type=int, default=10)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gaf", action="store_true", help="skip the extra measurements (from GAF text, two passes in flight)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    from pantax_amd import synth
    from pantax_amd.engine import Engine
    from pantax_amd.pipeline import LocalComm, StepConfig, TorchComm, profile_step, profile_steps_pipelined

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

    # deterministic synthetic shard of this rank (seed = 20260501 + cfg index 2, + rank)
    seed = 20260501 + 2 + 1000 * rank
    sset = synth.make_set(seed, args.species, args.haps, args.reads, args.genome_len)
    for i, g in enumerate(sset.species):
        g.name = "%d" % (100000 * rank + 1000 + i)
    species_names = [g.name for g in sset.species]
    hap_names = [hn for g in sset.species for hn in g.hap_names]
    avg_len = sset.avg_len()
    cfg = StepConfig()

    eng = Engine(local_rank)
    t_up = time.perf_counter()
    eng.upload_db(sset.species)
    eng.upload_packed(sset.reads)          # inputs resident in HBM before the timed region
    eng.sync()
    upload_ms = (time.perf_counter() - t_up) * 1e3   # host->device of packed reads + graph (pageable memory, incl. numpy packing)

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
    for i in range(args.warmup):
        if i == 1:
            eng.timing_reset()      # the very first step also allocates: its launches are not representative
            n_warm_timed = 0
        out = profile_step(eng, species_names, hap_names, avg_len, cfg, comm, shard_max=args.species, rows_max=args.species * args.haps)
        n_warm_timed += 1
    warm = eng.timing_get() if args.warmup else {}
    dom = max(warm.items(), key=lambda kv: kv[1][1])[0] if warm else "coverage_step_kernel"
    eng.timing_filter(dom)
    eng.timing_reset()
    barrier()
    # K steps back to back; with N > 1 the all-reduce of step i is in flight while step i+1 computes (every step's tables
    # are complete before the closing barrier)
    t0 = time.perf_counter()
    out = profile_steps_pipelined(eng, species_names, hap_names, avg_len, args.steps, cfg, comm, shard_max=args.species,
                                  rows_max=args.species * args.haps)[-1]
    barrier()
    dt = time.perf_counter() - t0
    timings = eng.timing_get()
    eng.timing_enable(False)
    eng.timing_filter(None)
    # extra (not `value`): the same step when the unique-trio index, which depends on the DB only, stays
    # resident between steps instead of being rebuilt like the reference does on every run
    cfg_cached = StepConfig(rebuild_trio=False)
    profile_step(eng, species_names, hap_names, avg_len, cfg_cached, comm, shard_max=args.species, rows_max=args.species * args.haps)
    barrier()
    t1 = time.perf_counter()
    profile_steps_pipelined(eng, species_names, hap_names, avg_len, args.steps, cfg_cached, comm, shard_max=args.species,
                            rows_max=args.species * args.haps)
    barrier()
    dt_cached = time.perf_counter() - t1
    # extra (not `value`): two independent passes in flight -- a second ctx with its own copy of the DB and the reads steps on a
    # second host thread (a stream of samples processed two at a time).  One pass is a chain of ~100 short dependent
    # kernels that cannot fill 256 CUs; two chains interleave on the device.
    two_in_flight = None
    if world == 1 and not args.no_gaf:
        import threading
        eng2 = Engine(local_rank)
        eng2.upload_db(sset.species)
        eng2.upload_packed(sset.reads)
        eng2.sync()
        profile_step(eng2, species_names, hap_names, avg_len, cfg)
        outs2 = [None, None]

        def run(e, slot):
            outs2[slot] = profile_steps_pipelined(e, species_names, hap_names, avg_len, args.steps, cfg, LocalComm())[-1]
        eng.sync(); eng2.sync()
        t3 = time.perf_counter()
        ths = [threading.Thread(target=run, args=(eng, 0)), threading.Thread(target=run, args=(eng2, 1))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        eng.sync(); eng2.sync()
        dt2 = time.perf_counter() - t3
        two_in_flight = {"steps": 2 * args.steps, "ms_per_step": dt2 / (2 * args.steps) * 1e3, "mreads_per_s": 2 * args.steps * args.reads / dt2 / 1e6,
                         "tables_equal": bool(outs2[0][:2] == outs2[1][:2] == out[:2])}
        eng2.close()
    # extra (not `value`): the same workload from GAF TEXT on disk -- device tokenizer (a1) -> resident reads -> one step
    gaf_extra = None
    if rank == 0 and not args.no_gaf:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            gp = os.path.join(td, "reads.gaf")
            synth.write_gaf(sset.reads, gp)
            eng.load_reads_from_gaf(gp)                      # warm (allocations)
            eng.sync()
            t2 = time.perf_counter()
            eng.load_reads_from_gaf(gp)
            eng.sync()
            t_load = time.perf_counter() - t2
            out_gaf = profile_step(eng, species_names, hap_names, avg_len, cfg, comm if world == 1 else LocalComm(), shard_max=args.species, rows_max=args.species * args.haps)
            eng.sync()
            t_e2e = time.perf_counter() - t2
            same = out is not None and out_gaf[0] == out[0] and out_gaf[1] == out[1]
            gaf_extra = {"gaf_bytes": os.path.getsize(gp), "tokenize_to_resident_ms": t_load * 1e3, "end_to_end_ms": t_e2e * 1e3,
                         "end_to_end_mreads_per_s": args.reads / t_e2e / 1e6, "tables_equal_to_packed_input_run": bool(same)}
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        species_rows, strain_rows, stats = out
        ms_per_step = dt / args.steps * 1e3
        total_reads = args.reads * world
        value = total_reads / (dt / args.steps) / 1e6
        ab, dims = algorithmic_bytes(sset, eng.U or 0)
        n_lp_rows = int(sum(stats["n_rows"]))
        ab["sort_hist_kernel"] = 8 * n_lp_rows             # one key word in
        ab["sort_scatter_kernel"] = 2 * 24 * n_lp_rows     # three key words in, three out
        # dominant kernel by HIP-event time on the library's stream
        kt = {k: v for k, v in timings.items()}
        roofline = None
        if dom and dom in kt:
            launches, tot_ms = kt[dom]
            avg_ms = tot_ms / max(launches, 1)
            bytes_per_launch = ab.get(dom)
            if bytes_per_launch is not None:
                per = bytes_per_launch
                ach = per / (avg_ms * 1e-3) / 1e9
                roofline = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                traffic=pmc_traffic(dom, args), avg_ms=avg_ms, algorithmic_bytes=per)
            else:
                roofline = dict(bound="hbm", kernel=dom, achieved=0.0, peak=HBM_PEAK_GBS, unit="GB/s", frac=0.0, traffic=None,
                                avg_ms=avg_ms, algorithmic_bytes=0,
                                note="no streaming-traffic model for this launch (latency-bound: the small-LP solver does "
                                     "O(#patterns*log n) searches per pivot)")
        line = {
            "metric": "PAO wall-time (s) + Mreads/s GAF->abundance (packed reads resident in HBM)",
            "value": value, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "pao_wall_s": ms_per_step / 1e3, "upload_ms_once": upload_ms,
            "ms_per_step_trio_index_resident": dt_cached / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": "cfg2: single-species E. coli-like, %d strains, %d short reads (150 bp) per GPU, "
                                   "genome %d bp, V=%d nodes, T=%d steps" % (args.haps, args.reads, args.genome_len, dims["V"], dims["T"]),
                       "species_per_gpu": args.species, "parallelism": "species-shard x%d" % world, "sample_nodes": 0,
                       "exchange": "none" if world == 1 else ("one rccl all_reduce per step, in flight during the next step" if backend == "nccl" else backend + " all_reduce (dry run)")},
            "from_gaf_text": gaf_extra,
            "two_passes_in_flight": two_in_flight,
            "roofline": roofline,
            "kernels_ms_per_step": {k: v[1] / max(n_warm_timed, 1) for k, v in sorted(warm.items(), key=lambda kv: -kv[1][1])},
            "kernels_ms_per_step_source": "warm-up steps (every launch bracketed by HIP events); the timed steps bracket roofline.kernel only",
            "solver": {"iters": stats["iters"][:4], "n_rows": stats["n_rows"][:4], "n_patterns": stats["n_patterns"][:4],
                       "objective": stats["obj"][:4]},
            "result": {"n_species_rows": len(species_rows), "n_strain_rows": len(strain_rows),
                       "top_strains": [(r[0], r[1], round(r[2], 4), round(r[3], 6)) for r in strain_rows[:3]]},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sset, args.cpu_sample, cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()          # rank 0 is still timing the CPU baseline: nobody tears the communicator down before it is done
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
