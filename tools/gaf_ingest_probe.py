#!/usr/bin/env python3
"""Measurement: GAF text on disk -> resident grouped reads (pantax_hip_reads_load_gaf) at a bench workload's size, under option settings of the staged
upload, alternating on one box.  usage: gaf_ingest_probe.py [workload=cfg4] [reps=3] [option=value,option=value ...]   e.g.  stage_ch_mb=128  stage_ch_mb=256,stage_threads=48"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import synthdata as synth
from pantax_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = [""] + sys.argv[3:]
spec = bench.workload_spec(wl)
threads = max(1, min(64, os.cpu_count() or 1))
ns = bench.native_set(spec, threads=threads)
rd = ns.reads()
root = bench.gaf_tmp_dir(145 * spec["reads"])
assert root
with tempfile.TemporaryDirectory(dir=root) as td:
    gp = os.path.join(td, "reads.gaf")
    nb = synth.write_gaf_parallel(rd, gp, threads=threads)
    del rd
    eng = Engine(0)
    eng.upload_db(ns.graphs())
    eng.load_reads_from_gaf(gp, columns=False); eng.sync()      # page cache, pinned ring, work buffers
    best = {v: [] for v in variants}
    for _ in range(reps):
        for var in variants:
            sets = [kv.split("=") for kv in var.split(",") if kv]
            for k, v in sets:
                eng.set_option(k, v)
            t0 = time.perf_counter()
            eng.load_reads_from_gaf(gp, columns=False); eng.sync()
            best[var].append(time.perf_counter() - t0)
            for k, v in sets:
                eng.set_option(k, None)
    # one more run with every launch clocked: the tokenizer's kernels, per run
    eng.timing_enable(True); eng.timing_reset()
    eng.load_reads_from_gaf(gp, columns=False); eng.sync()
    print("kernels (ms per run):", {k: round(ms, 3) for k, (n, ms) in sorted(eng.timing_get().items(), key=lambda kv: -kv[1][1])[:8]})
    eng.timing_enable(False)
    for var in variants:
        t = min(best[var])
        print("%-40s %.1f ms = %.1f GB/s (runs: %s)" % (var or "defaults", t * 1e3, nb / t / 1e9, " ".join("%.1f" % (x * 1e3) for x in best[var])))
    eng.close()
