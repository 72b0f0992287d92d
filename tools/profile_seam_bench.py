#!/usr/bin/env python3
"""Measurement: wall time of the FILE seam (pantax_hip_profile: DB files + GAF text in, abundance tables out) on the
cfg2 workload, second call (allocations warm), graphs from .bin."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthdata as synth
from pantax_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sset = synth.make_set(20260503, S, 10, n, 5_000_000)
root = tempfile.mkdtemp()
db = os.path.join(root, "db"); os.mkdir(db)
t0 = time.perf_counter(); synth.write_db(sset, db); gaf = os.path.join(root, "gfa_mapped.gaf"); synth.write_gaf(sset.reads, gaf)
print("wrote db + gaf (%.0f MB) in %.1f s" % (os.path.getsize(gaf) / 1e6, time.perf_counter() - t0))
eng = Engine(0)
IC = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 1: write graph images in call 0, use them afterwards
for i in range(3):
    wd = os.path.join(root, "wd%d" % i); os.mkdir(wd)
    cwd = os.getcwd(); os.chdir(wd)
    t0 = time.perf_counter()
    eng.profile(db, wd, gaf, zip="serialize", sample_nodes=500000, image_cache=(2 if i == 0 else 1) if IC else 0)
    dt = time.perf_counter() - t0
    os.chdir(cwd)
    print("pantax_hip_profile call %d: %.1f ms (%.1f Mreads/s from files to files)" % (i, dt * 1e3, n / dt / 1e6))
print(open(os.path.join(wd, "strain_abundance.txt")).read()[:400])
