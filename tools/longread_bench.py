#!/usr/bin/env python3
"""Measurement: the step on a cfg5-shaped workload scaled to one GPU (HiFi-like reads, mean 15 kb => ~500 graph steps per
read; many strains per species).  Prints the step time and the per-kernel table.
usage: longread_bench.py [n_species] [strains] [n_reads] [genome_len]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import synthdata as synth
from pantax_amd.engine import Engine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H = int(sys.argv[2]) if len(sys.argv) > 2 else 50
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 5_000_000
t0 = time.perf_counter()
sset = synth.make_set(20260505, S, H, R, L, long_reads=True)
rd = sset.reads
print("generated %d species x %d strains, %d reads, %d steps (%.0f per read) in %.1f s" % (S, H, rd.n_reads, len(rd.node_id), len(rd.node_id) / rd.n_reads, time.perf_counter() - t0))
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(rd)
avg = sset.avg_len()
for it in range(3):
    eng.profile_step(avg, fr=0.5)
eng.sync()
N = 10
t0 = time.perf_counter()
for it in range(N):
    out = eng.profile_step(avg, fr=0.5)
eng.sync()
dt = (time.perf_counter() - t0) / N
print("step %.3f ms  (%.2f Mreads/s, %.2f Gsteps/s)" % (dt * 1e3, rd.n_reads / dt / 1e6, len(rd.node_id) / dt / 1e9))
eng.timing_enable(True); eng.timing_reset()
for it in range(3):
    eng.profile_step(avg, fr=0.5)
eng.sync()
rows = eng.timing_get()
for name, (launches, ms) in sorted(rows.items(), key=lambda r: -r[1][1])[:14]:
    print("  %-28s %6d launches %9.3f ms/step" % (name, launches // 3, ms / 3))
info = out[3]
print("per species (columns, patterns, pivots LP1, pivots LP2):", [(info[s].n_candidates, info[s].n_patterns, info[s].iters1, info[s].iters2) for s in range(min(eng.S, 6))])
print("species solved:", sum(1 for s in range(eng.S) if info[s].n_candidates > 0 and info[s].status1 == 0), "of", eng.S,
      " max candidates", max(info[s].n_candidates for s in range(eng.S)))
