#!/usr/bin/env python3
"""Debug: where the host time of one bench step goes (C call vs Python around it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pantax_amd import pipeline
import synthdata as synth
from pantax_amd.engine import Engine
from pantax_amd.pipeline import StepConfig, LocalComm
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sset = synth.make_set(20260503, 1, 10, 1_000_000, 5_000_000) if wl == "cfg2" else synth.make_set(20260504, 100, 10, 10_000_000, 5_000_000)
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
names = [g.name for g in sset.species]; haps = [h for g in sset.species for h in g.hap_names]
avg = sset.avg_len(); cfg = StepConfig(); comm = LocalComm()
N = 200 if wl == "cfg2" else 20
for _ in range(5): pipeline.profile_step(eng, names, haps, avg, cfg, comm)
t0 = time.perf_counter()
for _ in range(N): pipeline.profile_step(eng, names, haps, avg, cfg, comm)
t_all = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N): eng.profile_step(avg)
t_c = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N): loc = pipeline.local_stage(eng, avg, cfg)
t_loc = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N): pipeline.finalize_stage(loc, names, haps, cfg, comm, None)
t_fin = (time.perf_counter() - t0) / N
print("profile_step %.1f us | engine.profile_step (C call + ctypes) %.1f | local_stage %.1f | finalize_stage %.1f" % (t_all * 1e6, t_c * 1e6, t_loc * 1e6, t_fin * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(N): pipeline.profile_step(eng, names, haps, avg, cfg, comm)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
