#!/usr/bin/env python3
"""Debug: wall-clock of each host call of one profile step (cfg2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import synthdata as synth
from pantax_amd.engine import Engine
from pantax_amd.pipeline import StepConfig
sset = synth.make_set(20260503, 1, 10, 1_000_000, 5_000_000)
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
cfg = StepConfig(); avg = sset.avg_len()
acc = {}
def T(name, f):
    eng.sync(); t = time.perf_counter(); r = f(); eng.sync(); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
N = 20
for it in range(N + 3):
    if it == 3: acc.clear()
    _, rc, bs, lm, uq = T("rcls_profile", lambda: eng.rcls_profile(want_species=False))
    keep, absolute, _ = T("species_profiling", lambda: eng.species_profiling((rc, bs, lm, uq), avg))
    T("db_reset+trio", lambda: (eng.db_reset(), eng.trio_nodes_info(fetch=False)))
    T("coverage", lambda: eng.get_node_abundances(species_active=keep, fetch=False))
    met, info = T("strain_profiling", lambda: eng.strain_profiling(absolute, species_active=keep))
    T("abundance_filter", lambda: eng.abundance_filter(met, np.ones(eng.S, dtype=np.uint8)))
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / N * 1e3, 3))
