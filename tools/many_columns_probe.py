#!/usr/bin/env python3
"""Probe: the strain step when nearly every haplotype is a candidate (4 species x 40 strains, 36 LP columns and 139
membership patterns per species): prints the step time, the LP sizes / pivots and the kernel table."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synthdata as synth
from pantax_amd.engine import Engine
sset = synth.make_set(99, 4, 40, 2_000_000, 2_000_000, present_frac=0.9)
eng = Engine(0); eng.upload_db(sset.species); eng.upload_packed(sset.reads)
avg = sset.avg_len()
for _ in range(2): out = eng.profile_step(avg, fr=0.05)
eng.sync(); t=time.perf_counter()
for _ in range(5): out = eng.profile_step(avg, fr=0.05)
eng.sync(); print("step %.2f ms" % ((time.perf_counter()-t)/5*1e3))
info = out[3]
print([(info[s].n_candidates, info[s].n_patterns, info[s].iters1, info[s].iters2, info[s].status1) for s in range(eng.S)])
eng.timing_enable(True); eng.timing_reset(); eng.profile_step(avg, fr=0.05); eng.sync()
for name, (l, ms) in sorted(eng.timing_get().items(), key=lambda kv: -kv[1][1])[:6]: print("  %-26s %3d %8.3f ms" % (name, l, ms))
