#!/usr/bin/env python3
"""Debug: per-step wall time of consecutive resident steps (does anything accumulate?).  usage: step_growth.py S R L [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthdata as synth
from pantax_amd.engine import Engine
S, R, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
sset = synth.make_set(20260504, S, 10, R, L)
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
avg = sset.avg_len()
for rebuild in (True, False, True):
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); eng.profile_step(avg, rebuild_trio=rebuild); ts.append((time.perf_counter() - t0) * 1e3)
    print("rebuild=%d:" % rebuild, " ".join("%.2f" % t for t in ts))
eng.timing_enable(True); eng.timing_reset()
for i in range(5): eng.profile_step(avg)
print({k: round(v[1] / 5, 3) for k, v in sorted(eng.timing_get().items(), key=lambda kv: -kv[1][1])[:12]})
