#!/usr/bin/env python3
"""Where a GAF load spends its time (PANTAX_HIP_TRACE laps of stage_gaf.hip on stderr): cfg3-sized text, second load.
usage: gaf_load_trace.py [n_reads] [n_species]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantax_amd import synth
from pantax_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sset = synth.cached_set(20260504, S, 10, n, 5_000_000) if os.environ.get("PANTAX_SYNTH_CACHE") else synth.make_set(20260504, S, 10, n, 5_000_000)
d = tempfile.mkdtemp()
p = os.path.join(d, "x.gaf")
synth.write_gaf(sset.reads, p)
print("GAF %.1f MB" % (os.path.getsize(p) / 1e6), flush=True)
eng = Engine(0)
os.environ.pop("PANTAX_HIP_TRACE", None)
eng.load_reads_from_gaf(p); eng.sync()
os.environ["PANTAX_HIP_TRACE"] = "1"
for _ in range(2):
    t0 = time.perf_counter(); eng.load_reads_from_gaf(p); eng.sync(); dt = time.perf_counter() - t0
    print("load: %.1f ms = %.2f GB/s, %.1f Mreads/s" % (dt * 1e3, os.path.getsize(p) / dt / 1e9, n / dt / 1e6), flush=True)
eng.close()
