#!/usr/bin/env python3
"""Measurement: pantax_hip_gaf_filter on a long-read shaped GAF (paths of ~600 node ids per line) against
the oracle's single-threaded C restatement on the same text."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import make_longread_gaf
from oracle import oracle as orc
from pantax_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = time.perf_counter()
txt = make_longread_gaf(11, n, path_ids=1200)
print("generated %d lines, %.1f MB in %.1f s" % (txt.count(b"\n"), len(txt) / 1e6, time.perf_counter() - t0))
td = tempfile.mkdtemp()
gp = os.path.join(td, "gfa_mapped.gaf")
open(gp, "wb").write(txt)
eng = Engine(0)
for i in range(3):
    t0 = time.perf_counter()
    r = eng.gaf_filter(gp)
    dt = time.perf_counter() - t0
    print("device filter call %d: %.1f ms (%.2f GB/s of text, files to file) -> %s" % (i, dt * 1e3, len(txt) / dt / 1e9, r))
eng.timing_enable(True); eng.timing_reset()
eng.gaf_filter(gp)
for name, (launches, ms) in sorted(eng.timing_get().items(), key=lambda kv: -kv[1][1])[:8]:
    print("  %-24s %4d launches %8.3f ms" % (name, launches, ms))
t0 = time.perf_counter()
keep, nrec = orc.gaf_filter(txt)
dt = time.perf_counter() - t0
print("oracle (1 thread, text in memory, no file output): %.1f ms (%.2f GB/s) -> %d records, %d kept" % (dt * 1e3, len(txt) / dt / 1e9, nrec, keep.sum()))
