#!/usr/bin/env python3
"""Where a GAF load spends its time (PANTAX_HIP_TRACE laps of stage_gaf.hip on stderr; the laps synchronise the stream, so
the traced load is slower than the plain one printed first): a bench workload as text, second load.
usage: gaf_load_trace.py [workload] [dir]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import native_set, workload_spec
import synthdata as synth
from pantax_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
ns = native_set(workload_spec(wl))
rd = ns.reads()
n = rd.n_reads
d = tempfile.mkdtemp(dir=sys.argv[2] if len(sys.argv) > 2 else "/tmp")
p = os.path.join(d, "x.gaf")
synth.write_gaf_parallel(rd, p)
print("GAF %.1f MB" % (os.path.getsize(p) / 1e6), flush=True)
eng = Engine(0)
os.environ.pop("PANTAX_HIP_TRACE", None)
eng.load_reads_from_gaf(p); eng.sync()
for _ in range(3):
    t0 = time.perf_counter(); eng.load_reads_from_gaf(p, columns=False); eng.sync(); dt = time.perf_counter() - t0
    print("plain load: %.1f ms = %.2f GB/s, %.1f Mreads/s" % (dt * 1e3, os.path.getsize(p) / dt / 1e9, n / dt / 1e6), flush=True)
os.environ["PANTAX_HIP_TRACE"] = "1"
for _ in range(1):
    t0 = time.perf_counter(); eng.load_reads_from_gaf(p); eng.sync(); dt = time.perf_counter() - t0
    print("load: %.1f ms = %.2f GB/s, %.1f Mreads/s" % (dt * 1e3, os.path.getsize(p) / dt / 1e9, n / dt / 1e6), flush=True)
eng.close()
os.unlink(p); os.rmdir(d)
