#!/usr/bin/env python3
"""Debug/measurement: host (multi-thread mmap) vs device GAF tokenizer on a cfg2-sized GAF (1M reads)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pantax_amd import io as pio
import synthdata as synth
from pantax_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
long_reads = len(sys.argv) > 2 and sys.argv[2] == "long"   # HiFi-shaped lines (~700 node ids each)
sset = synth.make_set(20260503, 1, 10, n, 5_000_000, long_reads=long_reads)
d = tempfile.mkdtemp()
p = os.path.join(d, "x.gaf")
t0 = time.perf_counter(); synth.write_gaf(sset.reads, p); print("wrote %.1f MB in %.1f s" % (os.path.getsize(p) / 1e6, time.perf_counter() - t0))
eng = Engine(0)
for th in (1, 8):
    t0 = time.perf_counter(); h = pio.load_gaf(p, n_threads=th); dt = time.perf_counter() - t0
    print("host tokenizer %d threads: %.1f ms (%.1f Mreads/s, %.2f GB/s)" % (th, dt * 1e3, n / dt / 1e6, os.path.getsize(p) / dt / 1e9))
pio.load_gaf(p, engine=eng)
eng.timing_enable(True); eng.timing_reset()
t0 = time.perf_counter(); g = pio.load_gaf(p, engine=eng); dt = time.perf_counter() - t0
print("device tokenizer end to end (H2D text, kernels, D2H arrays, numpy copies): %.1f ms (%.1f Mreads/s)" % (dt * 1e3, n / dt / 1e6))
print({k: round(v[1], 3) for k, v in eng.timing_get().items()})
for k in h: assert np.array_equal(h[k], g[k]), k
print("identical")
eng.upload_db(sset.species)
eng.timing_enable(False)
t0 = time.perf_counter()
h = pio.load_gaf(p, n_threads=8)
eng.upload_reads(h["step_off"], h["node_id"], h["pstart"], h["pend"], h["qlen"], h["mapq"], h["flags"]); eng.sync()
dt_host = time.perf_counter() - t0
eng.load_reads_from_gaf(p)
t0 = time.perf_counter(); eng.load_reads_from_gaf(p); eng.sync(); dt_dev = time.perf_counter() - t0
print("file -> resident grouped reads: host tokenizer + upload %.1f ms | device tokenizer %.1f ms (%.1f Mreads/s)" % (dt_host * 1e3, dt_dev * 1e3, n / dt_dev / 1e6))
