#!/usr/bin/env python3
"""File -> resident reads (pread into pinned chunks + PCIe + device tokenizer) of a cfg3-sized GAF; run under different
PANTAX_STAGE_* settings to size the staging.  usage: stage_bw_probe.py [reads]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthdata as synth
from pantax_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p = sys.argv[2] if len(sys.argv) > 2 else os.path.join(tempfile.gettempdir(), "stage_probe_%d.gaf" % n)
sset = synth.make_set(20260503, 1, 10, n, 5_000_000)
if not os.path.exists(p): synth.write_gaf(sset.reads, p)
eng = Engine(0); eng.upload_db(sset.species)
eng.load_reads_from_gaf(p); eng.sync()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.load_reads_from_gaf(p); eng.sync(); ts.append(time.perf_counter() - t0)
sz = os.path.getsize(p)
print("%s: %.1f MB, best %.1f ms = %.1f GB/s (median %.1f ms)" % ({k: v for k, v in os.environ.items() if k.startswith("PANTAX_STAGE")}, sz / 1e6, min(ts) * 1e3, sz / min(ts) / 1e9, sorted(ts)[2] * 1e3))
