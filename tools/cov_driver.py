#!/usr/bin/env python3
"""Minimal driver for profiling: uploads the cfg2 workload once and runs the coverage stage N times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthdata as synth
from pantax_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sset = synth.make_set(20260503, 1, 10, 1_000_000, 5_000_000)
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
eng.rcls_profile(want_species=False); eng.trio_nodes_info(fetch=False)
for _ in range(n):
    eng.get_node_abundances(fetch=False)
eng.close()
