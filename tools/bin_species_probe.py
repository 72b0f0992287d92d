#!/usr/bin/env python3
"""Binning kernel against the number of species (small genomes, so only S and the reads matter): time of
bin_reads_kernel for S in a list, same reads per run.  usage: bin_species_probe.py [reads] [S ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import synthdata as synth
from pantax_amd.engine import Engine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Ss = [int(x) for x in sys.argv[2:]] or [100, 400, 1000]
eng = Engine(0)
for S in Ss:
    sset = synth.make_set(77, S, 2, R, 40000, adversarial_frac=0.0)
    eng.upload_db(sset.species); eng.upload_packed(sset.reads)
    eng.rcls_profile(want_species=False)
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(5): eng.rcls_profile(want_species=False)
    t = eng.timing_get(); eng.timing_enable(False)
    l, ms = t["bin_reads_kernel"]
    print("S=%5d R=%d: bin_reads_kernel %.3f ms (%.2f ns per read)" % (S, R, ms / l, 1e6 * ms / l / R))
