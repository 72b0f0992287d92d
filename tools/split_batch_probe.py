#!/usr/bin/env python3
"""Probe: one resident batch of S species vs the same species split over two ctx stepped from two host threads.
usage: split_batch_probe.py [n_species] [reads]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import synthdata as synth
from pantax_amd.engine import Engine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
def make(seed, s, r):
    return synth.make_set(seed, s, 10, r, 5_000_000)
def timeit(engs, sets, n=10):
    for e, ss in zip(engs, sets):
        e.profile_step(ss.avg_len())
    for e in engs: e.sync()
    def run(e, ss):
        for _ in range(n): e.profile_step(ss.avg_len())
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(e, ss)) for e, ss in zip(engs, sets)]
    for t in ths: t.start()
    for t in ths: t.join()
    for e in engs: e.sync()
    return (time.perf_counter() - t0) / n * 1e3
whole = make(1, S, R)
e0 = Engine(0); e0.upload_db(whole.species); e0.upload_packed(whole.reads)
print("one batch of %d species: %.3f ms/step" % (S, timeit([e0], [whole])))
e0.close()
halves = [make(2, S // 2, R // 2), make(3, S - S // 2, R - R // 2)]
engs = []
for h in halves:
    e = Engine(0); e.upload_db(h.species); e.upload_packed(h.reads); engs.append(e)
print("two half batches, one after the other: %.3f ms" % (timeit(engs[:1], halves[:1]) + timeit(engs[1:], halves[1:])))
print("two half batches in flight together:   %.3f ms" % timeit(engs, halves))
