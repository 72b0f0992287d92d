#!/usr/bin/env python3
"""Probe of the wide LP path at size: species of 200 strains over a 1 Mbp genome, most strains present, so that the first
filter keeps 100+ columns per species; prints the step time, the LP sizes / pivots / status, the kernel table, and compares
the objectives of the first species with the oracle.  usage: wide_columns_probe.py [S] [H] [genome_len] [reads]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synthdata as synth
from pantax_amd.engine import Engine
from tests.helpers import select_reads
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
R = int(sys.argv[4]) if len(sys.argv) > 4 else 3_000_000
t = time.perf_counter()
sset = synth.make_set(98, S, H, R, L, present_frac=0.6)
print("set: %.1f s, nodes %s" % (time.perf_counter() - t, [g.n_nodes for g in sset.species]))
eng = Engine(0); eng.upload_db(sset.species); eng.upload_packed(sset.reads)
avg = sset.avg_len()
for _ in range(2): out = eng.profile_step(avg, fr=0.05)
eng.sync(); t = time.perf_counter()
for _ in range(3): out = eng.profile_step(avg, fr=0.05)
eng.sync(); print("step %.2f ms" % ((time.perf_counter() - t) / 3 * 1e3))
info = out[3]
print([(info[s].n_candidates, info[s].n_patterns, info[s].n_rows, info[s].iters1, info[s].iters2, info[s].status1, info[s].status2) for s in range(eng.S)])
eng.timing_enable(True); eng.timing_reset(); eng.profile_step(avg, fr=0.05); eng.sync()
for name, (l, ms) in sorted(eng.timing_get().items(), key=lambda kv: -kv[1][1])[:6]: print("  %-26s %3d %8.3f ms" % (name, l, ms))
from oracle import oracle as orc
sp = eng.rcls_profile()[0]
g = sset.species[0]
G = orc.Graph(g.node_len, g.path_off, g.path_nodes); T = orc.TrioTable(G)
so, nid, ps, pe = select_reads(sset.reads, np.nonzero(sp == 0)[0])
b, c, tb, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
t = time.perf_counter()
rc, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, tb, fr=0.05)
print("oracle species 0: rc %d, %d candidates, obj %.12g / %.12g in %.1f s; device obj %.12g / %.12g" % (rc, nc, o1, o2, time.perf_counter() - t, info[0].obj1, info[0].obj2))
