#!/usr/bin/env python3
"""One-off check at cfg3 size (100 species x 10 strains, 10 M reads): integer outputs bit for bit and LP objectives to
1e-9 against the oracle, species by species.  usage: cfg3_parity.py [n_species] [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as orc
import synthdata as synth
from pantax_amd.engine import Engine
from tests.helpers import select_reads
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
t0 = time.perf_counter()
sset = synth.make_set(20260504, S, 10, R, 5_000_000)
rd = sset.reads
print("generated in %.1f s: V=%d T=%d" % (time.perf_counter() - t0, sum(g.n_nodes for g in sset.species), len(rd.node_id)))
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(rd)
sp, rc, bs, lm, uq = eng.rcls_profile()
ref_sp = orc.bin_reads(rd.step_off, rd.node_id, [g.range_start for g in sset.species], [g.range_end for g in sset.species])
assert np.array_equal(sp, ref_sp), "binning"
keep, absolute, _ = eng.species_profiling((rc, bs, lm, uq), sset.avg_len())
eng.trio_nodes_info(fetch=False)
hto = None
bases, cov, tb, nab = eng.get_node_abundances()
met, info = eng.strain_profiling(absolute, species_active=keep)
abc, hap, ln, hto = eng.trio_nodes_info()
nb = np.cumsum([0] + [g.n_nodes for g in sset.species]); hb = np.cumsum([0] + [g.n_paths for g in sset.species])
order = np.argsort(sp, kind="stable"); cnt = np.bincount(sp[sp >= 0], minlength=S); first = np.searchsorted(sp[order], np.arange(S))
bad = 0
t0 = time.perf_counter()
for s, g in enumerate(sset.species):
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes); T = orc.TrioTable(G)
    sel = np.sort(order[first[s]:first[s] + cnt[s]])
    so, nid, ps, pe = select_reads(rd, sel)
    b, c, t, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
    u0, u1 = int(hto[hb[s]]), int(hto[hb[s + 1]])
    ok = np.array_equal(bases[nb[s]:nb[s + 1]], b) and np.array_equal(cov[nb[s]:nb[s + 1]], c) and np.array_equal(tb[u0:u1], t) and u1 - u0 == T.n_unique
    if ok and keep[s]:
        rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
        ok = info[s].n_candidates == nc and info[s].status1 == 0 and (nc == 0 or abs(info[s].obj1 - o1) <= 1e-9 * max(1.0, abs(o1)))
    bad += 0 if ok else 1
    if not ok: print("MISMATCH species", s)
print("%d species checked against the oracle in %.1f s: %d mismatches" % (S, time.perf_counter() - t0, bad))
sys.exit(1 if bad else 0)
