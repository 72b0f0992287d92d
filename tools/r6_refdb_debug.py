#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import synthdata as synth
from oracle import oracle as orc
from pantax_amd.engine import Engine, metrics_to_dicts
from tests.helpers import select_reads
ns = synth.RefDbSet(20260507, 1_200_000, scale=0.05, threads=16)
sset = ns.make(); rd = sset.reads
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(rd)
sp, rc, bs, lm, uq = eng.rcls_profile()
keep, absolute, met, info, passed, s_all, s_pass = eng.profile_step(sset.avg_len())
got = metrics_to_dicts(met, eng.H)
hb = np.cumsum([0] + [g.n_paths for g in sset.species])
bad = 0
for s, g in enumerate(sset.species):
    if not keep[s] or info[s].n_candidates == 0:
        continue
    G = orc.Graph(g.node_len, g.path_off, g.path_nodes); T = orc.TrioTable(G)
    so, nid, ps, pe = select_reads(rd, np.nonzero(sp == s)[0])
    b, c, t, na = orc.node_coverage(G, T, g.range_start, so, nid, ps, pe)
    rc_, omet, nc, o1, o2 = orc.optimize_species(G, T, b, c, t)
    for h, e in enumerate(orc.metrics_to_dicts(omet)):
        gv, ev = got[hb[s] + h]["path_base_cov"], e["path_base_cov"]
        if (gv is None) != (ev is None) or (ev is not None and abs(gv - ev) > 1e-6 * max(1, abs(ev))):
            bad += 1
            if bad <= 12:
                pn = g.path_nodes[int(g.path_off[h]):int(g.path_off[h + 1])]
                print("species", s, g.name, "H", g.n_paths, "V", g.n_nodes, "hap", h, "gpu", gv, "oracle", ev, "path steps", len(pn), "distinct", len(np.unique(pn)),
                      "sum cov over distinct", int(c[np.unique(pn)].sum()), "over steps", int(c[pn].sum()), "len distinct", int(g.node_len[np.unique(pn)].sum()), "len steps", int(g.node_len[pn].sum()))
print("mismatches", bad)
