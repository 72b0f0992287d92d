#!/usr/bin/env python3
"""A/B of library options on ONE process / ONE box: uploads a bench workload once, then times the binning and coverage
stages (HIP events on the library's stream) under every setting, alternating (pantax_hip_set_option between the runs).
usage: stage_probe.py <workload> <reps> option=a,b[,c] [option2=x,y]   e.g. stage_probe.py cfg3 5 cov_xcd=0,1 covf_shape=283,2823"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import native_set, workload_spec
from pantax_amd.engine import Engine
wl, reps = sys.argv[1], int(sys.argv[2])
switches = [(kv.split("=")[0], kv.split("=")[1].split(",")) for kv in sys.argv[3:]]
sset = native_set(workload_spec(wl)).make()
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
eng.rcls_profile(want_species=False); eng.trio_nodes_info(fetch=False)
eng.get_node_abundances(fetch=False); eng.sync()
eng.timing_enable(True)
combos = list(itertools.product(*[v for _, v in switches])) or [()]
acc = {c: {} for c in combos}
for rep in range(reps):
    for c in combos:
        for (k, _), v in zip(switches, c):
            eng.set_option(k, v)
        eng.timing_reset()
        eng.rcls_profile(want_species=False)
        eng.get_node_abundances(fetch=False)
        eng.sync()
        for k, (n, ms) in eng.timing_get().items():
            acc[c].setdefault(k, []).append(ms / max(n, 1))
for c in combos:
    tag = " ".join("%s=%s" % (k, v) for (k, _), v in zip(switches, c)) or "(default)"
    print(tag, {k: round(min(v), 4) for k, v in sorted(acc[c].items(), key=lambda kv: -min(kv[1]))})
eng.close()
