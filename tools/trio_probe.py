#!/usr/bin/env python3
"""Times the unique-trio index build of a bench workload (HIP events on the library's stream), several rebuilds.
usage: [PANTAX_HIP_LIB=<build>] trio_probe.py [workload] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import native_set, workload_spec
from pantax_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ns = native_set(workload_spec(wl))
eng = Engine(0)
eng.upload_db(ns.graphs())
eng.trio_nodes_info(fetch=False); eng.sync()
eng.timing_enable(True)
# further arguments: option settings to compare on the same resident db, e.g. "tv_u=2,tv_rounds=8" "trio_two_pass=1"
variants = [""] + sys.argv[3:]
for var in variants:
    sets = [kv.split("=") for kv in var.split(",") if kv]
    for k, v in sets:
        eng.set_option(k, v)
    acc = {}
    for _ in range(reps):
        eng.timing_reset()
        eng.trio_index_prefetch(); eng.sync()          # the step's build: on the side stream, without the exporters' window starts
        for k, (n, ms) in eng.timing_get().items():
            acc.setdefault(k, []).append(ms)
    print(os.environ.get("PANTAX_HIP_LIB", "product"), var or "defaults", {k: round(min(v), 4) for k, v in sorted(acc.items(), key=lambda kv: -min(kv[1]))})
    for k, v in sets:
        eng.set_option(k, None)
eng.close()
