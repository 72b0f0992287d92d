#!/usr/bin/env python3
"""round 6: the short-read coverage kernel over a db that holds a QUARTER of the species of the resident reads (what a group of the file seam sees):
only the items of the db's id range are launched.  usage: r6_cov_range_probe.py [workload=cfg4_share]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import native_set, workload_spec
from pantax_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4_share"
sset = native_set(workload_spec(wl)).make()
S = len(sset.species)
for lo, hi in ((0, S), (0, S // 4), (S // 4, S // 2), (3 * S // 4, S)):
    eng = Engine(0)
    eng.upload_db(sset.species[lo:hi]); eng.upload_packed(sset.reads)
    eng.timing_enable(True)
    eng.timing_reset()
    eng.rcls_profile(want_species=False); eng.trio_nodes_info(fetch=False); eng.get_node_abundances(fetch=False); eng.sync()
    print("  first pass on the fresh db", {k: round(ms / max(n, 1), 4) for k, (n, ms) in eng.timing_get().items() if k in ("coverage_fast_kernel", "bin_slots_kernel", "popcount_kernel", "zero_fill_kernel")})
    acc = {}
    for rep in range(3):
        eng.timing_reset()
        eng.rcls_profile(want_species=False); eng.get_node_abundances(fetch=False); eng.sync()
        for k, (n, ms) in eng.timing_get().items():
            acc.setdefault(k, []).append(ms / max(n, 1))
    print("species [%d, %d)" % (lo, hi), {k: round(min(v), 4) for k, v in sorted(acc.items(), key=lambda kv: -min(kv[1]))})
    eng.close()
