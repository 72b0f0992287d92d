#!/usr/bin/env python3
"""Debug A/B: per-step (default) vs per-read coverage kernel on the cfg2 workload."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from pantax_amd import synth
    from pantax_amd.engine import Engine
    sset = synth.make_set(20260503, 1, 10, 1_000_000, 5_000_000)
    eng = Engine(0)
    eng.upload_db(sset.species); eng.upload_packed(sset.reads)
    eng.rcls_profile(want_species=False); eng.trio_nodes_info(fetch=False)
    for _ in range(3): eng.get_node_abundances(fetch=False)
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(10): eng.get_node_abundances(fetch=False)
    t = eng.timing_get()
    print(json.dumps({k: v[1] / v[0] for k, v in t.items()}))
    sys.exit(0)
for mode in ["step", "read"]:
    env = dict(os.environ, PANTAX_HIP_COV_MODE=mode)
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print(mode, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
