#!/usr/bin/env python3
"""Profiling driver: uploads one bench workload and runs N resident steps (pantax_hip_profile_step), nothing else.
usage: step_driver.py [cfg2|cfg3|cfg4_share|cfg4|cfg5_share] [n_steps] [rebuild_trio 0/1]   (the set bench.py times: native generator, seconds)
Put it directly after `--` under rocprofv3 (tools/pmc_step.sh, tools/kernel_trace.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import native_set, workload_spec
from pantax_amd.engine import Engine
from pantax_amd.pipeline import StepConfig, profile_step, profile_steps_pipelined
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rebuild = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
spec = workload_spec(wl)
sset = native_set(spec).make()
names = [g.name for g in sset.species]
haps = [h for g in sset.species for h in g.hap_names]
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
cfg = StepConfig(rebuild_trio=rebuild, fr=0.5 if spec.get("long_reads") else 0.3)
out = profile_step(eng, names, haps, sset.avg_len(), cfg)                                   # allocations
out = profile_steps_pipelined(eng, names, haps, sset.avg_len(), n, cfg)[-1]                  # the path bench.py times: one step enqueued ahead
eng.sync()
print("step_driver: %s, %d steps, %d strain rows" % (wl, n, len(out[1])))
eng.close()
