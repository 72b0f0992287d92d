#!/usr/bin/env python3
"""Debug: ms per step of profile_steps_pipelined for growing step counts, with / without the timing filter and the lookahead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PANTAX_DEBUG_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import synthdata as synth
from pantax_amd.engine import Engine
from pantax_amd.pipeline import StepConfig, profile_steps_pipelined, profile_step
S, R, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sset = synth.make_set(20260504, S, 10, R, L)
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
names = [g.name for g in sset.species]; haps = [h for g in sset.species for h in g.hap_names]
avg = sset.avg_len(); cfg = StepConfig()
for _ in range(3): profile_step(eng, names, haps, avg, cfg)
for filt in (None, "lad_solve_kernel"):
    for la in ("1", "0"):
        os.environ["PANTAX_STEP_LOOKAHEAD"] = la
        if filt: eng.timing_enable(True); eng.timing_filter(filt); eng.timing_reset()
        res = []
        for n in (5, 10, 20, 40):
            eng.sync(); t0 = time.perf_counter(); profile_steps_pipelined(eng, names, haps, avg, n, cfg); eng.sync()
            res.append((n, (time.perf_counter() - t0) / n * 1e3))
        if filt: eng.timing_get(); eng.timing_enable(False); eng.timing_filter(None)
        print("timing filter %s lookahead %s:" % (filt, la), " ".join("%d steps %.3f ms" % r for r in res))
