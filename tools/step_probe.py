#!/usr/bin/env python3
"""A/B of library options over the whole resident step on ONE process / ONE box: uploads a bench workload once, then runs the
step (index resident) under every setting, alternating, and prints the per-kernel timer table of each (minimum over the repeats).
usage: step_probe.py <workload> <reps> option=a,b[,c] [option2=x,y] [only=<kernel name prefix>]"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import native_set, workload_spec
from pantax_amd.engine import Engine
wl, reps = sys.argv[1], int(sys.argv[2])
only = [kv.split("=")[1] for kv in sys.argv[3:] if kv.startswith("only=")]
switches = [(kv.split("=")[0], kv.split("=")[1].split(",")) for kv in sys.argv[3:] if not kv.startswith("only=")]
spec = workload_spec(wl)
ns = native_set(spec)
sset = ns.make()
eng = Engine(0)
eng.upload_db(sset.species); eng.upload_packed(sset.reads)
avg = ns.avg_len()
fr = 0.5 if spec.get("long_reads") else 0.3
eng.trio_nodes_info(fetch=False)
step = lambda: eng.profile_step(avg, fr=fr, rebuild_trio=False)
step(); eng.sync()
eng.timing_enable(True)
combos = list(itertools.product(*[v for _, v in switches])) or [()]
acc = {c: {} for c in combos}
for rep in range(reps):
    for c in combos:
        for (k, _), v in zip(switches, c):
            eng.set_option(k, v)
        eng.timing_reset()
        step(); eng.sync()
        for k, (n, ms) in eng.timing_get().items():
            acc[c].setdefault(k, []).append(ms)
# wall clock of a stream of steps, one enqueued ahead (what bench.py's value times)
import time
eng.timing_enable(False)
wall = {c: [] for c in combos}
def stream_of(n):
    eng.profile_step_enqueue(avg, fr=fr, rebuild_trio=False)
    for i in range(n):
        if i + 1 < n:
            eng.profile_step_enqueue(avg, fr=fr, rebuild_trio=False)
        eng.profile_step_collect()
for rep in range(reps):
    for c in combos:
        for (k, _), v in zip(switches, c):
            eng.set_option(k, v)
        stream_of(3); eng.sync()
        t0 = time.perf_counter(); stream_of(10); eng.sync()
        wall[c].append((time.perf_counter() - t0) * 100.0)
for c in combos:
    tag = " ".join("%s=%s" % (k, v) for (k, _), v in zip(switches, c)) or "(default)"
    print(tag, "pipelined ms/step", [round(x, 3) for x in wall[c]])
    tab = {k: round(min(v), 4) for k, v in sorted(acc[c].items(), key=lambda kv: -min(kv[1])) if not only or any(k.startswith(o) for o in only)}
    print(tag, "sum %.3f" % sum(min(v) for v in acc[c].values()), tab)
eng.close()
