#!/usr/bin/env python3
"""Measurement: the FILE seam (pantax_hip_profile: DB directory + GAF text in, the two tables out) at a bench workload's size --
bench.py's files-to-files leg alone (no resident steps, no CPU legs).  usage: seam_bench.py [workload=cfg4] [reads]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import synthdata as synth
from pantax_amd.engine import Engine
if os.environ.get("SEAM_WITH_TORCH"):      # experiment: does a process that has initialised torch's HIP context load more slowly?
    import torch
    torch.cuda.set_device(0)
    _t = torch.ones(1, device="cuda") if os.environ["SEAM_WITH_TORCH"] == "2" else None
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
spec = bench.workload_spec(wl, reads=int(sys.argv[2]) if len(sys.argv) > 2 else None)
threads = max(1, min(64, os.cpu_count() or 1))
t0 = time.perf_counter()
ns = bench.native_set(spec, threads=threads)
rd = ns.reads()
species = ns.graphs()
print("set generated in %.1f s" % (time.perf_counter() - t0), flush=True)
need = 145 * spec["reads"] + 12 * (sum(g.n_nodes for g in species) + int(sum(int(g.path_off[-1]) for g in species)))
root = bench.gaf_tmp_dir(need)
assert root, "no temporary directory with %.0f GB" % (need / 1e9)
with tempfile.TemporaryDirectory(dir=root) as td:
    gp = os.path.join(td, "reads.gaf")
    t0 = time.perf_counter()
    nb = synth.write_gaf_parallel(rd, gp, threads=threads)
    print("GAF: %.2f GB written in %.1f s under %s" % (nb / 1e9, time.perf_counter() - t0, root), flush=True)
    del rd
    res = bench.file_seam_leg(0, species, gp, td, threads, None, spec["reads"], fr=0.5 if spec.get("long_reads") else 0.3)   # (the calls run in a child process)
res["workload"] = spec["name"]
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "seam_bench_%s.json" % spec["name"]), "w") as f:
    json.dump(res, f, indent=1)
