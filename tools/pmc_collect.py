#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_step.sh) into one JSON: per kernel the mean counter values per launch, the
register / LDS footprint rocprofv3 reports for the dispatch, and HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) x 1024
(both counters are in KB and come from separate passes).  gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE =
TCC_EA0_RDREQ x 64 B and reports half the bytes of a wide (16 B/lane) coalesced streaming read; `hbm_bytes_fetch_x2` is
the corrected upper bound (2 x fetch + write), `hbm_bytes_per_launch` the raw sum.
usage: pmc_collect.py <dir with g*/..._counter_collection.csv> <workload> <out.json> [min_total_launch_share]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import workload_key, workload_spec
src, wl, out = sys.argv[1], sys.argv[2], sys.argv[3]
acc = {}    # kernel -> counter -> [sum, n]
meta = {}
for fn in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    with open(fn, newline="") as f:
        for row in csv.DictReader(f):
            full = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            k = full.replace("void ptx::", "").replace("ptx::", "").split("<")[0]
            if k == "scan_chained_kernel":      # one entry per instantiation, named like the library's timer labels (bench.py keys)
                for tag in ("Row", "Pat", "TrioFirst", "FlagWord", "GroupCount", "HeadCount", "Sample", "Len"):
                    if tag + "Load" in full:
                        k = "scan_chained_kernel<%s>" % tag
                        break
            if k.startswith("__amd") or "at::" in k:
                continue
            c = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"]); c[1] += 1
            meta[k] = dict(vgpr=int(row["VGPR_Count"]), agpr=int(row["Accum_VGPR_Count"]), sgpr=int(row["SGPR_Count"]),
                           lds_bytes=int(row["LDS_Block_Size"]), workgroup=int(row["Workgroup_Size"]), scratch=int(row["Scratch_Size"]))
res = {"_comment": ["rocprofv3 --pmc passes over tools/step_driver.py %s 2 (two resident steps, all launches averaged), one counter group per run" % wl,
                    "(tools/pmc_step.sh); FETCH_SIZE / WRITE_SIZE are KB.  hbm_bytes_per_launch = (FETCH + WRITE) x 1024 raw;",
                    "hbm_bytes_fetch_x2 = the guide's gfx950 correction for wide coalesced reads applied to the whole fetch (upper bound).",
                    "waves/SIMD by VGPRs = floor(512 / vgpr_alloc) capped at 8 (unified 512-entry file per SIMD lane on gfx950)."],
       "workload": workload_key(workload_spec(wl)), "kernels": {}}
for k, cs in sorted(acc.items()):
    d = {c: v[0] / v[1] for c, v in cs.items()}
    d["launches_seen"] = max(v[1] for v in cs.values())
    d.update(meta[k])
    tot = meta[k]["vgpr"] + meta[k]["agpr"]
    alloc = ((tot + 7) // 8) * 8 if tot else 8
    d["waves_per_simd_by_vgpr"] = min(8, 512 // alloc)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = int((d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
        d["hbm_bytes_fetch_x2"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
    res["kernels"][k] = d
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res["kernels"]), "kernels")
