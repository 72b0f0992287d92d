#!/usr/bin/env python3
"""Print the essentials of a bench.py JSON line."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("value %.1f %s  ms/step %.3f  (trio resident %.3f)  timer scopes/step %s" % (d["value"], d["unit"], d["ms_per_step"], d["ms_per_step_trio_index_resident"], d.get("kernel_timer_scopes_per_step", d.get("launches_per_step"))))
r = d["roofline"]
print("roofline", r["kernel"], "avg_ms %.3f frac %.3f traffic %s" % (r["avg_ms"], r["frac"], r.get("traffic")))
for k in ("runner_up", "coverage_step_kernel"):
    if k in r: print("  %s: %s avg_ms %.3f frac %.3f" % (k, r[k].get("kernel", k), r[k]["avg_ms"], r[k]["frac"]))
print("kernels", {k: round(v, 3) for k, v in list(d["kernels_ms_per_step"].items())[:16]})
if d.get("from_gaf_text"): print("gaf", d["from_gaf_text"])
if d.get("pao_hard"): print("hard", {k: v for k, v in d["pao_hard"].items() if k != "highs"})
c = d.get("cpu_baseline")
if c:
    print("cpu", c.get("value"), c.get("cores"), c.get("seconds"), c.get("error"), c.get("mem_available_gb"), "waited", c.get("parent_waited_s_for_oracle_leg"))
    h = c.get("highs") or {}
    print("highs", [(l["rows"], round(l["highs_seconds"], 2)) for l in h.get("legs", [])], "full:", {k: v for k, v in (h.get("full_lp") or {}).items() if k != "what"})
print("config", d["config"]["workload"], "| gen s", d.get("synthetic_set_generated_in_s"), "upload ms", d.get("upload_ms_once"), "host", d.get("host"))
