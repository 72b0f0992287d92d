#!/usr/bin/env python3
"""Print the essentials of a bench.py JSON line (and its length: the driver keeps about 8 KB of tail)."""
import json, sys
raw = [l for l in open(sys.argv[1]) if l.startswith("{")][0]
d = json.loads(raw)
c = d["config"]
print("value %.1f %s  ms/step %.3f  (with index rebuild %.3f, two-pass %.3f, first build %.1f)  n_gpus %s  line %d bytes" % (d["value"], d["unit"], d["ms_per_step"], c.get("ms_per_step_with_index_rebuild") or c.get("ms_per_step_trio_index_resident") or -1, c.get("ms_per_step_two_pass_rebuild") or -1, c.get("index_first_build_ms") or -1, d["n_gpus"], len(raw)))
r = d["roofline"]
print("roofline", r["kernel"], "avg_ms %.3f frac %.3f by-counter %s traffic %s" % (r["avg_ms"], r["frac"], r.get("frac_by_counter_bytes"), r.get("traffic")))
ms_of = lambda k: k.get("avg_ms", k.get("ms_per_step_summed_over_the_dbs", -1.0))
for kk in r.get("kernels", []):
    print("  %-28s avg_ms %.3f x %.1f/step  frac %.3f  by-counter %s" % (kk["kernel"], ms_of(kk), kk.get("launches_per_step", 1), kk["frac"], kk.get("frac_by_counter_bytes")))
print("  sum of kernels per step", r.get("sum_of_kernels_ms_per_step"))
if d.get("value_gaf_to_tables"): print("gaf->tables", d["value_gaf_to_tables"])
for k in ("runner_up", "coverage"):
    if k in r: print("  %s: %s avg_ms %.3f frac %.3f by-counter %s" % (k, r[k].get("kernel", k), ms_of(r[k]), r[k]["frac"], r[k].get("frac_by_counter_bytes")))
print("kernels", {k: round(v, 3) for k, v in list(d["kernels_ms_per_step"].items())[:16]})
print("gaf", {k: c.get(k) for k in ("from_gaf_text_s", "from_gaf_text_to_resident_s", "from_gaf_text_mreads_per_s", "gaf_gb", "gaf_gb_per_s", "pinned_h2d_ceiling_gb_per_s", "gaf_gb_per_s_of_ceiling", "tables_equal_to_packed_input_run", "from_gaf_text_error")})
print("l1", {k: c.get(k) for k in ("abundance_l1_vs_oracle", "abundance_l1_species_checked", "abundance_l1_error", "pao_hard_lad_ms_per_species", "pao_hard_objective_rel_diff_vs_oracle", "pao_hard_abundance_l1_vs_oracle")})
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"])
print("config", c["workload"], "| ranks", c.get("ranks_seen"), c.get("exchange"), "| detail", d.get("detail_file"))
