#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1500 python tools/seam_bench.py ${1:-cfg4} > gpurun_out/r6_seam_bench3.txt 2>gpurun_out/r6_seam_bench3.err
python3 - <<'PY'
import json,glob,os
f=max(glob.glob("gpurun_out/seam_bench_*.json"), key=os.path.getmtime)
d=json.load(open(f))
for k in ("workload","files_to_tables_cold_s","files_to_tables_warm_s","files_to_tables_warm_s_both","db_load_cold_s","db_load_warm_s","gaf_load_s","db_image_gb","cold_and_warm_tables_same_bytes"): print(k, d.get(k))
print(d.get("phases_ms_warm"))
print(d["trace"]["wd_warm1"][:7000])
PY
tail -3 gpurun_out/r6_seam_bench3.err | cut -c1-300
