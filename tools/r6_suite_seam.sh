#!/bin/bash
# round 6: whole GPU suite, then the file seam at cfg4
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_suite2.txt; cat gpurun_out/r6_suite2.txt
timeout 1500 python tools/seam_bench.py cfg4 > gpurun_out/r6_seam_bench2.txt 2>gpurun_out/r6_seam_bench2.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg4.json"))
for k in ("files_to_tables_cold_s","files_to_tables_warm_s","files_to_tables_warm_s_both","db_load_cold_s","db_load_warm_s","gaf_load_s","db_image_gb","cold_and_warm_tables_same_bytes"): print(k, d.get(k))
print(d.get("phases_ms_warm"))
t=d["trace"]["wd_warm1"]
print("\n".join(l[:200] for l in t.split("\n") if "gaf_tokenize]   piece" not in l and "upload_segments" not in l))
PY
tail -3 gpurun_out/r6_seam_bench2.err | cut -c1-300
