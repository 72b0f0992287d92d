#!/bin/bash
# round 5, nineteenth GPU call: suite + seam after the fill / packing changes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_s_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r5_s_pytest.log
echo "== seam cfg4"; PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_s_seam_cfg4.log 2>&1; grep -v "wd_" gpurun_out/r5_s_seam_cfg4.log | grep "files_to_tables\|db_load\|gaf_load"
python - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg4.json"))
for l in d["trace"]["wd_warm1"].split("\n"):
    if "piece:" in l: continue
    print(l)
PY
timeout 600 python tools/gaf_ingest_probe.py cfg4 5 > gpurun_out/r5_s_gaf_ingest.txt 2>&1; tail -2 gpurun_out/r5_s_gaf_ingest.txt
