#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tmp_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r5_tmp_pytest.log
timeout 900 python tools/gaf_ingest_probe.py cfg4 5 > gpurun_out/r5_tmp_gaf_ingest.txt 2>&1; tail -2 gpurun_out/r5_tmp_gaf_ingest.txt
PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_tmp_seam_cfg4.log 2>&1
grep -v "wd_" gpurun_out/r5_tmp_seam_cfg4.log | grep "files_to_tables\|gaf_load"
python - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg4.json"))
print([l for l in d["trace"]["wd_warm0"].split("\n") if "locus-grouped" in l or "upload_staged" in l or "total inside" in l][:4])
PY
timeout 600 python tools/stress.py 150 95000 > gpurun_out/r5_stress_c.log 2>&1; tail -1 gpurun_out/r5_stress_c.log
