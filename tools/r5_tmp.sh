#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tmp_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r5_tmp_pytest.log
PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_tmp_seam_cfg4.log 2>&1
grep -v "wd_" gpurun_out/r5_tmp_seam_cfg4.log | grep "files_to_tables"
python - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg4.json"))
print([l for l in d["trace"]["wd_warm0"].split("\n") if "visit table" in l or "node -> hap" in l or "db upload" in l][:4])
PY
PANTAX_HIP_TRACE=1 timeout 2400 python tools/seam_bench.py cfg5 > gpurun_out/r5_tmp_seam_cfg5.log 2>&1
grep -v "wd_" gpurun_out/r5_tmp_seam_cfg5.log | grep "files_to_tables"
python - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg5.json"))
print([l for l in d["trace"]["wd_warm0"].split("\n") if "visit table" in l or "node -> hap" in l][:4])
PY
