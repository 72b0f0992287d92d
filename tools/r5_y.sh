#!/bin/bash
# round 5: the FILE seam at BASELINE configs[4]'s full size on one GPU (1.1e10 path steps: groups of species inside one pantax_hip_profile call)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
df -h /tmp /dev/shm | tail -2
PANTAX_HIP_TRACE=1 timeout 2400 python tools/seam_bench.py cfg5 > gpurun_out/r5_y_seam_cfg5.log 2>&1
echo "exit $?"; grep -v "wd_" gpurun_out/r5_y_seam_cfg5.log | head -60
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/seam_bench_cfg5.json"))
    for l in d["trace"]["wd_warm0"].split("\n"):
        if "piece:" in l: continue
        print(l)
except Exception as e:
    print("no json", e)
PY
