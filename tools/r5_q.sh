#!/bin/bash
# round 5, seventeenth GPU call: prefetching coverage + node statistics, LDS-staged visit packing: suite, step, seam
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_q_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r5_q_pytest.log
for wl in cfg4 cfg3 cfg5_share; do
  timeout 600 python bench.py --workload $wl --no-seam --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_q_detail_${wl}.json > gpurun_out/r5_q_bench_${wl}.json 2> gpurun_out/r5_q_bench_${wl}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_q_bench_${wl}.json").read().strip().splitlines()[-1])
    print("$wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3), d["roofline"].get("runner_up", {}).get("avg_ms"), d["roofline"].get("runner_up", {}).get("frac"))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl: no line", e); print(open("gpurun_out/r5_q_bench_${wl}.err").read()[-1500:])
PY
done
echo "== seam cfg4"; PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_q_seam_cfg4.log 2>&1; grep -v "wd_" gpurun_out/r5_q_seam_cfg4.log | head -30
python - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg4.json"))
for l in d["trace"]["wd_warm0"].split("\n"):
    if "db_upload" in l or "upload_segments" in l: print(l)
PY
