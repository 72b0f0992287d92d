#!/bin/bash
# round 5, tenth GPU call: first build without per-row atomics, visit sort with half the shuffles, entry pairs in the coverage kernels; shapes of the one-pass rebuild
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_j_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/r5_j_pytest.log
echo "== trio_probe cfg4"; timeout 600 python tools/trio_probe.py cfg4 4 tv_rounds=2 tv_rounds=8 tv_u=2,tv_rounds=8 tv_u=2,tv_rounds=4 trio_xcd=0 trio_two_pass=1 > gpurun_out/r5_j_trio_probe_cfg4.txt 2>&1; cat gpurun_out/r5_j_trio_probe_cfg4.txt
echo "== trio_probe cfg5_share"; timeout 600 python tools/trio_probe.py cfg5_share 4 tv_rounds=2 tv_rounds=8 tv_u=2,tv_rounds=8 trio_xcd=0 > gpurun_out/r5_j_trio_probe_cfg5_share.txt 2>&1; cat gpurun_out/r5_j_trio_probe_cfg5_share.txt
for wl in cfg4 cfg3 cfg5_share; do
  timeout 600 python bench.py --workload $wl --no-seam --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_j_detail_${wl}.json > gpurun_out/r5_j_bench_${wl}.json 2> gpurun_out/r5_j_bench_${wl}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_j_bench_${wl}.json").read().strip().splitlines()[-1])
    print("$wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3), d["roofline"].get("a7_stage"))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl: no line", e); print(open("gpurun_out/r5_j_bench_${wl}.err").read()[-1500:])
PY
done
echo "== seam cfg4"; PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_j_seam_cfg4.log 2>&1; grep -v "wd_" gpurun_out/r5_j_seam_cfg4.log | head -40
