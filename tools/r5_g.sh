#!/bin/bash
# round 5, seventh GPU call: one-pass rebuild (trio_file_kernel), NUMA-bound upload crew, many-db bench with serial kernel clocks
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_g_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/r5_g_pytest.log
echo "== trio_probe cfg4"; timeout 300 python tools/trio_probe.py cfg4 5 2>&1 | tail -2
for wl in cfg4 cfg3 cfg5_share; do
  timeout 600 python bench.py --workload $wl --no-seam --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_g_detail_${wl}.json > gpurun_out/r5_g_bench_${wl}.json 2> gpurun_out/r5_g_bench_${wl}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_g_bench_${wl}.json").read().strip().splitlines()[-1])
    print("$wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl: no line", e); print(open("gpurun_out/r5_g_bench_${wl}.err").read()[-1500:])
PY
done
echo "== seam cfg4"; PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_g_seam_cfg4.log 2>&1; grep -i "numa\|files_to_tables\|warm\|cold" gpurun_out/r5_g_seam_cfg4.log | head -20
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --detail-file gpurun_out/r5_g_detail_cfg5.json > gpurun_out/r5_g_bench_cfg5.json 2> gpurun_out/r5_g_bench_cfg5.err
echo "cfg5 exit $?"; tail -c 3500 gpurun_out/r5_g_bench_cfg5.json; tail -5 gpurun_out/r5_g_bench_cfg5.err
