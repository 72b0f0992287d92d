#!/bin/bash
# round 5, eleventh GPU call: a9 statistics over compacted non-zero rows; shapes of the one-pass rebuild; GAF taper
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_k_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/r5_k_pytest.log
echo "== trio_probe cfg4"; timeout 600 python tools/trio_probe.py cfg4 4 tf_rounds=1 tf_rounds=3 tf_u=2,tf_rounds=2 tf_u=2,tf_rounds=3 tf_u=8,tf_rounds=1 tf_u=2,tf_rounds=1 > gpurun_out/r5_k_trio_probe_cfg4.txt 2>&1; cat gpurun_out/r5_k_trio_probe_cfg4.txt
echo "== trio_probe cfg5_share"; timeout 600 python tools/trio_probe.py cfg5_share 4 tf_rounds=1 tf_rounds=3 tf_u=2,tf_rounds=2 tf_u=8,tf_rounds=1 tf_u=2,tf_rounds=1 > gpurun_out/r5_k_trio_probe_cfg5_share.txt 2>&1; cat gpurun_out/r5_k_trio_probe_cfg5_share.txt
for wl in cfg4 cfg3 cfg5_share; do
  timeout 600 python bench.py --workload $wl --no-seam --no-cpu-baseline --no-hard --no-gaf --steps 10 --detail-file gpurun_out/r5_k_detail_${wl}.json > gpurun_out/r5_k_bench_${wl}.json 2> gpurun_out/r5_k_bench_${wl}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_k_bench_${wl}.json").read().strip().splitlines()[-1])
    print("$wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3), "L1", d["config"].get("abundance_l1_vs_oracle"))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl: no line", e); print(open("gpurun_out/r5_k_bench_${wl}.err").read()[-1500:])
PY
done
echo "== seam cfg4"; PANTAX_HIP_TRACE=1 timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_k_seam_cfg4.log 2>&1; grep -v "wd_" gpurun_out/r5_k_seam_cfg4.log | head -60
