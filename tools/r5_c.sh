#!/bin/bash
# round 5, third GPU call: the rebuilt index (rows in filing order): parity suite first, then the step at cfg4 / cfg3 / cfg5 share, then the file seam
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r5_c_pytest1.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_c_pytest1.log
tail -5 gpurun_out/r5_c_pytest1.log
for wl in cfg4 cfg3 cfg5_share; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_c_detail_$wl.json > gpurun_out/r5_c_bench_$wl.json 2> gpurun_out/r5_c_bench_$wl.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_c_bench_$wl.json").read().strip().splitlines()[-1])
    print("$wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl: no line", e); print(open("gpurun_out/r5_c_bench_$wl.err").read()[-1500:])
PY
done
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_pipeline.py > gpurun_out/r5_c_pytest2.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_c_pytest2.log
tail -5 gpurun_out/r5_c_pytest2.log
timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_c_seam_cfg4.log 2>&1
python - <<PY
import json
d = json.load(open("gpurun_out/seam_bench_cfg4.json"))
print({k: d[k] for k in ("files_to_tables_cold_s", "files_to_tables_warm_s", "files_to_tables_warm_s_both", "db_load_cold_s", "db_load_warm_s", "gaf_load_s", "strain_step_s")})
print(d["phases_ms_warm"])
PY
