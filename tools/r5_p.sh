#!/bin/bash
# round 5, sixteenth GPU call: coverage kernel with level 1 of the coming round requested a round ahead (variant build -DCOV_PREFETCH) against the product, same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
PANTAX_HIP_LIB=$PWD/pantax_amd/lib_v1/libpantax_hip.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5_p_pytest_v1.log 2>&1
echo "pytest v1 exit $?"; tail -3 gpurun_out/r5_p_pytest_v1.log
for lib in product v1 product v1; do
  for wl in cfg4 cfg3; do
    if [ $lib = v1 ]; then export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_v1/libpantax_hip.so; else unset PANTAX_HIP_LIB; fi
    timeout 600 python bench.py --workload $wl --no-seam --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_p_detail_${wl}_$lib.json > gpurun_out/r5_p_bench_${wl}_$lib.json 2> gpurun_out/r5_p_bench_${wl}_$lib.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_p_bench_${wl}_$lib.json").read().strip().splitlines()[-1])
    k = d["kernels_ms_per_step"]
    print("$lib $wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), "cov", k.get("coverage_fast_kernel"), "node stats", k.get("node_cov_stats_kernel"), "file", k.get("trio_file_kernel"))
except Exception as e:
    print("$lib $wl: no line", e); print(open("gpurun_out/r5_p_bench_${wl}_$lib.err").read()[-1500:])
PY
  done
done
