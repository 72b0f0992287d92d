#!/bin/bash
# round 5, twenty-second GPU call: the long-read coverage kernel with round-ahead stream loads (lib_dev) against the product; the seam's table comparison
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
PANTAX_HIP_LIB=$PWD/pantax_amd/lib_dev/libpantax_hip.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5_v_pytest_dev.log 2>&1
echo "pytest dev exit $?"; tail -3 gpurun_out/r5_v_pytest_dev.log
for lib in product dev product dev; do
  if [ $lib = dev ]; then export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_dev/libpantax_hip.so; else unset PANTAX_HIP_LIB; fi
  timeout 600 python bench.py --workload cfg5_share --no-seam --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_v_detail_$lib.json > gpurun_out/r5_v_bench_$lib.json 2> gpurun_out/r5_v_bench_$lib.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r5_v_bench_$lib.json").read().strip().splitlines()[-1])
k = d["kernels_ms_per_step"]
print("$lib cfg5_share", round(d["ms_per_step"], 3), "resident", round(d["config"]["ms_per_step_trio_index_resident"], 3), "cov_step", k.get("coverage_step_kernel"), "walk_sum", k.get("walk_sum_kernel"))
PY
done
unset PANTAX_HIP_LIB
timeout 900 python bench.py --workload cfg3 --no-cpu-baseline --no-hard --steps 10 > gpurun_out/r5_v_bench_cfg3.json 2> gpurun_out/r5_v_bench_cfg3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_v_bench_cfg3.json").read().strip().splitlines()[-1])
print("cfg3 seam equal:", d["config"].get("seam_tables_equal_to_resident_step"), d["config"].get("seam_error"), d["config"].get("files_to_tables_s"))
PY
