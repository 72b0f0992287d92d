#!/bin/bash
# round 5, fourth GPU call: parity of the index kernels, the step at cfg4 / cfg3 / cfg5 share, a kernel trace of cfg4
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_d_pytest1.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_d_pytest1.log
tail -4 gpurun_out/r5_d_pytest1.log
for wl in cfg4 cfg3 cfg5_share; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_d_detail_$wl.json > gpurun_out/r5_d_bench_$wl.json 2> gpurun_out/r5_d_bench_$wl.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_d_bench_$wl.json").read().strip().splitlines()[-1])
    print("$wl", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl: no line", e); print(open("gpurun_out/r5_d_bench_$wl.err").read()[-1500:])
PY
done
bash tools/kernel_trace.sh cfg4 r5d_cfg4 6 2>&1 | tail -45
