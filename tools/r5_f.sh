#!/bin/bash
# round 5, sixth GPU call: heads stored only where they changed; LDS window of trio_bases in the short-read coverage kernel (A/B); cfg5 at full size
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
for tw in 0 2048; do
  PANTAX_COV_TRIO_WIN=$tw timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r5_f_pytest_tw$tw.log 2>&1
  echo "tw=$tw pytest exit $?"; tail -2 gpurun_out/r5_f_pytest_tw$tw.log
done
echo "== trio_probe cfg4"; timeout 300 python tools/trio_probe.py cfg4 5 2>&1 | tail -1
for tw in 0 1024 2048; do
  for wl in cfg4 cfg3; do
    PANTAX_COV_TRIO_WIN=$tw timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_f_detail_${wl}_tw$tw.json > gpurun_out/r5_f_bench_${wl}_tw$tw.json 2> gpurun_out/r5_f_bench_${wl}_tw$tw.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_f_bench_${wl}_tw$tw.json").read().strip().splitlines()[-1])
    print("$wl tw=$tw", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl tw=$tw: no line", e); print(open("gpurun_out/r5_f_bench_${wl}_tw$tw.err").read()[-1500:])
PY
  done
done
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --detail-file gpurun_out/r5_f_detail_cfg5.json > gpurun_out/r5_f_bench_cfg5.json 2> gpurun_out/r5_f_bench_cfg5.err
echo "cfg5 exit $?"; tail -c 3000 gpurun_out/r5_f_bench_cfg5.json; tail -5 gpurun_out/r5_f_bench_cfg5.err
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k cfg5_full > gpurun_out/r5_f_pytest_cfg5.log 2>&1
echo "cfg5 test exit $?"; tail -5 gpurun_out/r5_f_pytest_cfg5.log
