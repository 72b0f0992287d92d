#!/bin/bash
# round 6 final records, part A: the whole GPU suite, the default bench run (what the driver runs), the other workloads
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_records_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/r6_records_pytest.log
( time timeout 1800 python bench.py --detail-file gpurun_out/r6_records_detail_default.json > gpurun_out/r6_records_bench_default.json 2> gpurun_out/r6_records_bench_default.err ) 2> gpurun_out/r6_records_bench_default.time
echo "default bench exit $?"; tail -3 gpurun_out/r6_records_bench_default.time; python3 tools/bench_summary.py gpurun_out/r6_records_bench_default.json | cut -c1-700
for wl in cfg3 cfg2 cfg5_share; do
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-hard --steps 10 --detail-file gpurun_out/r6_records_detail_${wl}.json > gpurun_out/r6_records_bench_${wl}.json 2> gpurun_out/r6_records_bench_${wl}.err
  echo "$wl exit $?"; python3 tools/bench_summary.py gpurun_out/r6_records_bench_${wl}.json | head -9 | cut -c1-400
done
timeout 1200 python bench.py --workload cfg5 --steps 5 --warmup 2 --detail-file gpurun_out/r6_records_detail_cfg5.json > gpurun_out/r6_records_bench_cfg5.json 2> gpurun_out/r6_records_bench_cfg5.err
echo "cfg5 exit $?"; python3 tools/bench_summary.py gpurun_out/r6_records_bench_cfg5.json | head -9 | cut -c1-400
timeout 1800 python bench.py --workload refdb --steps 10 --detail-file gpurun_out/r6_records_detail_refdb.json > gpurun_out/r6_records_bench_refdb.json 2> gpurun_out/r6_records_bench_refdb.err
echo "refdb exit $?"; python3 tools/bench_summary.py gpurun_out/r6_records_bench_refdb.json | head -12 | cut -c1-500; tail -3 gpurun_out/r6_records_bench_refdb.err | cut -c1-300
