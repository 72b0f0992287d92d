#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
( time timeout 1500 python bench.py --detail-file gpurun_out/r5_final_detail_default.json > gpurun_out/r5_final_bench_default.json 2> gpurun_out/r5_final_bench_default.err ) 2> gpurun_out/r5_final_bench_default.time
echo "default bench exit $?"; tail -3 gpurun_out/r5_final_bench_default.time
timeout 900 python bench.py --workload cfg5_share --no-cpu-baseline --no-hard --steps 10 --detail-file gpurun_out/r5_final_detail_cfg5_share.json > gpurun_out/r5_final_bench_cfg5_share.json 2> gpurun_out/r5_final_bench_cfg5_share.err
echo "cfg5_share exit $?"
