#!/bin/bash
# round 5, twenty-third GPU call: smoke(), the long-read records with the round-ahead loads in the general coverage kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --workload cfg5_share --no-cpu-baseline --no-hard --steps 10 --detail-file gpurun_out/r5_w_detail_cfg5_share.json > gpurun_out/r5_w_bench_cfg5_share.json 2> gpurun_out/r5_w_bench_cfg5_share.err
echo "cfg5_share exit $?"; tail -c 700 gpurun_out/r5_w_bench_cfg5_share.json
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --detail-file gpurun_out/r5_w_detail_cfg5.json > gpurun_out/r5_w_bench_cfg5.json 2> gpurun_out/r5_w_bench_cfg5.err
echo "cfg5 exit $?"; tail -c 1800 gpurun_out/r5_w_bench_cfg5.json
bash tools/kernel_trace.sh cfg5_share r5w_cfg5_share 6 > gpurun_out/r5_w_trace_cfg5_share.log 2>&1; tail -2 gpurun_out/r5_w_trace_cfg5_share.log | cut -c1-160
find gpurun_out -name '*.db' -size +20M -delete 2>/dev/null
