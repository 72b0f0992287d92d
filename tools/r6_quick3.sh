#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 1800 python bench.py --workload refdb --steps 10 --detail-file gpurun_out/r6_records_detail_refdb.json > gpurun_out/r6_records_bench_refdb.json 2> gpurun_out/r6_records_bench_refdb.err
echo "refdb exit $?"; python3 tools/bench_summary.py gpurun_out/r6_records_bench_refdb.json | head -8 | cut -c1-300
