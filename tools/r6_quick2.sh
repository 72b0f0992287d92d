#!/bin/bash
# the whole GPU suite + the default bench run on the round's last code
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_records_pytest.log 2>&1
echo "pytest exit $?"; tail -2 gpurun_out/r6_records_pytest.log
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file gpurun_out/r6_records_detail_default.json > gpurun_out/r6_records_bench_default.json 2> gpurun_out/r6_records_bench_default.err
echo "default bench exit $?"; python3 tools/bench_summary.py gpurun_out/r6_records_bench_default.json | cut -c1-400 | head -12
