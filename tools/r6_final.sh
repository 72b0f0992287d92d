#!/bin/bash
# the round's last check: the whole GPU suite, smoke(), the default bench run with the driver's flags, the reference-DB-shaped workload
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_final_pytest.log 2>&1; echo "pytest exit $?"; tail -1 gpurun_out/r6_final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file gpurun_out/r6_final_detail.json > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err; echo "bench exit $?"
python3 tools/bench_summary.py gpurun_out/r6_final_bench.json | cut -c1-330 | head -9
timeout 900 python bench.py --workload cfg5_share --no-cpu-baseline --no-hard --steps 10 --detail-file gpurun_out/r6_final_detail_cfg5share.json > gpurun_out/r6_final_bench_cfg5share.json 2> gpurun_out/r6_final_bench_cfg5share.err; echo "cfg5_share exit $?"
python3 tools/bench_summary.py gpurun_out/r6_final_bench_cfg5share.json | cut -c1-330 | head -8
