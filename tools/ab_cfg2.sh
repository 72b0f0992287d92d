cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --workload cfg2 --no-cpu-baseline --no-hard --no-gaf --steps 40 --warmup 5 > gpurun_out/ab_$i.json 2>/dev/null; python3 tools/bench_summary.py gpurun_out/ab_$i.json | head -1; done
python tools/step_host_overhead.py cfg2 2>&1 | head -1
