for w in 1024 2048 4096 16384; do
echo "== PANTAX_SSN_WGS=$w"
PANTAX_SSN_WGS=$w timeout 600 python bench.py --workload cfg4 --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 5 > gpurun_out/wgs_$w.json 2>/dev/null
python3 tools/bench_summary.py gpurun_out/wgs_$w.json | grep -E "^value|^kernels"
done
