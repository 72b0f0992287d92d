cd $GRAFT_REPO_ROOT
export PANTAX_SYNTH_CACHE=/tmp/pantax_synth_cache
for s in 512 256 128; do PANTAX_TB_SLOTS=$s python bench.py --no-cpu-baseline --no-hard --no-gaf --steps 8 --warmup 3 > gpurun_out/ab_tb_$s.json 2>/dev/null; echo "slots $s"; python3 tools/bench_summary.py gpurun_out/ab_tb_$s.json | head -3 | cut -c1-330; done
