#!/bin/bash
# one gpurun call: gpu tests, default bench, kernel trace and PMC passes at cfg3.  usage: gpu_round.sh <tag> [what...]
tag=$1; shift
what=${@:-tests bench trace pmc}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc > gpurun_out/${tag}_nproc.txt; free -g >> gpurun_out/${tag}_nproc.txt
for w in $what; do
case $w in
tests) ( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/${tag}_tests.log 2>&1; tail -5 gpurun_out/${tag}_tests.log ;;
triotests) ( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q ) > gpurun_out/${tag}_tests.log 2>&1; tail -4 gpurun_out/${tag}_tests.log ;;
newtests) ( time timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q --durations=15 ) > gpurun_out/${tag}_tests.log 2>&1; tail -25 gpurun_out/${tag}_tests.log ;;
bench) ( time timeout 1500 python bench.py ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python3 tools/bench_summary.py gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err ;;
bench3) ( time timeout 900 python bench.py --workload cfg3 ) > gpurun_out/${tag}_bench_cfg3.json 2> gpurun_out/${tag}_bench_cfg3.err; python3 tools/bench_summary.py gpurun_out/${tag}_bench_cfg3.json; tail -5 gpurun_out/${tag}_bench_cfg3.err ;;
flow) ( time timeout 600 python bench.py --workload cfg2 --highs-full-time-limit 10 --hard-species 2 ) > gpurun_out/${tag}_flow.json 2> gpurun_out/${tag}_flow.err; python3 tools/bench_summary.py gpurun_out/${tag}_flow.json; tail -5 gpurun_out/${tag}_flow.err ;;
cfg4test) ( time timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k cfg4_full --durations=5 ) > gpurun_out/${tag}_cfg4test.log 2>&1; tail -12 gpurun_out/${tag}_cfg4test.log ;;
qbench) ( time timeout 900 python bench.py --workload cfg3 --no-cpu-baseline --no-hard --no-gaf --steps 10 ) > gpurun_out/${tag}_qbench.json 2> gpurun_out/${tag}_qbench.err; python3 tools/bench_summary.py gpurun_out/${tag}_qbench.json; tail -3 gpurun_out/${tag}_qbench.err ;;
qcovshapes) for sh in 14 21 22 41 42; do PANTAX_COV_SHAPE=$sh timeout 600 python bench.py --no-cpu-baseline --no-hard --no-gaf --steps 5 --warmup 3 > gpurun_out/${tag}_cov$sh.json 2> gpurun_out/${tag}_cov$sh.err; echo "shape $sh"; python3 tools/bench_summary.py gpurun_out/${tag}_cov$sh.json | head -3; done ;;
covshapes2) for sh in 14 21 22 41 42; do PANTAX_COV_SHAPE=$sh timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --no-hard --no-gaf --steps 10 --warmup 3 > gpurun_out/${tag}_cov2_$sh.json 2> gpurun_out/${tag}_cov2_$sh.err; echo "cfg2 shape $sh"; python3 tools/bench_summary.py gpurun_out/${tag}_cov2_$sh.json | head -3; done ;;
pmctrio) bash tools/pmc_step.sh cfg3 ${tag}_trio "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES"
     python3 tools/pmc_collect.py gpurun_out/pmc_${tag}_trio cfg3 gpurun_out/${tag}_pmc_trio.json; python3 -c "
import json;d=json.load(open('gpurun_out/${tag}_pmc_trio.json'))
for k in ('trio_block_kernel','coverage_step_kernel','trio_lookup_kernel','scan_chained_kernel','bin_reads_kernel'):
    print(k, {a:(round(b) if isinstance(b,float) else b) for a,b in d['kernels'].get(k,{}).items()})" ;;
qbench2) ( time timeout 900 python bench.py --workload cfg2 --no-cpu-baseline --no-hard --no-gaf --steps 20 ) > gpurun_out/${tag}_qbench2.json 2> gpurun_out/${tag}_qbench2.err; python3 tools/bench_summary.py gpurun_out/${tag}_qbench2.json; tail -3 gpurun_out/${tag}_qbench2.err ;;
covshapes) for sh in 14 21 22 41 42; do PANTAX_COV_SHAPE=$sh timeout 600 python bench.py --no-cpu-baseline --no-hard --no-gaf --steps 5 --warmup 3 > gpurun_out/${tag}_cov$sh.json 2> gpurun_out/${tag}_cov$sh.err; echo "shape $sh"; python3 tools/bench_summary.py gpurun_out/${tag}_cov$sh.json | head -3; done ;;
bench2) ( time timeout 600 python bench.py --workload cfg2 ) > gpurun_out/${tag}_bench_cfg2.json 2> gpurun_out/${tag}_bench_cfg2.err; tail -c 600 gpurun_out/${tag}_bench_cfg2.json; tail -3 gpurun_out/${tag}_bench_cfg2.err ;;
qbench4) ( time timeout 900 python bench.py --workload cfg4 --no-cpu-baseline --no-hard --no-gaf --steps 5 ) > gpurun_out/${tag}_qbench4.json 2> gpurun_out/${tag}_qbench4.err; python3 tools/bench_summary.py gpurun_out/${tag}_qbench4.json; tail -3 gpurun_out/${tag}_qbench4.err ;;
masktests) ( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "row_pipelines or strain_profiling or thousand or more_than_64 or literal or cfg2" ) > gpurun_out/${tag}_masktests.log 2>&1; tail -8 gpurun_out/${tag}_masktests.log ;;
qbench4walk) ( time PANTAX_MASK=walk timeout 900 python bench.py --workload cfg4 --no-cpu-baseline --no-hard --no-gaf --steps 5 ) > gpurun_out/${tag}_qbench4walk.json 2> gpurun_out/${tag}_qbench4walk.err; python3 tools/bench_summary.py gpurun_out/${tag}_qbench4walk.json | head -4 ;;
hardleg) ( time timeout 600 python bench.py --workload cfg2 --no-cpu-baseline --no-gaf --steps 5 ) > gpurun_out/${tag}_hardleg.json 2> gpurun_out/${tag}_hardleg.err; python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/${tag}_hardleg.json') if l.startswith('{')][-1]); h=d.get('pao_hard') or d.get('hard') or {}
print('hard lad ms/species', h.get('lad_kernels_ms_per_species'), 'iters', h.get('iters'), 'step', h.get('ms_per_step_all_launches_bracketed'))"; tail -2 gpurun_out/${tag}_hardleg.err ;;
lptests) ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pao_solve or strain_profiling or more_than_64 or row_pipelines or literal or concurrent" ) > gpurun_out/${tag}_lptests.log 2>&1; tail -6 gpurun_out/${tag}_lptests.log ;;
sorttests) ( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "sort or row_pipelines or strain_profiling or thousand or cfg2 or cfg3" ) > gpurun_out/${tag}_sorttests.log 2>&1; tail -6 gpurun_out/${tag}_sorttests.log ;;
prefetchtest) ( time timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "prefetch or step" ) > gpurun_out/${tag}_prefetchtest.log 2>&1; tail -5 gpurun_out/${tag}_prefetchtest.log ;;
gafleg) for wl in cfg4 cfg3; do timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-hard --steps 4 > gpurun_out/${tag}_gafleg_$wl.json 2>/dev/null; python3 tools/bench_summary.py gpurun_out/${tag}_gafleg_$wl.json | grep -E "^value|^gaf" | cut -c1-330; done ;;
hugetests) ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 -k "huge or more_than_64 or beyond_64 or batch_equals or wide or row_pipelines" ) > gpurun_out/${tag}_hugetests.log 2>&1; tail -25 gpurun_out/${tag}_hugetests.log ;;
trace4) bash tools/kernel_trace.sh cfg4 ${tag}_cfg4 4 ;;
qbench5) ( time timeout 900 python bench.py --workload cfg5_share --no-cpu-baseline --no-hard --no-gaf --steps 5 ) > gpurun_out/${tag}_qbench5.json 2> gpurun_out/${tag}_qbench5.err; python3 tools/bench_summary.py gpurun_out/${tag}_qbench5.json; tail -3 gpurun_out/${tag}_qbench5.err ;;
trace5) bash tools/kernel_trace.sh cfg5_share ${tag}_cfg5share 4 ;;
pmc4) bash tools/pmc_step.sh cfg4 ${tag}_cfg4 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCP_TCC_READ_REQ_sum"
     python3 tools/pmc_collect.py gpurun_out/pmc_${tag}_cfg4 cfg4 gpurun_out/${tag}_pmc_cfg4.json ;;
trace) bash tools/kernel_trace.sh cfg3 ${tag}_cfg3 6 ;;
trace2) bash tools/kernel_trace.sh cfg2 ${tag}_cfg2 10 ;;
pmc) bash tools/pmc_step.sh cfg3 ${tag}_cfg3 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCP_TCC_READ_REQ_sum"
     python3 tools/pmc_collect.py gpurun_out/pmc_${tag}_cfg3 cfg3 gpurun_out/${tag}_pmc_cfg3.json ;;
pmc2) bash tools/pmc_step.sh cfg2 ${tag}_cfg2 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
     python3 tools/pmc_collect.py gpurun_out/pmc_${tag}_cfg2 cfg2 gpurun_out/${tag}_pmc_cfg2.json ;;
esac
done
