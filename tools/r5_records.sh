#!/bin/bash
# round 5, twenty-first GPU call: the final records -- full suite, the default bench run, the other workloads, traces, PMC passes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_records_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/r5_records_pytest.log
( time timeout 1500 python bench.py --detail-file gpurun_out/r5_records_detail_default.json > gpurun_out/r5_records_bench_default.json 2> gpurun_out/r5_records_bench_default.err ) 2> gpurun_out/r5_records_bench_default.time
echo "default bench exit $?"; tail -3 gpurun_out/r5_records_bench_default.time; tail -c 1500 gpurun_out/r5_records_bench_default.json
for wl in cfg3 cfg2 cfg5_share; do
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-hard --steps 10 --detail-file gpurun_out/r5_records_detail_${wl}.json > gpurun_out/r5_records_bench_${wl}.json 2> gpurun_out/r5_records_bench_${wl}.err
  echo "$wl exit $?"; tail -c 600 gpurun_out/r5_records_bench_${wl}.json
done
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --detail-file gpurun_out/r5_records_detail_cfg5.json > gpurun_out/r5_records_bench_cfg5.json 2> gpurun_out/r5_records_bench_cfg5.err
echo "cfg5 exit $?"; tail -c 2500 gpurun_out/r5_records_bench_cfg5.json
for wl in cfg4 cfg3 cfg5_share; do
  bash tools/kernel_trace.sh $wl r5rec_$wl 6 > gpurun_out/r5_records_trace_$wl.log 2>&1; tail -2 gpurun_out/r5_records_trace_$wl.log | cut -c1-200
done
G1="FETCH_SIZE"; G2="WRITE_SIZE"
G3="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
G4="TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCP_TCC_READ_REQ_sum"
G5="SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES"
for wl in cfg4 cfg5_share cfg3; do
  bash tools/pmc_step.sh $wl r5rec_$wl "$G1" "$G2" "$G3" "$G4" "$G5" > gpurun_out/r5_records_pmc_$wl.log 2>&1; cut -c1-100 gpurun_out/r5_records_pmc_$wl.log
  python3 tools/pmc_collect.py gpurun_out/pmc_r5rec_$wl $wl gpurun_out/r05_pmc_$wl.json 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r5rec_bench -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gaf --no-seam --no-hard --no-l1 > gpurun_out/r5_records_bench_traced.json 2> gpurun_out/r5_records_bench_traced.err
db=$(find gpurun_out/prof_r5rec_bench -name '*.db' | head -1); [ -n "$db" ] && python3 tools/rocpd_summary.py $db > gpurun_out/r05_bench_cmd_cfg4_kernel_stats.txt
head -8 gpurun_out/r05_bench_cmd_cfg4_kernel_stats.txt
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r5rec_cfg5 -o cfg5 -- python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-l1 > gpurun_out/r5_records_cfg5_traced.json 2> gpurun_out/r5_records_cfg5_traced.err
db=$(find gpurun_out/prof_r5rec_cfg5 -name '*.db' | head -1); [ -n "$db" ] && python3 tools/rocpd_summary.py $db > gpurun_out/r05_bench_cmd_cfg5_kernel_stats.txt
head -8 gpurun_out/r05_bench_cmd_cfg5_kernel_stats.txt
find gpurun_out -name '*.db' -size +20M -delete 2>/dev/null
du -sh gpurun_out | tail -1
