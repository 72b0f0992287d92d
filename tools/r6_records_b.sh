#!/bin/bash
# round 6 final records, part B: kernel traces, PMC passes (separate --pmc runs), the traced bench command
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in cfg4 cfg3 cfg5_share; do
  bash tools/kernel_trace.sh $wl r6rec_$wl 6 > gpurun_out/r6_records_trace_$wl.log 2>&1; tail -2 gpurun_out/r6_records_trace_$wl.log | cut -c1-200
done
G1="FETCH_SIZE"; G2="WRITE_SIZE"
G3="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
G4="TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCP_TCC_READ_REQ_sum"
G5="SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES"
for wl in cfg4 cfg5_share cfg3; do
  bash tools/pmc_step.sh $wl r6rec_$wl "$G1" "$G2" "$G3" "$G4" "$G5" > gpurun_out/r6_records_pmc_$wl.log 2>&1; cut -c1-100 gpurun_out/r6_records_pmc_$wl.log
  python3 tools/pmc_collect.py gpurun_out/pmc_r6rec_$wl $wl gpurun_out/r06_pmc_$wl.json 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r6rec_bench -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gaf --no-seam --no-hard --no-l1 > gpurun_out/r6_records_bench_traced.json 2> gpurun_out/r6_records_bench_traced.err
db=$(find gpurun_out/prof_r6rec_bench -name '*.db' | head -1); [ -n "$db" ] && python3 tools/rocpd_summary.py $db > gpurun_out/r06_bench_cmd_cfg4_kernel_stats.txt
head -12 gpurun_out/r06_bench_cmd_cfg4_kernel_stats.txt | cut -c1-150
find gpurun_out -name '*.db' -size +20M -delete 2>/dev/null
du -sh gpurun_out | tail -1
