#!/bin/bash
# PMC passes over the coverage kernel (one counter group per run, --pmc only; see MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "FETCH_SIZE TCC_ATOMIC_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc/g$i -o g$i -- python3 tools/cov_driver.py 2 > gpurun_out/pmc/g$i.log 2>&1
done
ls -R gpurun_out/pmc | head -40
