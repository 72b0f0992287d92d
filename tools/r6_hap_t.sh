#!/bin/bash
# a9 statistics of species of 17 .. 64 haplotypes: lane-owned accumulators (default) against the slabs of 16 (-DHS_NO_TRANSPOSE build)
cd $GRAFT_REPO_ROOT
for v in prod notr prod notr; do
if [ $v = prod ]; then unset PANTAX_HIP_LIB; else export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_ablnotr/libpantax_hip.so; fi
echo "$v: $(timeout 600 python tools/step_probe.py cfg5_share 3 only=hap_ 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-200)"
done
unset PANTAX_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg5" 2>&1 | tail -2
timeout 900 python tools/stress.py 300 790000 2>&1 | tail -1
