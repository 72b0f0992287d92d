#!/bin/bash
# A/B of the -DCOV_NO_PF2 build against the product library (coverage_fast_kernel with the read / slot records requested a round ahead)
cd $GRAFT_REPO_ROOT
for wl in cfg3 refdb cfg4; do
  for v in prod nopf2 prod nopf2; do
    if [ $v = nopf2 ]; then export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_ablnopf2/libpantax_hip.so; else unset PANTAX_HIP_LIB; fi
    echo "$wl $v: $(timeout 600 python tools/stage_probe.py $wl 3 2>&1 | tail -1 | cut -c1-300)"
  done
done
unset PANTAX_HIP_LIB
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
