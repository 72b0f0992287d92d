#!/bin/bash
# where ssn_hist_kernel's time goes: the -DSSN_ABLATE build (make -C pantax_amd/csrc OUT=../lib_ablssn EXTRA=-DSSN_ABLATE), parts left out one at a time
cd $GRAFT_REPO_ROOT
export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_ablssn/libpantax_hip.so
timeout 1200 python tools/step_probe.py ${1:-cfg4} 3 ssn_ablate=0,1,2,4,8,16,32,63 only=ssn_ 2>&1 | tail -9 | cut -c1-330
