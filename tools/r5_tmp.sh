#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1700 python tools/stress.py 800 110000 > gpurun_out/r5_stress_d.log 2>&1; echo "stress d exit $?"; tail -1 gpurun_out/r5_stress_d.log | cut -c1-300
PANTAX_COV_GENERAL=1 timeout 900 python tools/stress.py 300 120000 > gpurun_out/r5_stress_covgen.log 2>&1; echo "stress cov_general exit $?"; tail -1 gpurun_out/r5_stress_covgen.log | cut -c1-300
PANTAX_MASK=walk timeout 900 python tools/stress.py 300 130000 > gpurun_out/r5_stress_maskwalk.log 2>&1; echo "stress mask=walk exit $?"; tail -1 gpurun_out/r5_stress_maskwalk.log | cut -c1-300
