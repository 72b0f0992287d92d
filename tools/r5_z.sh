#!/bin/bash
# round 5: shapes of the short-read coverage kernel with the round-ahead loads; the suite after the seam's grouping change
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 900 python tools/stage_probe.py cfg4 4 covf_shape=2823,2423,1823,4423,2825 > gpurun_out/r5_z_stage_probe_cfg4.txt 2>&1; cut -c1-200 gpurun_out/r5_z_stage_probe_cfg4.txt
timeout 900 python tools/stage_probe.py cfg3 4 covf_shape=2423,2823,1823,4423 > gpurun_out/r5_z_stage_probe_cfg3.txt 2>&1; cut -c1-200 gpurun_out/r5_z_stage_probe_cfg3.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_z_pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r5_z_pytest.log
