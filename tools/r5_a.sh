#!/bin/bash
# round 5, first GPU call: box facts, the GPU suite, the file seam at cfg3 / cfg4
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
{ df -h /tmp /dev/shm . ; free -g; nproc; ulimit -n; } > gpurun_out/r5_box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_a_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_a_pytest.log
tail -5 gpurun_out/r5_a_pytest.log
timeout 300 python tools/seam_bench.py cfg3 > gpurun_out/r5_a_seam_cfg3.log 2>&1
tail -3 gpurun_out/r5_a_seam_cfg3.log
timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_a_seam_cfg4.log 2>&1
tail -60 gpurun_out/r5_a_seam_cfg4.log
