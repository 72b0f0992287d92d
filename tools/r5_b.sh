#!/bin/bash
# round 5, second GPU call: the whole GPU suite + the file seam at cfg4 with the fine trace
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 900 python tools/seam_bench.py cfg4 > gpurun_out/r5_b_seam_cfg4.log 2>&1
grep -v '^  "\|^ "' gpurun_out/r5_b_seam_cfg4.log | tail -5
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5_b_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_b_pytest.log
tail -15 gpurun_out/r5_b_pytest.log
