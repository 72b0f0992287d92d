#!/bin/bash
# round 5: the file seam in groups of species; suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_x_pytest.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/r5_x_pytest.log
