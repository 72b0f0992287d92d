#!/bin/bash
# PMC passes over the coverage kernel (one counter group per run, --pmc only; see MI355X_MICROARCH.md).
# Every pass is wrapped in `timeout`: a TA_* group aborted rocprofv3 and hung once.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
i=${1:-10}
shift
for grp in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc/g$i -o g$i -- python3 tools/cov_driver.py 2 > gpurun_out/pmc/g$i.log 2>&1
  echo "group $i ($grp) rc=$?"
done
