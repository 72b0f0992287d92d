#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in cfg4 refdb; do
timeout 1200 python tools/step_probe.py $wl 3 cov_clean_async=0,2,4 only=zero_fill 2>&1 | tail -6 | cut -c1-200
done
PANTAX_COV_CLEAN_ASYNC=4 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
