#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in cfg4 cfg3; do
echo "$wl: $(timeout 600 python tools/stage_probe.py $wl 4 2>&1 | tail -1 | cut -c1-200)"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
