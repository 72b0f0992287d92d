#!/bin/bash
# the LPs of more than 64 columns (W in global memory): time per solve and per pivot on the committed fixtures, then the tests of those paths
cd $GRAFT_REPO_ROOT
timeout 900 python tools/huge_columns_probe.py 2>&1 | tail -11
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide or huge or 64 or pao" 2>&1 | tail -2
STRESS_WIDE=1 timeout 1500 python tools/stress.py 40 780000 2>&1 | tail -1
