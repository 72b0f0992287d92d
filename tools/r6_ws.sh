#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "cfg5_share: $(timeout 600 python tools/step_probe.py cfg5_share 3 only=walk_sum 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-200)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long or mixed or sample" 2>&1 | tail -2
