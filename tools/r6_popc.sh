#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "reference_db" 2>&1 | tail -3
echo "refdb: $(timeout 600 python tools/stage_probe.py refdb 3 ncs_no_prefix=0,1 2>&1 | tail -2 | cut -c1-300)"
echo "refdb step: $(timeout 600 python tools/step_probe.py refdb 3 only=node_cov 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
