#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "reference_db or thousand or single_call or mixed_database" 2>&1 | tail -4
bash tools/r6_bench_nochild.sh
