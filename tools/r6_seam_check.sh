#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "seam or coverage or groups or cfg3 or reference_db or mixed_database or thousand" 2>&1 | tail -6
bash tools/r6_seam_only.sh cfg4 2>&1 | grep -v "upload_segments\|db_upload\]" | cut -c1-220 | head -70
