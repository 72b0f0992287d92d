#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_seam_sharded.py tests/test_gpu_configs.py -x -q -m gpu -k "cli or sharded or rccl or reference_db" 2>&1 | tail -5
for th in 32 64; do echo "== stage_threads $th"; PANTAX_STAGE_THREADS=$th bash tools/r6_seam_only.sh cfg4 2>&1 | grep -E "files_to_tables_warm|gaf_load_s|upload_staged" | cut -c1-240 | head -4; done
