#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
for t in "" 1 2; do echo "== SEAM_WITH_TORCH=$t"; SEAM_WITH_TORCH=$t bash tools/r6_seam_only.sh cfg4 2>&1 | grep -E "files_to_tables_warm_s_both|gaf_load_s|upload_staged" | cut -c1-240 | head -3; done
