#!/bin/bash
# round 5, eighteenth GPU call: chunk size / crew size / NUMA binding of the staged GAF upload (no host columns asked for: the file seam's load)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1500 python tools/gaf_ingest_probe.py cfg4 5 numa_bind=0 stage_ch_mb=128 stage_ch_mb=32 stage_threads=48 stage_threads=24 > gpurun_out/r5_r_gaf_ingest.txt 2>&1
cat gpurun_out/r5_r_gaf_ingest.txt | tail -12
