#!/bin/bash
# round 5, fourteenth GPU call: LDS window sizes of the short-read coverage kernel (occupancy: 23.5 KB per workgroup = 6 waves per SIMD)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 900 python tools/stage_probe.py cfg4 4 covf_shape=283,2823,2825,282,2423,243 > gpurun_out/r5_n_stage_probe_cfg4.txt 2>&1; cat gpurun_out/r5_n_stage_probe_cfg4.txt | cut -c1-400
timeout 900 python tools/stage_probe.py cfg3 4 covf_shape=243,2423,2425,242,283,2823 > gpurun_out/r5_n_stage_probe_cfg3.txt 2>&1; cat gpurun_out/r5_n_stage_probe_cfg3.txt | cut -c1-400
