#!/bin/bash
# round 6: the select-only long-walk coverage kernel -- parity tests, then an A/B of its shapes against round 5's kernel at the cfg5 share
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coverage_kernels_agree or long_reads_and_empty or long_walks_with_revisits or short_and_long_reads or binning_and_coverage" 2>&1 | tail -15 > gpurun_out/r6_cov_tests.txt
cat gpurun_out/r6_cov_tests.txt
timeout 900 python tools/stage_probe.py cfg5_share 4 cov_long=,step > gpurun_out/r6_cov_probe_a.txt 2>gpurun_out/r6_cov_probe_a.err; cat gpurun_out/r6_cov_probe_a.txt | cut -c1-600
timeout 1200 python tools/stage_probe.py cfg5_share 3 covl_shape=2234,1234,2244,2434,2238,2224,1434,2834,2230 > gpurun_out/r6_cov_probe_b.txt 2>gpurun_out/r6_cov_probe_b.err; cat gpurun_out/r6_cov_probe_b.txt | cut -c1-400
