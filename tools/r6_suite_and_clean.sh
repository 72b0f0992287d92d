#!/bin/bash
# round 6: whole GPU suite (arena verification on), more shapes of the long-walk kernel, cfg4 step with / without the self-cleaning arena
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_suite.txt; cat gpurun_out/r6_suite.txt
timeout 900 python tools/stage_probe.py cfg5_share 3 covl_shape=2834,21634,21234,2844,21644,2836,2832,2824,4834,1834 > gpurun_out/r6_cov_probe_c.txt 2>gpurun_out/r6_cov_probe_c.err; cut -c1-200 gpurun_out/r6_cov_probe_c.txt
for sc in 1 0; do
PANTAX_COV_SELF_CLEAN=$sc timeout 900 python bench.py --no-cpu-baseline --no-gaf --no-hard --no-l1 --steps 10 > gpurun_out/r6_cfg4_clean$sc.json 2>gpurun_out/r6_cfg4_clean$sc.err
echo "== self clean $sc"; python3 tools/bench_summary.py gpurun_out/r6_cfg4_clean$sc.json | grep -E "^value|^kernels" | cut -c1-900; tail -2 gpurun_out/r6_cfg4_clean$sc.err | cut -c1-300
done
