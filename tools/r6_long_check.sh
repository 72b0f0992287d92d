#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "long or cfg5 or route or sharded or mixed or stress" 2>&1 | tail -4
timeout 600 python tools/stage_probe.py cfg5_share 3 walk_sum_kernel=0,1 2>&1 | tail -3 | cut -c1-300
PANTAX_X=1 timeout 900 python tools/stress.py 400 670000 > gpurun_out/r6_stress_fused.log 2>&1; tail -1 gpurun_out/r6_stress_fused.log
for wl in cfg5_share cfg5; do timeout 1200 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-gaf --steps 5 > gpurun_out/r6_fused_$wl.json 2>gpurun_out/r6_fused_$wl.err; python3 tools/bench_summary.py gpurun_out/r6_fused_$wl.json | grep -E "^value|^kernels" | cut -c1-400; done
