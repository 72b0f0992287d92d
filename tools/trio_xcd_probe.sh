#!/bin/bash
# XCD-contiguous chunks in the two kernels of the index rebuild (PANTAX_TRIO_XCD: bit 0 visit, bit 1 rows), cfg4 and the cfg5 share
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
for wl in cfg4 cfg5_share; do for x in 0 1 2 3; do
echo "== $wl PANTAX_TRIO_XCD=$x"
PANTAX_TRIO_XCD=$x timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 5 > gpurun_out/xcd_${wl}_$x.json 2>/dev/null
python3 tools/bench_summary.py gpurun_out/xcd_${wl}_$x.json | grep -E "^value|^kernels" | cut -c1-260
done; done
