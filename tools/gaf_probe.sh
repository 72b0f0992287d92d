#!/bin/bash
# the from-GAF-text leg alone, with and without the host-side column pruning, on one box.  usage: gaf_probe.sh <workload>
cd $GRAFT_REPO_ROOT
wl=${1:-cfg4}
for pr in 1 0 1 0; do
  PANTAX_GAF_PRUNE=$pr timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-l1 --steps 3 --warmup 2 > gpurun_out/gafprobe_$pr.json 2> gpurun_out/gafprobe_$pr.err
  echo "prune=$pr: $(python3 tools/bench_summary.py gpurun_out/gafprobe_$pr.json | grep '^gaf')"
done
PANTAX_HIP_TRACE=1 PANTAX_GAF_PRUNE=1 timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-l1 --steps 2 --warmup 1 2>&1 >/dev/null | grep -E "upload_pruned|upload_staged|gaf_tokenize\]  *(pieces|join|id-hash|host col|locus|total)" | tail -12
