#!/bin/bash
# A/B of the from-GAF-text leg on one box: the product library against another build (PANTAX_HIP_LIB), alternating.  usage: gaf_ab.sh <other lib> [workload]
cd $GRAFT_REPO_ROOT
other=$1; wl=${2:-cfg4}
for k in 1 2; do for lib in product other; do
  if [ $lib = other ]; then export PANTAX_HIP_LIB=$PWD/$other; else unset PANTAX_HIP_LIB; fi
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-l1 --steps 3 --warmup 2 > gpurun_out/gafab_${lib}_$k.json 2> gpurun_out/gafab_${lib}_$k.err
  echo "$lib run $k: $(python3 tools/bench_summary.py gpurun_out/gafab_${lib}_$k.json | grep -E '^gaf' | cut -c1-260)"
done; done
