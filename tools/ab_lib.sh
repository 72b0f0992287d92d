#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: $AB_OLD (default pantax_amd/lib/ab_old.so) against the product
# library.  The builds are selected through PANTAX_HIP_LIB (pantax_amd/_ffi.py): the product file is never overwritten.
# usage: ab_lib.sh [bench args]
set -e
cd $GRAFT_REPO_ROOT
L=$PWD/pantax_amd/lib
OLD=${AB_OLD:-$L/ab_old.so}
[ -f "$OLD" ] || { echo "ab_lib.sh: $OLD not found (copy the build to compare against there)"; exit 1; }
for round in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then lib=$OLD; else lib=$L/libpantax_hip.so; fi
    PANTAX_HIP_LIB=$lib timeout 600 python bench.py --workload ${AB_WORKLOAD:-cfg3} --no-cpu-baseline --no-hard --no-gaf --steps 10 "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    echo "$v: $(python3 tools/bench_summary.py gpurun_out/ab_$v.json | head -1)"
  done
done
