#!/bin/bash
# A/B of two builds of the library on ONE box: pantax_amd/lib/ab_old.so against the current one, alternating.  usage: ab_lib.sh [bench args]
cd $GRAFT_REPO_ROOT
L=pantax_amd/lib
cp $L/libpantax_hip.so $L/ab_new.so
for round in 1 2 3; do
  for v in old new; do
    cp $L/ab_$v.so $L/libpantax_hip.so
    timeout 600 python bench.py --workload ${AB_WORKLOAD:-cfg3} --no-cpu-baseline --no-hard --no-gaf --steps 10 "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    echo "$v: $(python3 tools/bench_summary.py gpurun_out/ab_$v.json | head -1)"
  done
done
cp $L/ab_new.so $L/libpantax_hip.so
