#!/bin/bash
# A/B of one environment switch on ONE box, alternating.  usage: ab_env.sh VAR=value [bench args]
cd $GRAFT_REPO_ROOT
kv=$1; shift
for round in 1 2 3; do
  for v in on off; do
    if [ $v = on ]; then env "$kv" timeout 600 python bench.py --workload ${AB_WORKLOAD:-cfg3} --no-cpu-baseline --no-hard --no-gaf --steps 10 "$@" > gpurun_out/abenv_$v.json 2> gpurun_out/abenv_$v.err
    else timeout 600 python bench.py --workload ${AB_WORKLOAD:-cfg3} --no-cpu-baseline --no-hard --no-gaf --steps 10 "$@" > gpurun_out/abenv_$v.json 2> gpurun_out/abenv_$v.err; fi
    echo "$kv $v: $(python3 tools/bench_summary.py gpurun_out/abenv_$v.json | head -1)"
  done
done
