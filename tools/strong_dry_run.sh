#!/bin/bash
# Dry run of bench.py's DEFAULT N > 1 flow (strong scaling of one set, --gpus N) on a one-GPU box: N ranks share the GPU and the
# exchange goes over gloo (PANTAX_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device).  What it checks: every rank generates
# only its slice of the reads, bins it, the counts are all-reduced, the reads travel to the owner of their species, every rank
# uploads only the graphs it owns, one all-reduce per step -- and rank 0's tables equal the one-process run of the same set.
# The N ranks are started by bench.py itself (python bench.py --gpus N: what the driver calls).  Prints every rank's peak host RSS.  usage: strong_dry_run.sh [workload=cfg3] [ranks=2] [tag=dry]
cd $GRAFT_REPO_ROOT
wl=${1:-cfg3}; n=${2:-2}; tag=${3:-dry}
timeout 300 python -m pytest tests/test_gpu_route.py -m gpu -x -q -k rccl 2>&1 | tail -3
export PANTAX_BENCH_BACKEND=gloo PANTAX_BENCH_RSS=1
timeout 1500 python bench.py --gpus $n --steps 5 --warmup 2 --workload $wl > gpurun_out/${tag}_strong$n.json 2> gpurun_out/${tag}_strong$n.err
echo rc=$?; grep -h "peak host RSS" gpurun_out/${tag}_strong$n.err; grep -v "^\[W\|amdgpu.ids\|OMP_NUM\|\*\*\*\*\|peak host RSS" gpurun_out/${tag}_strong$n.err | tail -3
python3 tools/bench_summary.py gpurun_out/${tag}_strong$n.json | head -3
unset PANTAX_BENCH_BACKEND
timeout 1500 python bench.py --steps 5 --warmup 2 --workload $wl --no-cpu-baseline --no-hard --no-gaf > gpurun_out/${tag}_one.json 2> gpurun_out/${tag}_one.err
python3 - <<PY
import json
a = [json.loads(l) for l in open('gpurun_out/${tag}_strong$n.json') if l.startswith('{')][0]
b = [json.loads(l) for l in open('gpurun_out/${tag}_one.json') if l.startswith('{')][0]
print("N=$n :", a["value"], a["ms_per_step"], a["scaling"], a["n_gpus"], a["config"]["exchange"], a["config"].get("ingest_route_ms"))
print("N=1 :", b['value'], b['ms_per_step'], b['scaling'])
print("tables equal:", a['result'] == b['result'], a['result']['n_species_rows'], a['result']['n_strain_rows'], a['result']['top_strains'][:2])
PY
