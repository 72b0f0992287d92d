#!/bin/bash
# round 5, fifth GPU call: parity with first-fit visit groups + lane-register statistics; A/B of the row layout (SoA u16 owner vs packed {len, owner});
# stand-alone rebuild times; counter bytes of the rebuild kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_e_pytest1.log 2>&1
echo "pytest exit $?" >> gpurun_out/r5_e_pytest1.log
tail -4 gpurun_out/r5_e_pytest1.log
V1=$PWD/pantax_amd/lib_v1/libpantax_hip.so
for wl in cfg4 cfg5_share; do
  echo "== trio_probe $wl product"; timeout 300 python tools/trio_probe.py $wl 5 2>&1 | tail -1
  echo "== trio_probe $wl packed";  PANTAX_HIP_LIB=$V1 timeout 300 python tools/trio_probe.py $wl 5 2>&1 | tail -1
  echo "== trio_probe $wl product rows_u=2"; PANTAX_ROWS_U=2 timeout 300 python tools/trio_probe.py $wl 5 2>&1 | tail -1
done
for lib in product packed; do
  for wl in cfg4 cfg5_share; do
    if [ $lib = packed ]; then export PANTAX_HIP_LIB=$V1; else unset PANTAX_HIP_LIB; fi
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 10 --detail-file gpurun_out/r5_e_detail_${wl}_$lib.json > gpurun_out/r5_e_bench_${wl}_$lib.json 2> gpurun_out/r5_e_bench_${wl}_$lib.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r5_e_bench_${wl}_$lib.json").read().strip().splitlines()[-1])
    print("$wl $lib", "ms_per_step", round(d["ms_per_step"], 3), "resident-index", round(d["config"]["ms_per_step_trio_index_resident"], 3), d["roofline"]["kernel"], d["roofline"]["avg_ms"], round(d["roofline"]["frac"], 3))
    print("   ", d["kernels_ms_per_step"])
except Exception as e:
    print("$wl $lib: no line", e); print(open("gpurun_out/r5_e_bench_${wl}_$lib.err").read()[-1500:])
PY
  done
done
unset PANTAX_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_r5e/g$i -o g$i -- python3 tools/trio_probe.py cfg4 2 > gpurun_out/pmc_r5e_g$i.log 2>&1
  echo "pmc group $i ($grp) rc=$?"
done
python - <<'PY'
import csv, glob, collections
for g in sorted(glob.glob("gpurun_out/pmc_r5e/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(g)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[(k, r["Counter_Name"])][0] += 1
        acc[(k, r["Counter_Name"])][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
        print(g.split("/")[2], k, c, "launches", n, "per launch %.3f GB (raw value x 1 KB?)" % (v / n / 1e6))
PY
