#!/bin/bash
# ablations of the long-read coverage kernel at cfg5_share (the -DCOV_ABLATE build: make -C pantax_amd/csrc OUT=../lib_abl EXTRA=-DCOV_ABLATE)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
for ab in 0 1 2 4 8 3 7 15; do
echo "== PANTAX_COV_ABLATE=$ab"
PANTAX_HIP_LIB=$PWD/pantax_amd/lib_abl/libpantax_hip.so PANTAX_COV_ABLATE=$ab timeout 600 python bench.py --workload cfg5_share --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 5 > gpurun_out/cov5_$ab.json 2>gpurun_out/cov5_$ab.err
python3 tools/bench_summary.py gpurun_out/cov5_$ab.json | grep -E "^value|^kernels" | cut -c1-400; tail -1 gpurun_out/cov5_$ab.err | cut -c1-200
done
