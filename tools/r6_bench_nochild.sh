#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1500 python bench.py --no-cpu-baseline --no-hard --no-l1 --steps 5 --detail-file gpurun_out/r6_nochild_detail.json > gpurun_out/r6_nochild.json 2> gpurun_out/r6_nochild.err
python3 tools/bench_summary.py gpurun_out/r6_nochild.json | grep -E "^value|^gaf" | cut -c1-600
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r6_nochild_detail.json'))
fs=d['file_seam']
for l in fs['trace']['wd_warm1'].split('\n'):
    if 'upload_staged' in l: print(l[:260])
PY
timeout 1500 python bench.py --workload refdb --steps 5 --no-gaf --no-l1 > gpurun_out/r6_refdb_quick.json 2> gpurun_out/r6_refdb_quick.err
python3 tools/bench_summary.py gpurun_out/r6_refdb_quick.json | grep -E "^value|^kernels|^  " | cut -c1-500
