#!/bin/bash
# round 6 final records, part C: the N = 8 bench flow as a dry run on ONE GPU (eight ranks share the device, exchange over gloo): tables == N = 1,
# per-rank host memory, ingest_route_ms
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PANTAX_DEV_CACHE_GB=24      # eight processes on one device: every rank caps its cache of released blocks
bash tools/strong_dry_run.sh cfg4 8 r6_dry8 2>&1 | tail -16
python3 - <<'PY'
import json
a = [json.loads(l) for l in open('gpurun_out/r6_dry8_strong8.json') if l.startswith('{')][0]
c = a["config"]
print({k: c.get(k) for k in ("abundance_l1_vs_oracle", "abundance_l1_species_checked", "abundance_l1_error", "ingest_route_ms", "ms_per_step_ranks_min_max", "rccl_ranks", "ranks_seen", "parallelism", "species_per_gpu")})
PY
