#!/bin/bash
# round 5, twentieth GPU call: dry run of the N > 1 bench flow (2 and 4 ranks share the GPU over gloo), cfg3 and cfg4
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/strong_dry_run.sh cfg3 2 r5t_cfg3 2>&1 | tail -12
python - <<'PY'
import json
a = [json.loads(l) for l in open('gpurun_out/r5t_cfg3_strong2.json') if l.startswith('{')][0]
c = a["config"]
print({k: c.get(k) for k in ("abundance_l1_vs_oracle", "abundance_l1_species_checked", "abundance_l1_error", "ingest_route_ms", "ms_per_step_ranks_min_max", "rccl_ranks", "ranks_seen", "parallelism")})
print(a.get("cpu_baseline"), a.get("roofline", {}).get("kernel"))
PY
bash tools/strong_dry_run.sh cfg4 4 r5t_cfg4 2>&1 | tail -8
python - <<'PY'
import json
a = [json.loads(l) for l in open('gpurun_out/r5t_cfg4_strong4.json') if l.startswith('{')][0]
c = a["config"]
print({k: c.get(k) for k in ("abundance_l1_vs_oracle", "abundance_l1_species_checked", "abundance_l1_error", "ingest_route_ms", "ms_per_step_ranks_min_max", "rccl_ranks", "ranks_seen", "parallelism")})
PY
