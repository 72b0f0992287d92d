#!/bin/bash
# PMC passes over the resident step of one bench workload: one counter group per run, --pmc only (no trace domains),
# as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes: FETCH_SIZE and WRITE_SIZE cannot
# share a pass.  Every pass is wrapped in `timeout`.  usage: pmc_step.sh <workload> <tag> <group> [<group> ...]
# (a group is a space-separated counter list in quotes); results -> gpurun_out/pmc_<tag>/g<i>/, summarised by
# tools/pmc_collect.py into profiles/r<round>_pmc_<workload>.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=$1; tag=$2; shift 2
# the set is generated again in every pass (native generator: seconds)
TMO=${PMC_TIMEOUT:-400}; if [ "$wl" = cfg4 ]; then TMO=${PMC_TIMEOUT:-1200}; fi
mkdir -p gpurun_out/pmc_$tag
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout $TMO rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_$tag/g$i -o g$i -- python3 tools/step_driver.py $wl 2 > gpurun_out/pmc_$tag/g$i.log 2>&1
  echo "group $i ($grp) rc=$?"
done
