#!/bin/bash
# how the next step's index rebuild is placed beside the current step (cfg4): default, behind the first filter, behind the whole step, equal priorities
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $2 timeout 600 python bench.py --workload cfg4 --no-cpu-baseline --no-hard --no-gaf --no-l1 --steps 8 > gpurun_out/ov_$1.json 2>/dev/null; python3 tools/bench_summary.py gpurun_out/ov_$1.json | grep -E "^value"; }
run default X=1
run free_at_filter PANTAX_TRIO_FREE=filter
run after_step PANTAX_TRIO_AFTER_STEP=1
run equal_prio PANTAX_STREAM_PRIO=0
run default2 X=1
