#!/bin/bash
# round 6: random configurations against the oracle on the round's last code: default, every group through the long-walk kernel, round 5's long-walk
# kernel, highs semantics, the self-cleaning arena
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
run() { name=$1; n=$2; seed=$3; shift 3; env "$@" timeout 2400 python tools/stress.py $n $seed > gpurun_out/r6_stress_$name.log 2>&1; echo "$name: $(tail -1 gpurun_out/r6_stress_$name.log | cut -c1-200)"; grep -c "^FAIL" gpurun_out/r6_stress_$name.log; grep "^FAIL" gpurun_out/r6_stress_$name.log | head -3 | cut -c1-300; }
if [ "$1" = "final" ]; then   # the round's last code: other seeds; the side-stream arena fill forced on (these dbs are below its 1-GiB threshold)
run default ${2:-1000} ${7:-710000} X=1
run cleanasync ${3:-400} 720000 PANTAX_COV_CLEAN_ASYNC=1 PANTAX_COV_ARENA_VERIFY=1
run general ${4:-200} 730000 PANTAX_COV_GENERAL=1
run highs ${5:-200} 740000 STRESS_HIGHS=1
run wide 30 750000 STRESS_WIDE=1
run refdb ${6:-60} 760000 STRESS_REFDB=1
exit 0
fi
run default 1500 610000 X=1
run general 300 620000 PANTAX_COV_GENERAL=1
run longstep 300 630000 PANTAX_COV_LONG=step
run highs 400 640000 STRESS_HIGHS=1
run selfclean 300 650000 PANTAX_COV_SELF_CLEAN=1 PANTAX_COV_ARENA_VERIFY=1
run wide 40 660000 STRESS_WIDE=1
