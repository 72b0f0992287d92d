#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
STRESS_REFDB=1 timeout 2400 python tools/stress.py 300 760000 > gpurun_out/r6_stress_refdb.log 2>&1
tail -1 gpurun_out/r6_stress_refdb.log; grep "^FAIL" gpurun_out/r6_stress_refdb.log | head -5 | cut -c1-300
STRESS_REFDB=1 PANTAX_NCS_NO_PREFIX=1 timeout 2400 python tools/stress.py 100 770000 > gpurun_out/r6_stress_refdb_noprefix.log 2>&1
tail -1 gpurun_out/r6_stress_refdb_noprefix.log
