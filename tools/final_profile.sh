#!/bin/bash
# rocprofv3 kernel trace (+stats) of the default bench command; summary goes to profiles/ via tools/rocpd_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1 -o $1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gaf > gpurun_out/prof_$1.log 2>&1
tail -1 gpurun_out/prof_$1.log | head -c 600
