#!/bin/bash
# rocprofv3 kernel trace (+stats) of N resident steps of one bench workload; summary via tools/rocpd_summary.py.
# usage: kernel_trace.sh <workload> <tag> [steps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
wl=$1; tag=$2; n=${3:-6}
timeout 1200 rocprofv3 --kernel-trace --stats -d gpurun_out/trace_$tag -o $tag -- python3 tools/step_driver.py $wl $n > gpurun_out/trace_$tag.log 2>&1
echo "trace rc=$?"
db=$(find gpurun_out/trace_$tag -name '*.db' | head -1)
if [ -n "$db" ]; then python3 tools/rocpd_summary.py $db > gpurun_out/trace_${tag}_kernel_stats.txt; head -40 gpurun_out/trace_${tag}_kernel_stats.txt; else find gpurun_out/trace_$tag | head; fi
