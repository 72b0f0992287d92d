#!/bin/bash
# the default bench with the pipeline trace
cd $GRAFT_REPO_ROOT
PANTAX_PIPE_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hard --no-gaf > gpurun_out/pipe_trace.json 2> gpurun_out/pipe_trace.err
grep "pipelined\|\[bench\]" gpurun_out/pipe_trace.err | cut -c1-300 | sed -n 5,7p
python tools/bench_summary.py gpurun_out/pipe_trace.json | head -8
