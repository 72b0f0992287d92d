#!/bin/bash
# the step's statistics / histogram passes with and without the species the species level dropped (option no_absent_skip)
cd $GRAFT_REPO_ROOT
for wl in refdb cfg4; do
timeout 1200 python tools/step_probe.py $wl 3 no_absent_skip=0,1 only=node_cov 2>&1 | tail -4 | cut -c1-160 | sed "s/^/$wl /"
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
