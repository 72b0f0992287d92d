#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in cfg4 cfg3 cfg5_share; do
    timeout 600 python tools/step_probe.py $wl 3 ncs_prefix_min=48,0 only=node_cov 2>&1 | tail -4 | cut -c1-200
done
