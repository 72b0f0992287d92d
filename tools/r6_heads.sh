#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in cfg5_share cfg3; do
    echo "$wl: $(timeout 600 python tools/step_probe.py $wl 3 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-900)"
done
