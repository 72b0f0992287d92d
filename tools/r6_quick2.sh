#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/stress.py 1 712253 2>&1 | tail -2 | cut -c1-300
