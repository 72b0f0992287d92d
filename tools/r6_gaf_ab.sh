#!/bin/bash
# the GAF readers with the 8-byte register window (TxtWin) against the byte-by-byte ones (lib_ablgafold)
cd $GRAFT_REPO_ROOT
for v in old new; do
  if [ $v = old ]; then export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_ablgafold/libpantax_hip.so; else unset PANTAX_HIP_LIB; fi
  echo "== $v"; timeout 900 python tools/gaf_filter_bench.py 100000 2>&1 | grep -v "^generated" | head -8
done
unset PANTAX_HIP_LIB
timeout 900 python -m pytest tests -x -q -m gpu -k "gaf or seam or file or filter" 2>&1 | tail -3
