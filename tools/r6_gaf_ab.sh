#!/bin/bash
# the tokenizer with the 8-byte register window against the byte-by-byte one (lib_ablgafold), GAF text on disk -> resident reads
cd $GRAFT_REPO_ROOT
for v in old new old new; do
  if [ $v = old ]; then export PANTAX_HIP_LIB=$PWD/pantax_amd/lib_ablgafold/libpantax_hip.so; else unset PANTAX_HIP_LIB; fi
  echo "$v: $(timeout 900 python tools/gaf_ingest_probe.py ${1:-cfg4} 3 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-600)"
done
unset PANTAX_HIP_LIB
timeout 900 python -m pytest tests -x -q -m gpu -k "gaf or seam or file" 2>&1 | tail -3
