#!/bin/bash
# shapes and ablations of trio_visit_kernel on one box (the ablation build: make -C pantax_amd/csrc OUT=../lib_abl EXTRA=-DTV_ABLATE).
# usage: [TV_SHAPES="4 4,8 8"] [TV_ABLS="0 1 7"] tv_probe.sh <workload>
cd $GRAFT_REPO_ROOT
wl=${1:-cfg4}
IFS=, read -ra SHAPES <<< "${TV_SHAPES:-4 4,8 8,2 8}"
for sh in "${SHAPES[@]}"; do set -- $sh
  echo "U=$1 rounds=$2: $(PANTAX_TV_U=$1 PANTAX_TV_ROUNDS=$2 timeout 600 python tools/trio_probe.py $wl 3 2>&1 | tail -1)"
done
for ab in ${TV_ABLS:-0 1 7}; do
  echo "ablate=$ab: $(PANTAX_HIP_LIB=$PWD/pantax_amd/lib_abl/libpantax_hip.so PANTAX_TV_ABLATE=$ab timeout 600 python tools/trio_probe.py $wl 3 2>&1 | tail -1)"
done
