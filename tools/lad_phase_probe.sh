#!/bin/bash
# Where a pivot of the LAD solver spends its time: runs a probe with the -DLAD_PROFILE build of the library
# (pantax_amd/lib_prof/libpantax_hip.so, built HERE beforehand by `make -C pantax_amd/csrc OUT=../lib_prof EXTRA=-DLAD_PROFILE`)
# swapped in, then puts the product library back.  usage: lad_phase_probe.sh <probe.py> [args]
cd $GRAFT_REPO_ROOT
L=pantax_amd/lib
cp $L/libpantax_hip.so $L/keep_product.so
cp pantax_amd/lib_prof/libpantax_hip.so $L/libpantax_hip.so
timeout 900 python "$@" 2>&1 | tail -60
cp $L/keep_product.so $L/libpantax_hip.so
