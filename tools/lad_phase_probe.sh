#!/bin/bash
# Where a pivot of the LAD solver spends its time: runs a probe with the -DLAD_PROFILE build of the library
# (pantax_amd/lib_prof/libpantax_hip.so, built HERE beforehand by `make -C pantax_amd/csrc OUT=../lib_prof EXTRA=-DLAD_PROFILE`),
# selected through PANTAX_HIP_LIB (pantax_amd/_ffi.py): the product library is never touched.  usage: lad_phase_probe.sh <probe.py> [args]
cd $GRAFT_REPO_ROOT
PANTAX_HIP_LIB=$PWD/pantax_amd/lib_prof/libpantax_hip.so timeout 900 python "$@" 2>&1 | tail -60
