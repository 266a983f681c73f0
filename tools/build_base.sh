#!/bin/bash
# Build robosuite_amd/librsim_hip_base.so from the last COMMITTED sources (A/B baseline for tools/ab_lib.sh).
set -e
R=/root/repo; T=/tmp/basebuild; rm -rf $T; mkdir -p $T/robosuite_amd/csrc $T/include
for f in robosuite_amd/csrc/rsim_step.hip robosuite_amd/csrc/rsim_api.cpp robosuite_amd/csrc/rsim_internal.h robosuite_amd/csrc/Makefile include/rsim.h; do git -C $R show ${1:-HEAD}:$f > $T/$f; done
make -s -C $T/robosuite_amd/csrc OUT=$R/robosuite_amd/librsim_hip_base.so 2>&1 | grep -E " error |Error" || true
ls -la $R/robosuite_amd/librsim_hip_base.so
