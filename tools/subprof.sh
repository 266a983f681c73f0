#!/bin/bash
# Profiling build with the sub-phase marks (-DRSIM_SUBPROF: slots x0..x9 inside the solver and the OSC controller) -> robosuite_amd/librsim_hip_prof.so.
# Use: tools/subprof.sh && gpurun -- 'RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so python tools/phase_profile_env.py 3 0 1'
set -e
# tools/subprof.sh mpr: MPR outcome counters (-DRSIM_MPRSTAT) instead of sub-phase marks
if [ "$1" = mpr ]; then FLAG=-DRSIM_MPRSTAT; else FLAG=-DRSIM_SUBPROF=${1:-1}; fi
R=/root/repo; T=/tmp/profbuild; rm -rf $T; mkdir -p $T/robosuite_amd/csrc $T/include
cp $R/robosuite_amd/csrc/{rsim_step.hip,rsim_api.cpp,rsim_internal.h,Makefile} $T/robosuite_amd/csrc/; cp $R/include/rsim.h $T/include/
make -s -j -C $T/robosuite_amd/csrc OUT=$R/robosuite_amd/librsim_hip_prof.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero $FLAG" 2>&1 | grep -E " error |Error" || true
ls -la $R/robosuite_amd/librsim_hip_prof.so
