#!/bin/bash
# The library of another commit as robosuite_amd/librsim_hip_prev.so, for same-box A/Bs (RSIM_LIB=... selects it: backend.py).  Boxes differ by +-15 %, so
# every before / after number in profiles/ comes from ONE gpurun session that runs both builds (tools/gpu_session.sh <tag> ab:<libA>:<libB>).
# Usage: tools/build_prev.sh [commit=HEAD~1]
set -e
c=${1:-HEAD~1}; R=$(git rev-parse --show-toplevel); T=/tmp/prevbuild; rm -rf $T; mkdir -p $T
git -C $R archive $c robosuite_amd/csrc include | tar -x -C $T
make -s -j8 -C $T/robosuite_amd/csrc
cp $T/robosuite_amd/librsim_hip.so $R/robosuite_amd/librsim_hip_prev.so
echo "robosuite_amd/librsim_hip_prev.so = $(git -C $R rev-parse --short $c)"
