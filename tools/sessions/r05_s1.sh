#!/bin/bash
# round-5 session 1: default build (advisor fixes) through the GPU suite; the four prepared 64 x 48 variants + their union + the fp64-evaluated polish, A/B at 5 round-robin reps
# on PickPlace @8192 + DR, each through the PickPlace / tendon subset of the suite; PickPlace full-size parity printed for default and polish
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_a probe tests || exit 3
V="mglobal jrows jprefetch jg256 all4 m3 polish"
libs="librsim_hip.so"; for v in $V; do libs="$libs librsim_hip_$v.so"; done
bash tools/ab_many.sh r05_a pickplace ${REPS:-5} $libs
for v in $V; do
  echo "=== suite subset on $v"
  RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_$v.so timeout 600 python -m pytest tests -m gpu -q -x -k "pickplace or PickPlace or pick_place or tendon or robotiq or Robotiq" > $O/r05_a_pytest_$v.txt 2>&1; tail -4 $O/r05_a_pytest_$v.txt | cut -c1-300
done
for v in default polish; do
  lib=librsim_hip.so; [ $v != default ] && lib=librsim_hip_$v.so
  RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace" > $O/r05_a_parity_pickplace_$v.txt 2>&1; grep -v "^$" $O/r05_a_parity_pickplace_$v.txt | tail -25 | cut -c1-400
done
