#!/bin/bash
# A/B several compile-flag variants of the fused kernel with tools/ramp.py in one GPU session: tools/ab.sh "<flagsA>" "<flagsB>" ...
cd $GRAFT_REPO_ROOT/robosuite_amd/csrc
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $flags -c rsim_step.hip -o rsim_step.o && make -s >/dev/null 2>&1
  echo -n "[$flags]: "
  (cd $GRAFT_REPO_ROOT && python tools/ramp.py 250 2>&1 | grep -E "first   10|last 50" | sed 's/launches: mean//; s/env-steps.*//' | tr '\n' ' '); echo
done
