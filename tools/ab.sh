#!/bin/bash
# A/B two builds of the fused kernel with tools/ramp.py: tools/ab.sh "<flagsA>" "<flagsB>"
cd $GRAFT_REPO_ROOT/robosuite_amd/csrc
for flags in "$1" "$2"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $flags -c rsim_step.hip -o rsim_step.o && make -s >/dev/null 2>&1
  echo "== flags: [$flags]"
  (cd $GRAFT_REPO_ROOT && python tools/ramp.py 300 2>&1 | grep -E "first   10|first  300|last 50")
done
