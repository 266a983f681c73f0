#!/bin/bash
# throughput vs batch size for two register-allocation variants of the fused kernel
cd $GRAFT_REPO_ROOT/robosuite_amd/csrc
for mw in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DRSIM_MINWAVES=$mw -c rsim_step.hip -o rsim_step.o && make -s >/dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  for B in 256 512 1024 2048 4096 8192; do
    echo -n "minwaves=$mw "; python tools/phase_profile.py $B 6 2>&1 | grep "ms/step"
  done
  cd robosuite_amd/csrc
done
