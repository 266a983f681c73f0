#!/bin/bash
# Round 5, session 12 (CPU side): variants of the 32 x 32 configuration (Stack) with the constraint Jacobian and the mass matrix in the per-env global buffer
# (the move that took PickPlace from two to four envs per CU), held to 256 registers so that two wavefronts share a SIMD:
#   s6: 23.4 KB of LDS = six envs per CU          s7: the same without the LDS hull pool (22.6 KB = seven envs per CU)
# Each is linked against the other objects of the default build (run `make` first):  ->  robosuite_amd/librsim_hip_{s6,s7}.so
set -eu
cd "$(dirname "$0")/../../robosuite_amd/csrc"
for v in CXXFLAGS TORCH_LIB HIPCC ARCH; do eval "$v=\"$(make -s print-$v)\""; done
S6="-DRSIM_JGLOBAL -DRSIM_MGLOBAL -DRSIM_MINWAVES=2 -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills"
D=/tmp/rsim_variant_s; mkdir -p $D
$HIPCC $CXXFLAGS -I. -DRSIM_CFG=1 $S6 -x hip -c rsim_step.hip -o $D/s6.o &
$HIPCC $CXXFLAGS -I. -DRSIM_CFG=1 $S6 -DRSIM_NOHULLPOOL=1 -x hip -c rsim_step.hip -o $D/s7.o &
wait
for n in s6 s7; do
  $HIPCC --offload-arch=$ARCH -shared -fPIC -o ../librsim_hip_$n.so rsim_step.o $D/$n.o rsim_step_cfg2.o rsim_step_cfg3.o rsim_step_cfg4.o rsim_step_cfg5.o rsim_step_cfg6.o rsim_step_cfg7.o rsim_api.o rsim_mjcf.o -L$TORCH_LIB -Wl,-rpath,$TORCH_LIB
  echo built ../librsim_hip_$n.so
done
