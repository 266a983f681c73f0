#!/bin/bash
# Round 5, session 13 (CPU side): variants on top of the seven-envs-per-CU Stack build (the default since session 12), each linked against the other objects of the
# default build (run `make` first):
#   s8      the 32 x 32 configuration with the contact block (frames, material parameters) in the per-env global buffer as well (-DRSIM_CGLOBAL: 19.8 KB of LDS = EIGHT envs
#           per CU, two wavefronts on every SIMD)
#   t7a     the capacity tier above it (32 x 32, 128 rows: 50.1 KB, 383 registers) with J and M in the global buffer and no hull pool (28.3 KB: fits once two native
#           workgroups of a CU have ended instead of three)
#   t7b     t7a with the register allocator held to 256 (k_step_list 315)
#   ->  robosuite_amd/librsim_hip_{s8,t7a,s8t7a,s8t7b}.so
set -eu
cd "$(dirname "$0")/../../robosuite_amd/csrc"
for v in CXXFLAGS CFG1FLAGS TORCH_LIB HIPCC ARCH; do eval "$v=\"$(make -s print-$v)\""; done
D=/tmp/rsim_variant_s; mkdir -p $D
T7="-DRSIM_JGLOBAL -DRSIM_MGLOBAL -DRSIM_NOHULLPOOL=1"
$HIPCC $CXXFLAGS -DRSIM_CFG=1 $CFG1FLAGS -DRSIM_CGLOBAL -c rsim_step.hip -o $D/s8.o &
$HIPCC $CXXFLAGS -DRSIM_CFG=7 $T7 -c rsim_step.hip -o $D/t7a.o &
$HIPCC $CXXFLAGS -DRSIM_CFG=7 $T7 -DRSIM_MINWAVES=2 -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills -c rsim_step.hip -o $D/t7b.o &
wait
link() { $HIPCC --offload-arch=$ARCH -shared -fPIC -o ../librsim_hip_$1.so rsim_step.o $2 rsim_step_cfg2.o rsim_step_cfg3.o rsim_step_cfg4.o rsim_step_cfg5.o rsim_step_cfg6.o $3 rsim_api.o rsim_mjcf.o -L$TORCH_LIB -Wl,-rpath,$TORCH_LIB; echo built ../librsim_hip_$1.so; }
link s8 $D/s8.o rsim_step_cfg7.o
link t7a rsim_step_cfg1.o $D/t7a.o
link s8t7a $D/s8.o $D/t7a.o
link s8t7b $D/s8.o $D/t7b.o
