#!/bin/bash
# Build a variant of the 64 x 48 configurations (cfg 3 / 5) next to the default library, sharing every other object of the default build (run `make` first):
#   [SRC=/path/to/other_step.hip] tools/build_variant.sh <name> "<extra CFG3FLAGS>" ["<extra APIFLAGS>"]   ->  robosuite_amd/librsim_hip_<name>.so
set -eu
name=$1; f3=${2:-}; fa=${3:-}
cd "$(dirname "$0")/../robosuite_amd/csrc"
for v in CXXFLAGS CFG3FLAGS TORCH_LIB HIPCC ARCH; do eval "$v=\"$(make -s print-$v)\""; done
SRC=${SRC:-rsim_step.hip}
D=/tmp/rsim_variant_$name; mkdir -p $D
$HIPCC $CXXFLAGS -I. -DRSIM_CFG=3 $CFG3FLAGS $f3 -x hip -c $SRC -o $D/cfg3.o &
$HIPCC $CXXFLAGS -I. -DRSIM_CFG=5 $CFG3FLAGS $f3 -x hip -c $SRC -o $D/cfg5.o &
if [ -n "$fa" ]; then $HIPCC $CXXFLAGS $fa -x hip -c rsim_api.cpp -o $D/api.o & else cp rsim_api.o $D/api.o; fi
wait
$HIPCC --offload-arch=$ARCH -shared -fPIC -o ../librsim_hip_$name.so rsim_step.o rsim_step_cfg1.o rsim_step_cfg2.o $D/cfg3.o rsim_step_cfg4.o $D/cfg5.o rsim_step_cfg6.o rsim_step_cfg7.o $D/api.o rsim_mjcf.o -L$TORCH_LIB -Wl,-rpath,$TORCH_LIB
echo built ../librsim_hip_$name.so
