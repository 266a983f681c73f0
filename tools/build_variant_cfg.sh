#!/bin/bash
# Build a variant of ONE kernel configuration next to the default library, sharing every other object of the default build (run `make` first):
#   [SRC=/path/to/other_step.hip] tools/build_variant_cfg.sh <name> <cfg 0..7> "<flags that REPLACE the configuration's CFGnFLAGS>" ["<extra APIFLAGS>"]   ->  robosuite_amd/librsim_hip_<name>.so
set -eu
name=$1; cfg=$2; fl=${3:-}; fa=${4:-}
cd "$(dirname "$0")/../robosuite_amd/csrc"
for v in CXXFLAGS TORCH_LIB HIPCC ARCH; do eval "$v=\"$(make -s print-$v)\""; done
D=/tmp/rsim_variant_$name; mkdir -p $D
SRC=${SRC:-rsim_step.hip}
$HIPCC $CXXFLAGS -I. -I../../include -DRSIM_CFG=$cfg $fl -x hip -c $SRC -o $D/cfg.o &
if [ -n "$fa" ]; then $HIPCC $CXXFLAGS $fa -x hip -c rsim_api.cpp -o $D/api.o & else cp rsim_api.o $D/api.o; fi
wait
objs=""
for c in 0 1 2 3 4 5 6 7; do
  o=rsim_step_cfg$c.o; [ $c = 0 ] && o=rsim_step.o
  [ $c = $cfg ] && o=$D/cfg.o
  objs="$objs $o"
done
$HIPCC --offload-arch=$ARCH -shared -fPIC -o ../librsim_hip_$name.so $objs $D/api.o rsim_mjcf.o -L$TORCH_LIB -Wl,-rpath,$TORCH_LIB
echo built ../librsim_hip_$name.so
