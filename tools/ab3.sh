#!/bin/bash
# A/B any number of builds of librsim_hip.so in ONE GPU session: tools/ab3.sh N lib1.so lib2.so ...   (paths relative to the repo root)
cd $GRAFT_REPO_ROOT
N=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "; RSIM_LIB=$GRAFT_REPO_ROOT/$lib python tools/ramp.py $N 2>&1 | grep -E "first   10|last 50|Error|error" | sed 's/launches: mean//; s/env-steps.*//' | tr '\n' ' '; echo
  done
done
