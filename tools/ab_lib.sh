#!/bin/bash
# A/B two builds of librsim_hip.so in ONE GPU session (box-to-box clock variance is ~15%): tools/ab_lib.sh base.so new.so [ramp steps]
cd $GRAFT_REPO_ROOT
N=${3:-250}
for rep in 1 2; do
  for lib in "$1" "$2"; do
    echo -n "$lib: "; RSIM_LIB=$GRAFT_REPO_ROOT/$lib python tools/ramp.py $N 2>&1 | grep -E "first   10|last 50" | sed 's/launches: mean//; s/env-steps.*//' | tr '\n' ' '; echo
  done
done
