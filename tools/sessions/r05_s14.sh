#!/bin/bash
# Round 5, session 14 (GPU side; 6 GPU-minutes left): what is left of a Stack control step after the occupancy change is the capacity tier's tail (k_step 5.76 ms of 6.7).
# Lockstep A/B on the fastest build of session 13 (s8t7b), round-robin, two reps: the list kernel held to 256 registers as well (s8t7c: starts in any freed wave slot),
# the advance-flagging thresholds (RSIM_TIER_UP_CON / _EFC: 0 / 0, the default 2 / 6, 4 / 12), and the wide pass ahead of the native one on the batch's stream again.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
rm -f gpurun_out/r05_s14_ab_many_stack.txt
bash tools/ab_many.sh r05_s14 stack 2 librsim_hip_s8t7b.so librsim_hip_s8t7c.so librsim_hip_s8t7b.so@RSIM_TIER_UP_CON=0,RSIM_TIER_UP_EFC=0 librsim_hip_s8t7b.so@RSIM_TIER_UP_CON=4,RSIM_TIER_UP_EFC=12 librsim_hip_s8t7c.so@RSIM_TIER_UP_CON=4,RSIM_TIER_UP_EFC=12 librsim_hip_s8t7b.so@RSIM_TIER_MODE=1
echo "[s14] end at $SECONDS s"
