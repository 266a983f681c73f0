#!/bin/bash
# round-5 session 8: polish = Newton with an exact fp64 line search across the kinks, gated by the first fp64 gradient: parity (192 envs, dump), A/B of gate / budget, suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
RSIM_PARITY_DUMP=$GRAFT_REPO_ROOT/$O/r05_h_dump.npz RSIM_PARITY_SAMPLE=192 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_h_parity_pickplace.txt 2>&1
grep -E "polish exits|oracle fed|passed|failed|^E  |^         [0-9]" $O/r05_h_parity_pickplace.txt | cut -c1-420 | head -14
for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_h_parity_pickplace.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-13;i<=n;i++) printf " %s", $i; print ""}'; done
bash tools/ab_many.sh r05_h pickplace ${REPS:-3} librsim_hip.so librsim_hip.so@RSIM_POLISH_GATE=1e-3 librsim_hip.so@RSIM_POLISH_GATE=1e-7 librsim_hip.so@RSIM_NEWTON_REFINE=1 librsim_hip.so@RSIM_NEWTON_REFINE=0
bash tools/gpu_session.sh r05_h tests
