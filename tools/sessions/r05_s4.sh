#!/bin/bash
# round-5 session 4: polish behind EVERY solve (not only behind a factorisation), fp64 factor only where the gradient is far from the tolerance; suite, A/B, parity, DR rate test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_d probe tests || exit 3
bash tools/ab_many.sh r05_d pickplace ${REPS:-4} librsim_hip_r4.so librsim_hip.so librsim_hip.so@RSIM_NEWTON_REFINE=1 librsim_hip.so@RSIM_NO_H64=1 librsim_hip.so@RSIM_POLISH_TOL=100 librsim_hip.so@RSIM_NEWTON_REFINE=0
for tag in default "RSIM_NO_H64=1" "RSIM_POLISH_TOL=100"; do
  envs=""; [ "$tag" != default ] && envs=$tag
  env $envs RSIM_PARITY_SAMPLE=128 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_d_parity_pickplace_${tag%%=*}.txt 2>&1
  echo "=== parity, $tag"; grep -E "fp64 factor|oracle fed|passed|failed|^E  " $O/r05_d_parity_pickplace_${tag%%=*}.txt | cut -c1-400
  for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_d_parity_pickplace_${tag%%=*}.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-13;i<=n;i++) printf " %s", $i; print ""}'; done
done
echo "=== DR bad-state test"
timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "bad_state_rate" > $O/r05_d_dr_bad_state.txt 2>&1; grep -E "kernel:|oracle, same|passed|failed|^E  " $O/r05_d_dr_bad_state.txt | cut -c1-600
