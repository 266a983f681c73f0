#!/bin/bash
# round-5 session 6: polish with an exact fp64 line search: parity diagnostics (192 envs), A/B of the pass budget, suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_f probe || exit 3
for tag in default "RSIM_NEWTON_REFINE=12"; do
  envs=""; [ "$tag" != default ] && envs=$tag
  env $envs RSIM_PARITY_SAMPLE=192 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_f_parity_pickplace_${tag%%=*}.txt 2>&1
  echo "=== parity $tag"; grep -E "fp64 factor|oracle fed|passed|failed|^E  |^         [0-9]" $O/r05_f_parity_pickplace_${tag%%=*}.txt | cut -c1-330 | head -16
  for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_f_parity_pickplace_${tag%%=*}.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-13;i<=n;i++) printf " %s", $i; print ""}'; done
done
bash tools/ab_many.sh r05_f pickplace ${REPS:-3} librsim_hip.so librsim_hip.so@RSIM_NEWTON_REFINE=1 librsim_hip.so@RSIM_NEWTON_REFINE=0 librsim_hip.so@RSIM_NEWTON_REFINE=12
bash tools/gpu_session.sh r05_f tests
