#!/bin/bash
# round-5 session 10: plain Newton step first, fp64 line search on demand: parity (192 envs) for both modes, A/B, suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
for tag in default "RSIM_POLISH_GATE=-1"; do
  envs=""; [ "$tag" != default ] && envs=$tag
  env $envs RSIM_PARITY_DUMP=$GRAFT_REPO_ROOT/$O/r05_j_dump_${tag%%=*}.npz RSIM_PARITY_SAMPLE=192 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_j_parity_pickplace_${tag%%=*}.txt 2>&1
  echo "== parity $tag"; grep -E "polish exits|oracle fed|passed|failed|^E  " $O/r05_j_parity_pickplace_${tag%%=*}.txt | cut -c1-420 | head -6
  for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_j_parity_pickplace_${tag%%=*}.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-9;i<=n;i++) printf " %s", $i; print ""}'; done
done
bash tools/ab_many.sh r05_j pickplace ${REPS:-3} librsim_hip_r4.so librsim_hip.so librsim_hip.so@RSIM_POLISH_GATE=-1 librsim_hip.so@RSIM_NEWTON_REFINE=0
bash tools/gpu_session.sh r05_j tests
