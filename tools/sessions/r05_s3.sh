#!/bin/bash
# round-5 session 3: fp64-factor fallback of the polish + JG256 as shipped; friction known answer r4 build vs now; suite; A/B of the polish settings; parity at 128 envs; DR bad-state test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_c probe || exit 3
for lib in librsim_hip_r4.so librsim_hip.so; do echo "== friction probe $lib"; RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib timeout 300 python tools/friction_probe.py 2>&1 | tail -2; done | tee $O/r05_c_friction_probe.txt
bash tools/gpu_session.sh r05_c tests
bash tools/ab_many.sh r05_c pickplace ${REPS:-5} librsim_hip_r4.so librsim_hip.so librsim_hip.so@RSIM_NEWTON_REFINE=1 librsim_hip.so@RSIM_NO_H64=1 librsim_hip.so@RSIM_POLISH_TOL=100
for tag in default "RSIM_NO_H64=1" "RSIM_NEWTON_REFINE=1"; do
  envs=""; [ "$tag" != default ] && envs=$tag
  env $envs RSIM_PARITY_SAMPLE=128 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_c_parity_pickplace_${tag%%=*}.txt 2>&1
  echo "=== parity, $tag"; grep -E "fp64 factor|oracle fed|passed|failed|^E  " $O/r05_c_parity_pickplace_${tag%%=*}.txt | cut -c1-400
  for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_c_parity_pickplace_${tag%%=*}.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-13;i<=n;i++) printf " %s", $i; print ""}'; done
done
echo "=== DR bad-state test"
timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "bad_state_rate" > $O/r05_c_dr_bad_state.txt 2>&1; grep -E "kernel:|oracle on|passed|failed|^E  " $O/r05_c_dr_bad_state.txt | cut -c1-600
