#!/bin/bash
# round-5 session 9: does the 256-row tier's line-search code (spills) cost the PickPlace step?  default vs RSIM_LS_MAXSLOT=2, each with and without the polish; parity again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/ab_many.sh r05_i pickplace ${REPS:-3} librsim_hip.so librsim_hip_ls2.so librsim_hip.so@RSIM_NEWTON_REFINE=0 librsim_hip_ls2.so@RSIM_NEWTON_REFINE=0 librsim_hip_ls2.so@RSIM_NEWTON_REFINE=1
for lib in librsim_hip.so librsim_hip_ls2.so; do
RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib RSIM_PARITY_SAMPLE=192 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_i_parity_pickplace_${lib%.so}.txt 2>&1
echo "== parity $lib"; grep -E "polish exits|oracle fed|passed|failed|^E  " $O/r05_i_parity_pickplace_${lib%.so}.txt | cut -c1-420 | head -6
for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_i_parity_pickplace_${lib%.so}.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-9;i<=n;i++) printf " %s", $i; print ""}'; done
done
