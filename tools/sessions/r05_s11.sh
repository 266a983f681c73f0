#!/bin/bash
# round-5 session 13: last robustness tweak of the line search; suite: parity (256 envs), A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
RSIM_PARITY_DUMP=$GRAFT_REPO_ROOT/$O/r05_m_dump.npz RSIM_PARITY_SAMPLE=256 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_m_parity_pickplace.txt 2>&1
grep -E "polish exits|oracle fed|passed|failed|^E  |^         [0-9]" $O/r05_m_parity_pickplace.txt | cut -c1-420 | head -12
for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_m_parity_pickplace.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-9;i<=n;i++) printf " %s", $i; print ""}'; done
bash tools/gpu_session.sh r05_m tests
