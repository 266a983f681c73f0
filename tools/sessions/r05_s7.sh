#!/bin/bash
# round-5 session 7: dump the worst PickPlace envs of the full-size parity sample for offline analysis on the CPU (state, per-env model, kernel's accelerations / forces / contacts)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
RSIM_PARITY_DUMP=$GRAFT_REPO_ROOT/$O/r05_g_dump.npz RSIM_PARITY_SAMPLE=192 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_g_parity_pickplace.txt 2>&1
grep -E "fp64 factor|oracle fed|passed|failed|^E  |^         [0-9]" $O/r05_g_parity_pickplace.txt | cut -c1-330 | head -16
bash tools/ab_many.sh r05_g pickplace 2 librsim_hip.so librsim_hip.so@RSIM_NEWTON_REFINE=1 librsim_hip.so@RSIM_NEWTON_REFINE=0
ls -la $O/r05_g_dump.npz
