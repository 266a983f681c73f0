#!/bin/bash
# round-5 session 5: where does the PickPlace per-env tail come from?  polish diagnostics (RSIM_POLISH) and the per-state fp32-input floor of the fp64 solve, worst envs listed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_e probe || exit 3
RSIM_PARITY_SAMPLE=192 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_e_parity_pickplace.txt 2>&1
grep -E "fp64 factor|oracle fed|passed|failed|^E  |worst envs|^         [0-9]" $O/r05_e_parity_pickplace.txt | cut -c1-330
bash tools/gpu_session.sh r05_e tests
