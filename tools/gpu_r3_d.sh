#!/bin/bash
# round 3, GPU session 4: the call-by-call controller tests, solo envs diagnosed from the wave log, fp32 Newton stopping rules swept
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "call_by_call or worst_conditioned" > gpurun_out/r3d_pytest_ctl.log 2>&1; grep -E "calls, worst|cond in|passed|failed|Error|assert" gpurun_out/r3d_pytest_ctl.log | cut -c1-300
timeout 900 python tools/newton_sweep.py 200 "0,0,0" "1e-6,1e-6,0" "1e-5,1e-5,0" "1e-4,1e-4,0" "0,0,3e-7" "0,0,1e-6" "0,0,3e-6" "0,0,1e-5" "1e-5,1e-5,1e-6" > gpurun_out/r3d_newton_sweep.txt 2>&1; cat gpurun_out/r3d_newton_sweep.txt | cut -c1-700
