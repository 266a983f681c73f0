#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r3h_pytest.log 2>&1; grep -E "passed|failed|Error|portal warm start vs|solver objective|^E  |tests/.*Error" gpurun_out/r3h_pytest.log | cut -c1-400 | tail -30
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for rep in 1 2; do
  RSIM_NO_MPR_PORTAL_WARMSTART=1 $B 2>gpurun_out/r3h_err.log | field "separating direction only"
  $B 2>>gpurun_out/r3h_err.log | field "+ portal warm start (polytopes)"
done > gpurun_out/r3h_ab.txt 2>&1
cat gpurun_out/r3h_ab.txt
RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so timeout 400 python tools/tail_report.py 200 > gpurun_out/r3h_tail_report.txt 2>&1; head -16 gpurun_out/r3h_tail_report.txt | cut -c1-600
