#!/bin/bash
# round 3, session l: pair list in global memory + line-search exit against the previous build; PickPlace cost-gap check under both builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PREV=$PWD/robosuite_amd/librsim_hip_prev.so
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for rep in 1 2; do
  RSIM_LIB=$PREV $B 2>gpurun_out/r3l_err.log | field "previous build"
  RSIM_BP_REACH=0 $B 2>>gpurun_out/r3l_err.log | field "new, list off"
  RSIM_NEWTON_LS=0 $B 2>>gpurun_out/r3l_err.log | field "new, ls rule off"
  $B 2>>gpurun_out/r3l_err.log | field "new"
done > gpurun_out/r3l_ab.txt 2>&1
cat gpurun_out/r3l_ab.txt
timeout 600 python tools/tail_report.py 200 > gpurun_out/r3l_tail_report.txt 2>&1; grep -E "^step|^schedule|^last envs" gpurun_out/r3l_tail_report.txt | cut -c1-900
pp() { timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace" 2>&1 | grep -E "objective|passed|failed|^E  " | cut -c1-600; }
(echo "previous build"; RSIM_LIB=$PREV pp; echo "new build"; pp) > gpurun_out/r3l_pp_gap.txt 2>&1
cat gpurun_out/r3l_pp_gap.txt
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3l_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error|worst qacc" gpurun_out/r3l_pytest.log | cut -c1-700 | tail -30
