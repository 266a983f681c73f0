#!/bin/bash
# round 3, session k: previous build against pair list + line-search exit, same box; whole GPU suite on the new build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for rep in 1 2; do
  RSIM_LIB=$PWD/robosuite_amd/librsim_hip_prev.so $B 2>gpurun_out/r3k_err.log | field "previous build"
  RSIM_NEWTON_LS=0 RSIM_BP_REACH=0 $B 2>>gpurun_out/r3k_err.log | field "new, list off, ls rule off"
  RSIM_NEWTON_LS=0 $B 2>>gpurun_out/r3k_err.log | field "new, pair list"
  $B 2>>gpurun_out/r3k_err.log | field "new, pair list + ls 1"
  RSIM_NEWTON_LS=10 $B 2>>gpurun_out/r3k_err.log | field "new, pair list + ls 10"
done > gpurun_out/r3k_ab.txt 2>&1
cat gpurun_out/r3k_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3k_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error" gpurun_out/r3k_pytest.log | cut -c1-500 | tail -30
for c in stack peg; do
  RSIM_LIB=$PWD/robosuite_amd/librsim_hip_prev.so $B --config $c 2>>gpurun_out/r3k_err.log | field "$c previous build"
  $B --config $c 2>>gpurun_out/r3k_err.log | field "$c new"
done > gpurun_out/r3k_ab_other.txt 2>&1
cat gpurun_out/r3k_ab_other.txt
