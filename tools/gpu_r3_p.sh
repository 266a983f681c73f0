#!/bin/bash
# round 3, session p: implicit-damping Euler restricted to the damped trees (Stack, PickPlace); GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3p_box_probe.txt 2>&1; rc=$?; cat gpurun_out/r3p_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
B="timeout 600 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for c in stack pickplace peg; do for rep in 1 2; do
  RSIM_EULER_FULL=1 $B --config $c 2>>gpurun_out/r3p_err.log | field "$c all dofs"
  $B --config $c 2>>gpurun_out/r3p_err.log | field "$c damped trees"
done; done > gpurun_out/r3p_ab.txt 2>&1
cat gpurun_out/r3p_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3p_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error" gpurun_out/r3p_pytest.log | cut -c1-700 | tail -30
