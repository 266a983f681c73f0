#!/bin/bash
# round 3, session j: broadphase active pair list -- bitwise test, the whole GPU suite, A/B over the reach
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Node:|Max Waves' | head -12; python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, p.total_memory>>30, 'GiB', torch.cuda.device_count(), 'devices')") > gpurun_out/r3j_box.txt 2>&1; cat gpurun_out/r3j_box.txt
timeout 300 python -m pytest tests/test_hip_edge_cases.py -m gpu -q -s -k "pair_list or separating" > gpurun_out/r3j_pytest_list.log 2>&1; tail -5 gpurun_out/r3j_pytest_list.log | cut -c1-400
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for rep in 1 2; do
  for r in 0 0.02 0.04 0.08; do
    RSIM_BP_REACH=$r $B 2>gpurun_out/r3j_err.log | field "lift reach $r"
  done
done > gpurun_out/r3j_ab.txt 2>&1
cat gpurun_out/r3j_ab.txt
for c in stack peg pickplace; do for r in 0 0.04; do
  RSIM_BP_REACH=$r $B --config $c 2>>gpurun_out/r3j_err.log | field "$c reach $r"
done; done > gpurun_out/r3j_ab_other.txt 2>&1
cat gpurun_out/r3j_ab_other.txt
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3j_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error" gpurun_out/r3j_pytest.log | cut -c1-500 | tail -30
