#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Node:|Max Waves' | head -12; rocm-smi --showmemuse --showuse 2>/dev/null | grep -E 'GPU\[' | head -4; python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, p.total_memory>>30, 'GiB', torch.cuda.device_count(), 'devices')") > gpurun_out/r3i_box.txt 2>&1; cat gpurun_out/r3i_box.txt
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r3i_pytest.log 2>&1; grep -E "passed|failed|Error|portal warm start vs|^E  |tests/.*Error|worst deviations" gpurun_out/r3i_pytest.log | cut -c1-500 | tail -30
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for rep in 1 2; do
  RSIM_NO_MPR_PORTAL_WARMSTART=1 $B 2>gpurun_out/r3i_err.log | field "separating direction only"
  $B 2>>gpurun_out/r3i_err.log | field "+ portal (shallow polytopes)"
done > gpurun_out/r3i_ab.txt 2>&1
cat gpurun_out/r3i_ab.txt
