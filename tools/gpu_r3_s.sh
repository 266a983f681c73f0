#!/bin/bash
# round 3, session s: the fp32 Newton rules on the wide configurations -- GPU suite under each setting, Stack / PickPlace / peg throughput
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3s_box_probe.txt 2>&1; rc=$?; tail -1 gpurun_out/r3s_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
for w in 1 2; do
  echo "== RSIM_NEWTON_WIDE=$w"
  RSIM_NEWTON_WIDE=$w timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3s_pytest_wide$w.log 2>&1; grep -E "passed|failed|^FAILED|^E  " gpurun_out/r3s_pytest_wide$w.log | cut -c1-300 | tail -12
done
B="timeout 600 python bench.py --no-cpu-baseline --no-open-loop --steps 60 --warmup 10"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for c in stack peg; do for w in 0 1 2; do RSIM_NEWTON_WIDE=$w $B --config $c 2>>gpurun_out/r3s_err.log | field "$c wide $w"; done; done > gpurun_out/r3s_ab.txt 2>&1
cat gpurun_out/r3s_ab.txt
