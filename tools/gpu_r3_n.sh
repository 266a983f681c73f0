#!/bin/bash
# round 3, session n: force / torque sensors on the device; hot kernel without the debug block against the previous build; GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3n_box_probe.txt 2>&1; rc=$?; cat gpurun_out/r3n_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "sensor or shim_on_the_hip" > gpurun_out/r3n_pytest_sensors.log 2>&1; grep -E "passed|failed|Error|^E  |worst relative sensor" gpurun_out/r3n_pytest_sensors.log | cut -c1-600 | tail -20
PREV=$PWD/robosuite_amd/librsim_hip_prev.so
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d reward %.3f' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs'], d['config']['reward_sum']))"; }
for rep in 1 2; do
  RSIM_LIB=$PREV $B 2>gpurun_out/r3n_err.log | field "previous build"
  $B 2>>gpurun_out/r3n_err.log | field "new"
done > gpurun_out/r3n_ab.txt 2>&1
cat gpurun_out/r3n_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3n_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error" gpurun_out/r3n_pytest.log | cut -c1-700 | tail -30
