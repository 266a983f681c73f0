#!/bin/bash
# round 3, session o: PickPlace single-object mode 1 on the device; GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3o_box_probe.txt 2>&1; rc=$?; cat gpurun_out/r3o_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "mode_1 or single_object or pickplace_obs" > gpurun_out/r3o_pytest_mode1.log 2>&1; grep -E "passed|failed|Error|^E  " gpurun_out/r3o_pytest_mode1.log | cut -c1-600 | tail -20
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3o_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error" gpurun_out/r3o_pytest.log | cut -c1-700 | tail -30
