#!/bin/bash
# round 3, session m: PickPlace solver-metric gap, distribution over 192 envs, previous build against the new one; GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3m_box_probe.txt 2>&1; rc=$?; cat gpurun_out/r3m_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
PREV=$PWD/robosuite_amd/librsim_hip_prev.so
(RSIM_LIB=$PREV timeout 600 python tools/pp_gap_stats.py 192 50; timeout 600 python tools/pp_gap_stats.py 192 50; RSIM_LIB=$PREV timeout 600 python tools/pp_gap_stats.py 192 30; timeout 600 python tools/pp_gap_stats.py 192 30) > gpurun_out/r3m_pp_gap_stats.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3m_pp_gap_stats.txt | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r3m_pytest.log 2>&1; grep -E "passed|failed|Error|^E  |tests/.*Error|worst qacc" gpurun_out/r3m_pytest.log | cut -c1-700 | tail -30
