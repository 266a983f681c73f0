#!/bin/bash
# round 3, first GPU session: baseline tests, the measured VALU issue ceiling, and what sets the length of a lockstep launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; tail -3 gpurun_out/r3a_pytest.log
timeout 120 tools/ubench/valu_peak > gpurun_out/r3a_valu_peak.txt 2>&1; cat gpurun_out/r3a_valu_peak.txt
RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so timeout 400 python tools/tail_report.py 200 > gpurun_out/r3a_tail_report.txt 2>&1; cat gpurun_out/r3a_tail_report.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; cat gpurun_out/r3a_bench.json | cut -c1-900
