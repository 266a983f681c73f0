#!/bin/bash
# round 3, second GPU session: the new GPU tests, A/B of the narrow-phase warm start and the solo envs on the lockstep bench, tail report
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_pytest.log 2>&1; tail -15 gpurun_out/r3b_pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs']))"; }
for rep in 1 2; do
  RSIM_NO_MPR_WARMSTART=1 $B --solo 0 2>gpurun_out/r3b_err.log | field "cold MPR, solo 0"
  $B --solo 0 2>>gpurun_out/r3b_err.log | field "warm start, solo 0"
  $B --solo 64 2>>gpurun_out/r3b_err.log | field "warm start, solo 64"
  $B --solo 128 2>>gpurun_out/r3b_err.log | field "warm start, solo 128"
  $B --solo 256 2>>gpurun_out/r3b_err.log | field "warm start, solo 256"
done > gpurun_out/r3b_ab.txt 2>&1
cat gpurun_out/r3b_ab.txt; tail -5 gpurun_out/r3b_err.log
RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so timeout 400 python tools/tail_report.py 200 > gpurun_out/r3b_tail_report.txt 2>&1; cat gpurun_out/r3b_tail_report.txt
timeout 400 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; tail -c 3000 gpurun_out/r3b_bench.json; tail -3 gpurun_out/r3b_bench.err
