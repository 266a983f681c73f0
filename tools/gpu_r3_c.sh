#!/bin/bash
# round 3, third GPU session: GPU tests again (the record_stream crash fixed), solo envs with the launch order swapped
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3c_pytest.log 2>&1; tail -25 gpurun_out/r3c_pytest.log | cut -c1-400
B="timeout 300 python bench.py --no-cpu-baseline --no-open-loop"
field() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s value %9.0f  ms/step %.3f  stale %d diverged %d' % ('$1', d['value'], d['ms_per_step'], d['config']['reset_ring']['bank_stale'], d['config']['diverged_envs']))"; }
for rep in 1 2; do
  $B --solo 0 2>gpurun_out/r3c_err.log | field "warm start, solo 0"
  $B --solo 32 2>>gpurun_out/r3c_err.log | field "warm start, solo 32"
  $B --solo 64 2>>gpurun_out/r3c_err.log | field "warm start, solo 64"
  $B --solo 128 2>>gpurun_out/r3c_err.log | field "warm start, solo 128"
  $B --solo 256 2>>gpurun_out/r3c_err.log | field "warm start, solo 256"
  $B --solo 512 2>>gpurun_out/r3c_err.log | field "warm start, solo 512"
done > gpurun_out/r3c_ab.txt 2>&1
cat gpurun_out/r3c_ab.txt; tail -5 gpurun_out/r3c_err.log
RSIM_XSLOTS=ticks RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so timeout 400 python tools/tail_report.py 200 > gpurun_out/r3c_tail_subprof.txt 2>&1; cat gpurun_out/r3c_tail_subprof.txt
