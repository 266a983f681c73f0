#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_full_size_parity.py::test_pickplace_8192_with_dynamics_randomisation_reached_states > gpurun_out/r3f_pytest.log 2>&1; tail -5 gpurun_out/r3f_pytest.log | cut -c1-300
timeout 600 python tools/pp_dump.py 5 8192 1 > gpurun_out/r3f_pp_dump.txt 2>&1; tail -12 gpurun_out/r3f_pp_dump.txt | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-open-loop > gpurun_out/r3f_bench.json 2>gpurun_out/r3f_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3f_bench.json')); print('lockstep', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['reset_ring'])"
