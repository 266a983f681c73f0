#!/bin/bash
# round 3, session x: kernel trace of the bench with the double-buffered leg (how the two halves' launches overlap)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 40 python tools/box_probe.py > gpurun_out/r3x_probe.txt 2>&1; rc=$?; if [ $rc -ne 0 ]; then echo "probe rc $rc"; exit 3; fi
rm -rf gpurun_out/prof_x
timeout 100 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_x -o r -- python bench.py --no-cpu-baseline --steps 60 --warmup 5 > gpurun_out/r3x_bench.json 2> gpurun_out/r3x_err.log
python tools/trace_double_buffer.py gpurun_out/prof_x 60 | tee gpurun_out/r3x_double_buffer_trace.txt
rm -rf gpurun_out/prof_x
