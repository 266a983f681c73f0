#!/bin/bash
# Round 5, session 18 (GPU side; the last 1.2 GPU-minutes): tools/sustained_load.py -- the identical Stack work before and behind 12 s of sustained Lift load.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout 58 python tools/sustained_load.py 12 > gpurun_out/r05_s18_sustained_load.txt 2> gpurun_out/r05_s18_err.txt; echo "rc $?" >> gpurun_out/r05_s18_sustained_load.txt
cat gpurun_out/r05_s18_sustained_load.txt; tail -3 gpurun_out/r05_s18_err.txt | cut -c1-300
