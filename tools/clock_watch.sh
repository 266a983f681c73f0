#!/bin/bash
# sample clocks / power while the bench runs (DVFS diagnosis)
cd $GRAFT_REPO_ROOT
rocm-smi --showperflevel --showpower --showclocks 2>&1 | grep -v "^$" | head -30
python bench.py --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/cw_bench.log 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
grep -o '"value": [0-9.]*' gpurun_out/cw_bench.log
echo "--- perflevel high"
rocm-smi --setperflevel high 2>&1 | tail -2
python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'
rocm-smi --showperflevel 2>&1 | grep -i perf
echo "--- perfdeterminism"
rocm-smi --setperfdeterminism 2400 2>&1 | tail -2
python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*'
rocm-smi --resetperfdeterminism 2>&1 | tail -1; rocm-smi --setperflevel auto 2>&1 | tail -1
