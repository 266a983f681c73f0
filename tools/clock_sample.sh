#!/bin/bash
# engine clock / power while the bench's timed region runs
cd $GRAFT_REPO_ROOT
python bench.py --steps 4000 --warmup 20 --no-cpu-baseline > gpurun_out/cs_bench.log 2>&1 &
BP=$!
sleep 14
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | sed 's/GPU\[0\]//' | tr '\n' ' '; echo; sleep 1; done
wait $BP
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/cs_bench.log
