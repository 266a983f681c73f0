#!/bin/bash
# One GPU-box session for round 2: bench (steady-state phase), rocprofv3 kernel stats of the same command, PMC passes (SQ counters, HBM bytes), phase profile.
set -x
tag=${1:-r02}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 2500 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o r -- python bench.py --no-cpu-baseline --no-lockstep > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv; cat gpurun_out/${tag}_kernel_stats.csv; rm -rf gpurun_out/prof_$tag
KEEP=1 bash tools/pmc_pass.sh $tag sq1 hbm1 hbm2
python tools/pmc_valu.py gpurun_out/$tag.sq1 4
python tools/pmc_traffic.py gpurun_out/$tag.hbm1 gpurun_out/$tag.hbm2 4
rm -rf gpurun_out/$tag.sq1 gpurun_out/$tag.hbm1 gpurun_out/$tag.hbm2
cp profiles/valu_count.json profiles/hbm_traffic.json gpurun_out/
timeout 300 python tools/phase_profile.py 4096 6 3 > gpurun_out/${tag}_phase.txt 2>&1; head -20 gpurun_out/${tag}_phase.txt
