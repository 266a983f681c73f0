#!/bin/bash
# One GPU-box session for the round-3 evidence: bench of every BASELINE configuration (lockstep headline + open-loop figure + cpu baseline),
# rocprofv3 kernel stats of the same commands, PMC passes of the Lift workload (SQ counters, HBM bytes) keyed to the library build, tail report.
# Usage: gpurun -- 'bash tools/gpu_round3.sh r03_z'
set -x
tag=${1:-r03}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/box_probe.py > gpurun_out/${tag}_box_probe.txt 2>&1; rc=$?; cat gpurun_out/${tag}_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.txt
# PMC first: profiles/valu_count.json and hbm_traffic.json must carry the sha of THIS build before the bench reads them
KEEP=1 bash tools/pmc_pass.sh $tag sq1 hbm1 hbm2
python tools/pmc_valu.py gpurun_out/$tag.sq1 4
python tools/pmc_traffic.py gpurun_out/$tag.hbm1 gpurun_out/$tag.hbm2 4
rm -rf gpurun_out/$tag.sq1 gpurun_out/$tag.hbm1 gpurun_out/$tag.hbm2
cp profiles/valu_count.json profiles/hbm_traffic.json gpurun_out/
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 3000 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o r -- python bench.py --no-cpu-baseline --no-open-loop > gpurun_out/${tag}_prof.log 2>&1
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv; head -6 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200; rm -rf gpurun_out/prof_$tag
for c in stack peg pickplace; do
  case $c in pickplace) extra="--steps 30 --warmup 5 --preroll 100";; *) extra="--steps 100 --warmup 10";; esac
  timeout 900 python bench.py --config $c $extra > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err; tail -c 1500 gpurun_out/${tag}_bench_$c.json
  rm -rf gpurun_out/prof_${tag}_$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$c -o r -- python bench.py --config $c --no-cpu-baseline --no-open-loop --steps 20 --warmup 5 --preroll 50 > gpurun_out/${tag}_prof_$c.log 2>&1
  cp $(find gpurun_out/prof_${tag}_$c -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats_$c.csv; head -3 gpurun_out/${tag}_kernel_stats_$c.csv | cut -c1-200; rm -rf gpurun_out/prof_${tag}_$c
done
[ -f robosuite_amd/librsim_hip_prof.so ] && RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so timeout 400 python tools/tail_report.py 200 > gpurun_out/${tag}_tail_report.txt 2>&1; head -12 gpurun_out/${tag}_tail_report.txt | cut -c1-400
timeout 300 python tools/phase_profile_env.py 200 0 1 2 > gpurun_out/${tag}_phase_env.txt 2>&1; tail -4 gpurun_out/${tag}_phase_env.txt | cut -c1-600
