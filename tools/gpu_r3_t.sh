#!/bin/bash
# round 3, session t: PMC passes (VALU count, HBM bytes) of the other three BASELINE configurations, keyed to the build; then their bench lines again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3t_box_probe.txt 2>&1; rc=$?; tail -1 gpurun_out/r3t_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
for c in stack peg pickplace; do
  export RSIM_CONFIG=$c
  case $c in pickplace) export RSIM_BENCH_EXTRA="--preroll 60";; *) export RSIM_BENCH_EXTRA="";; esac
  KEEP=1 bash tools/pmc_pass.sh r3t_$c sq1 hbm1 hbm2
  python tools/pmc_valu.py gpurun_out/r3t_$c.sq1 4
  python tools/pmc_traffic.py gpurun_out/r3t_$c.hbm1 gpurun_out/r3t_$c.hbm2 4
  rm -rf gpurun_out/r3t_$c.sq1 gpurun_out/r3t_$c.hbm1 gpurun_out/r3t_$c.hbm2
  cp profiles/valu_count_$c.json profiles/hbm_traffic_$c.json gpurun_out/
done
unset RSIM_CONFIG RSIM_BENCH_EXTRA
for c in stack peg pickplace; do
  case $c in pickplace) extra="--steps 30 --warmup 5 --preroll 100";; *) extra="--steps 100 --warmup 10";; esac
  timeout 900 python bench.py --config $c $extra > gpurun_out/r03_z_bench_$c.json 2> gpurun_out/r03_z_bench_$c.err
  python -c "import json; d=json.loads(open('gpurun_out/r03_z_bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['value']), d['roofline']['issue'] and {k: d['roofline']['issue'][k] for k in ('valu_instr_per_env_substep','achieved','frac')}, d['roofline']['traffic'], d['roofline']['algorithmic_bytes_per_launch'])"
done
