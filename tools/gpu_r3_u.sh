#!/bin/bash
# round 3, session u: VALU instruction counts (one SQ pass each) of the other three BASELINE configurations on this build, and a short lockstep bench line
# of each that carries roofline.issue.  Tight time limits: some boxes of the pool fault on every process.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python tools/box_probe.py > gpurun_out/r3u_box_probe.txt 2>&1; rc=$?; head -4 gpurun_out/r3u_box_probe.txt | cut -c1-120; if [ $rc -ne 0 ]; then echo "probe rc $rc: stopping"; exit 3; fi
export PMC_TIMEOUT=70
for c in stack peg pickplace; do
  export RSIM_CONFIG=$c
  case $c in pickplace) export RSIM_BENCH_EXTRA="--preroll 40"; extra="--steps 10 --warmup 2 --preroll 40";; *) export RSIM_BENCH_EXTRA="--preroll 200"; extra="--steps 50 --warmup 5 --preroll 200";; esac
  KEEP=1 bash tools/pmc_pass.sh r3u_$c sq1 > /dev/null
  python tools/pmc_valu.py gpurun_out/r3u_$c.sq1 4 > /dev/null; rm -rf gpurun_out/r3u_$c.sq1
  cp profiles/valu_count_$c.json gpurun_out/
  timeout 60 python bench.py --config $c $extra --no-open-loop --no-cpu-baseline > gpurun_out/r03_z_issue_$c.json 2> gpurun_out/r03_z_issue_$c.err
  python -c "import json; d=json.loads(open('gpurun_out/r03_z_issue_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['value']), d['roofline']['issue'] and {k: d['roofline']['issue'][k] for k in ('valu_instr_per_env_substep','achieved','frac')})"
done
