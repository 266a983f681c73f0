#!/bin/bash
# round 6, session w: gain of the contact-onset hint (RSIM_NEAR_GAIN) at the adopted threshold of 2 mm, round robin x 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2; do for g in 0.25 0.5 1.0 2.0 4.0; do
  for cfgargs in "peg --steps 100 --warmup 10" "lift --steps 100 --warmup 10" "stack --steps 60 --warmup 10"; do
    set -- $cfgargs; cfg=$1; shift
    RSIM_NEAR_GAIN=$g timeout 300 python bench.py --config $cfg "$@" --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/r06_w_${cfg}_${g}_$rep.json 2> $O/r06_w.err
    python - $O/r06_w_${cfg}_${g}_$rep.json $g $cfg <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[3]:6s} near_gain {sys.argv[2]:>5}: {d['value']/1e3:8.1f} K  {d['ms_per_step']:.3f} ms/step  p50 {d['step_ms']['p50']:.2f} p90 {d['step_ms']['p90']:.2f} max {d['step_ms']['max']:.2f}  reward_sum {d['config']['reward_sum']:.3f}")
PY
  done
done; done
