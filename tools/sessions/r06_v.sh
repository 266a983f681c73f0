#!/bin/bash
# round 6, session v: contact-onset hint in the dispatch-order key (RSIM_NEAR_THRESH, metres; 0 = off): sweep on the three tail-bound configurations, round robin x 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2; do for th in ${THS:-0 0.002 0.005 0.01 0.02 0.05}; do
  for cfgargs in "peg --steps 100 --warmup 10" "lift --steps 100 --warmup 10" "stack --steps 60 --warmup 10"; do
    set -- $cfgargs; cfg=$1; shift
    RSIM_NEAR_THRESH=$th timeout 300 python bench.py --config $cfg "$@" --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/r06_v${SFX:-}_${cfg}_${th}_$rep.json 2> $O/r06_v.err
    python - $O/r06_v${SFX:-}_${cfg}_${th}_$rep.json $th $cfg <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[3]:6s} near_thresh {sys.argv[2]:>6}: {d['value']/1e3:8.1f} K  {d['ms_per_step']:.3f} ms/step  p50 {d['step_ms']['p50']:.2f} p90 {d['step_ms']['p90']:.2f} max {d['step_ms']['max']:.2f}  reward_sum {d['config']['reward_sum']:.3f}")
PY
  done
done; done
