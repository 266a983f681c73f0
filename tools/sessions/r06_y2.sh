#!/bin/bash
# PickPlace: how early an env moves to the 256-row tier (RSIM_TIER_UP_CON / RSIM_TIER_UP_EFC: contacts / rows short of the native capacity at which an env is flagged
# for the wide pass of the NEXT step) against what the redo of an unflagged overflow costs.  Round-robin, two reps, same box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; TAG=${TAG:-r06_y2}
for rep in 1 2; do for s in "2 6" "4 14" "6 22" "10 38"; do
  set -- $s
  RSIM_TIER_UP_CON=$1 RSIM_TIER_UP_EFC=$2 timeout 300 python bench.py --config pickplace --steps 40 --warmup 3 --preroll 100 --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/${TAG}_up_$1_$2_$rep.json 2> $O/${TAG}.err
  python - $O/${TAG}_up_$1_$2_$rep.json "$s" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tier_up", sys.argv[2], "value %.0f ms %.2f" % (d["value"], d["ms_per_step"]), d["step_ms"], "tier_env_steps", d["tier_env_steps"], "mid_step", d["tier_changes_in_mid_step"], "diverged", d["config"].get("diverged_envs"), "overflow", d["config"].get("overflow_envs"))
PY
done; done
