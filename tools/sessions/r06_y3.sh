#!/bin/bash
# "lone" placement A/B: the first K dispatch slots (the slowest envs of the previous step) as a padded launch of their own (RSIM_LONE_K / RSIM_LONE_LDS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; TAG=${TAG:-r06_y3}; CFG=${CFG:-lift}
[ -n "$SKIP_TESTS" ] || timeout 300 python -m pytest tests/test_hip_edge_cases.py -m gpu -q -x 2>&1 | tail -3
[ -n "$SKIP_TESTS" ] || RSIM_LONE_K=4 timeout 300 python -m pytest tests/test_hip_edge_cases.py tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do for s in ${SETTINGS:-"0 0" "1 65536" "2 65536" "4 65536" "8 65536" "16 65536" "2 30000" "2 122880" "4 122880"}; do
  set -- ${s//:/ }
  RSIM_LONE_K=$1 RSIM_LONE_LDS=$2 timeout 300 python bench.py --config $CFG ${EXTRA:---steps 100 --warmup 10} --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/${TAG}_${CFG}_$1_$2_$rep.json 2> $O/${TAG}.err || tail -3 $O/${TAG}.err
  python - $O/${TAG}_${CFG}_$1_$2_$rep.json "$s" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("lone", sys.argv[2], "value %.0f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["step_ms"].items()}, "reward_sum", d["config"].get("reward_sum"), "tier", d["tier_env_steps"], d["tier_changes_in_mid_step"])
PY
done; done
