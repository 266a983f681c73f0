#!/bin/bash
# round 6, session o: the 256-row tier reads the native constant blocks (no k_prepare before a wide pass): PickPlace GPU tests, then A/B by run-time switch RSIM_NO_SHARE_CM
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -k "pickplace or pick_place or tier or capacity" > $O/r06_o_pytest_gpu.txt 2>&1; tail -6 $O/r06_o_pytest_gpu.txt | head -3 | cut -c1-300
for rep in 1 2; do for sw in 1 0; do
  if [ $sw = 1 ]; then export RSIM_NO_SHARE_CM=1; else unset RSIM_NO_SHARE_CM; fi
  timeout 400 python bench.py --config pickplace --steps 20 --warmup 3 --preroll 100 --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/r06_o_ab_noshare${sw}_$rep.json 2> $O/r06_o_ab.err
  python - $O/r06_o_ab_noshare${sw}_$rep.json $sw <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"RSIM_NO_SHARE_CM={sys.argv[2]}: {d['value']/1e3:8.2f} K env-steps/s  {d['ms_per_step']:.2f} ms/step  step_ms p50 {d['step_ms']['p50']:.1f} p90 {d['step_ms']['p90']:.1f} max {d['step_ms']['max']:.1f}  tier {d['tier_env_steps']} / {d['tier_changes_in_mid_step']}  diverged {d['config']['diverged_envs']} overflow {d['config']['overflow_envs']}")
PY
done; done
