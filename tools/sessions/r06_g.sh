#!/bin/bash
# round 6, session g: the smooth-pair restart cone (RSIM_MPR_CONE) -- GPU suite on the build, then Lift quick lines at cone 0 (off) / 0.003 / 0.01 / 0.03, round robin x 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r06_g_pytest_gpu.txt 2>&1; tail -4 $O/r06_g_pytest_gpu.txt | cut -c1-300
for rep in 1 2; do for cone in 0 0.003 0.01 0.03; do
  RSIM_MPR_CONE=$cone timeout 300 python bench.py --config lift --steps 100 --warmup 10 --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/r06_g_cone_${cone}_$rep.json 2> $O/r06_g_cone.err
  python - $O/r06_g_cone_${cone}_$rep.json $cone <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"cone {sys.argv[2]:>6}: {d['value']/1e3:8.1f} K env-steps/s  {d['ms_per_step']:.3f} ms/step  step_ms {d['step_ms']}  reward_sum {d['config']['reward_sum']:.3f} diverged {d['config']['diverged_envs']}")
PY
done; done
