#!/bin/bash
# round 6, session m: Stack's tier fused into its kernel (+ slot-wise row staging, packed descriptor words in every configuration): GPU suite, Stack A/B against the
# same source without -DRSIM_FUSED_TIER for configuration 1 (A B A B A B, quick protocol), window trace of Stack, quick lines of the other three configurations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $O/r06_m_pytest_gpu.txt 2>&1; tail -6 $O/r06_m_pytest_gpu.txt | head -4 | cut -c1-300
for rep in 1 2 3; do for lib in librsim_hip_stack_unfused.so librsim_hip.so; do
  RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib timeout 300 python bench.py --config stack --steps 100 --warmup 10 --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/r06_m_ab_${lib%.so}_$rep.json 2> $O/r06_m_ab.err
  python - $O/r06_m_ab_${lib%.so}_$rep.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:30s}: {d['value']/1e3:8.1f} K env-steps/s  {d['ms_per_step']:.3f} ms/step  step_ms p50 {d['step_ms']['p50']:.2f} p90 {d['step_ms']['p90']:.2f} max {d['step_ms']['max']:.2f}  tier {d['tier_env_steps']} / {d['tier_changes_in_mid_step']}  reward_sum {d['config']['reward_sum']:.3f} diverged {d['config']['diverged_envs']} overflow {d['config']['overflow_envs']}")
PY
done; done
timeout 300 python tools/window_trace.py --config stack --prerolls 500 2>&1 | grep -v amdgpu | cut -c1-170
bash tools/gpu_session.sh r06_m quick:lift quick:peg quick:pickplace
