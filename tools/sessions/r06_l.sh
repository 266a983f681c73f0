#!/bin/bash
# round 6, session l: one-instruction DPP steps in wave_max / wave_min_i + select-based hull scans: GPU suite, then Lift A/B against the build before (cfg0 object of the
# previous commit), A B A B A B on the 100-step protocol, and the three other configurations' quick lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > $O/r06_l_pytest_gpu.txt 2>&1; tail -4 $O/r06_l_pytest_gpu.txt | cut -c1-300
for rep in 1 2 3; do for lib in librsim_hip_prev.so librsim_hip.so; do
  RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib timeout 300 python bench.py --config lift --steps 100 --warmup 10 --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/r06_l_ab_${lib%.so}_$rep.json 2> $O/r06_l_ab.err
  python - $O/r06_l_ab_${lib%.so}_$rep.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:24s}: {d['value']/1e3:8.1f} K env-steps/s  {d['ms_per_step']:.3f} ms/step  step_ms p50 {d['step_ms']['p50']:.2f} p90 {d['step_ms']['p90']:.2f} max {d['step_ms']['max']:.2f}  reward_sum {d['config']['reward_sum']:.3f}")
PY
done; done
bash tools/gpu_session.sh r06_l quick:stack quick:peg quick:pickplace
