#!/bin/bash
# round-6 final evidence on the shipped build: GPU suite, PMC passes of the four BASELINE configurations (keyed to the library / code-object sha), kernel stats, the driver's
# windows with tier counts, the driver's exact command, the full-protocol bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
TAG=${TAG:-r06_zz}
export TAG_FILE=gpurun_out/${TAG}_bench_driver_like.json
bash tools/gpu_session.sh ${TAG} probe tests pmc:lift pmc:stack pmc:peg pmc:pickplace stats:lift stats:stack stats:peg stats:pickplace || exit 3
echo "=== windows"
timeout 600 python tools/window_trace.py --prerolls 500,700,900,1100 --out $O/${TAG}_window_trace.json > $O/${TAG}_window_trace.txt 2>&1; grep -v amdgpu.ids $O/${TAG}_window_trace.txt | cut -c1-200
echo "=== driver-like bench"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_like.json 2> $O/${TAG}_bench_driver_like.err ) 2>&1 | tail -3
python - <<'PY'
import json, os
d = json.loads(open(os.environ["TAG_FILE"]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["step_ms"], "tier", d["tier_env_steps"], d["tier_changes_in_mid_step"], "issue", (d["roofline"].get("issue") or {}).get("frac") if isinstance(d["roofline"].get("issue"), dict) else d["roofline"].get("issue"), "traffic", d["roofline"]["traffic"])
for k, v in (d["config"].get("other_configs") or {}).items(): print(k, {a: v.get(a) for a in ("value", "ms_per_step", "step_ms", "overflow_envs", "diverged_envs", "issue_frac", "traffic", "error")}, (v.get("double_buffered") or {}).get("value"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
echo "=== full-protocol lines"
bash tools/gpu_session.sh ${TAG} "bench:lift:--no-other-configs" bench:stack bench:peg bench:pickplace
