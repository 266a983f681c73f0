#!/bin/bash
# round-5 final evidence on the shipped build: suite, PMC passes of the four BASELINE configurations (keyed to the library sha), bench lines, kernel stats, PickPlace parity at 256 envs,
# and the driver-like default bench command (headline + config.other_configs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_z probe tests pmc:lift pmc:stack pmc:peg pmc:pickplace bench:lift bench:stack bench:peg bench:pickplace stats:lift stats:stack stats:peg stats:pickplace || exit 3
RSIM_PARITY_SAMPLE=256 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_z_parity_pickplace.txt 2>&1
grep -E "polish exits|oracle fed|unfinished|stopped short|passed|failed|^E  " $O/r05_z_parity_pickplace.txt | cut -c1-420 | head -10
for k in gripper objects "rel dforce" "objective gap"; do grep -E "$k per env" $O/r05_z_parity_pickplace.txt | awk '{n=NF; printf "   %s tail:", $1; for(i=n-9;i<=n;i++) printf " %s", $i; print ""}'; done
echo "=== driver-like bench"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_z_bench_driver_like.json 2> $O/r05_z_bench_driver_like.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_z_bench_driver_like.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "issue", (d["roofline"].get("issue") or {}).get("frac") if isinstance(d["roofline"].get("issue"), dict) else d["roofline"].get("issue"), "traffic", d["roofline"]["traffic"])
for k, v in (d["config"].get("other_configs") or {}).items(): print(k, {a: v.get(a) for a in ("value", "ms_per_step", "overflow_envs", "diverged_envs", "issue_frac", "error")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
