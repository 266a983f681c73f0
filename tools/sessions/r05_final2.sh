#!/bin/bash
# round-5 closing evidence on the shipped build (Stack fault fixed): PMC passes of the four BASELINE configurations (keyed to the library sha), the default bench command
# (headline + config.other_configs in child processes), the other three bench lines, kernel stats, smoke(), then the PickPlace parity at 256 sampled envs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
export PMC_TIMEOUT=90
bash tools/gpu_session.sh r05_zz probe pmc:lift pmc:stack pmc:peg pmc:pickplace bench:lift bench:stack bench:peg bench:pickplace stats:lift stats:stack stats:peg stats:pickplace || exit 3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_zz_bench_lift.json").read().strip().splitlines()[-1])
for k, v in (d["config"].get("other_configs") or {}).items(): print("other", k, {a: v.get(a) for a in ("value", "ms_per_step", "overflow_envs", "diverged_envs", "issue_frac", "error")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"), "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "kernel_ms")})
PY
echo "=== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-300
echo "=== parity 256"
RSIM_PARITY_SAMPLE=256 timeout 420 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace_8192" > $O/r05_zz_parity_pickplace.txt 2>&1
grep -E "polish exits|oracle fed|unfinished|descent direction|stopped short|passed|failed|^E  " $O/r05_zz_parity_pickplace.txt | cut -c1-420 | head -10
