#!/bin/bash
# Round 5, session 15 (GPU side; 4.9 GPU-minutes left): closing evidence on the adopted build (Stack: eight envs per CU, the smaller tier, nothing flagged in advance):
# the -m gpu suite, the default bench command (what the driver runs: Lift headline + config.other_configs), the Stack line and kernel stats if the clock allows.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; T=r05_s15; LIMIT=${SESSION_LIMIT:-260}
left() { echo $((LIMIT - SECONDS)); }
sha256sum robosuite_amd/librsim_hip.so | cut -c1-16 | tee $O/${T}_log.txt
timeout 150 python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/${T}_pytest_gpu.txt | tail -2 | cut -c1-300 | tee -a $O/${T}_log.txt
echo "[s15] suite done at $SECONDS s" | tee -a $O/${T}_log.txt
if [ $(left) -gt 60 ]; then
  timeout $(( $(left) - 5 )) python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
  python - <<'PY' | tee -a gpurun_out/r05_s15_log.txt
import json
try:
    d = json.loads(open("gpurun_out/r05_s15_bench_default.json").read().strip().splitlines()[-1])
    print("lift", round(d["value"]), d["ms_per_step"], "dbuf", round(d["config"]["double_buffered"]["value"]), "traffic", d["roofline"]["traffic"], "issue", d["roofline"]["issue"] if isinstance(d["roofline"]["issue"], str) else d["roofline"]["issue"]["frac"])
    for k, v in (d["config"].get("other_configs") or {}).items(): print("other", k, {a: v.get(a) for a in ("value", "ms_per_step", "overflow_envs", "diverged_envs", "issue_frac", "error")})
except Exception as e:
    print("bench line unreadable:", e)
PY
fi
echo "[s15] bench done at $SECONDS s" | tee -a $O/${T}_log.txt
if [ $(left) -gt 45 ]; then
  bash tools/gpu_session.sh $T stats:stack 2>&1 | tail -6 | cut -c1-260 | tee -a $O/${T}_log.txt
fi
if [ $(left) -gt 40 ]; then
  bash tools/gpu_session.sh $T quick:stack 2>&1 | tail -3 | cut -c1-300 | tee -a $O/${T}_log.txt
fi
echo "[s15] end at $SECONDS s" | tee -a $O/${T}_log.txt
