#!/bin/bash
# Round 5, session 16 (GPU side; 2.7 GPU-minutes left): the default bench command with the longer secondary regions (bench.py OTHER_REGION), then the full-protocol Stack line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; T=r05_s16
sha256sum robosuite_amd/librsim_hip.so | cut -c1-16 | tee $O/${T}_log.txt
timeout 100 python bench.py > $O/${T}_bench_lift.json 2> $O/${T}_bench_lift.err
python - <<'PY' | tee -a gpurun_out/r05_s16_log.txt
import json
try:
    d = json.loads(open("gpurun_out/r05_s16_bench_lift.json").read().strip().splitlines()[-1])
    print("lift", round(d["value"]), d["ms_per_step"], "dbuf", round(d["config"]["double_buffered"]["value"]), "traffic", d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"])
    for k, v in (d["config"].get("other_configs") or {}).items(): print("other", k, {a: v.get(a) for a in ("value", "ms_per_step", "steps", "preroll", "overflow_envs", "diverged_envs", "issue_frac", "error")})
except Exception as e:
    print("bench line unreadable:", e)
PY
echo "[s16] default bench done at $SECONDS s" | tee -a $O/${T}_log.txt
[ $SECONDS -lt 95 ] && bash tools/gpu_session.sh $T bench:stack:"--no-cpu-baseline" 2>&1 | tail -3 | cut -c1-400 | tee -a $O/${T}_log.txt
echo "[s16] end at $SECONDS s" | tee -a $O/${T}_log.txt
