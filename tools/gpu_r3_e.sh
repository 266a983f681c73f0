#!/bin/bash
# round 3, GPU session 5: full GPU tests (solo build removed, geometry-fed oracle in the full-size tests), Newton rules on identical states, kernel trace of the lockstep bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r3e_pytest.log 2>&1; grep -E "passed|failed|Error|error|oracle fed|arm:|gripper:|objects:|PickPlace step|Baxter step|assert" gpurun_out/r3e_pytest.log | cut -c1-400 | tail -40
timeout 900 python tools/newton_sweep.py 200 "0,0,0" "1e-6,1e-7,0" "1e-5,1e-6,0" "1e-4,1e-5,0" "1e-3,1e-4,0" "1e-5,1e-5,0" "1e-4,1e-4,0" > gpurun_out/r3e_newton_sweep.txt 2>&1; cat gpurun_out/r3e_newton_sweep.txt | cut -c1-700
rm -rf gpurun_out/prof_r3e
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r3e -o r -- python bench.py --no-cpu-baseline --no-open-loop --steps 60 --warmup 5 > gpurun_out/r3e_trace.log 2>&1
python - <<'PY'
import csv, glob, numpy as np
f = glob.glob("gpurun_out/prof_r3e/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "") for r in rows]
st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
# the last 60 control steps: sequences k_order -> k_step -> k_prepare -> k_reset_obs
idx = [i for i, n in enumerate(names) if n == "k_step"][-60:]
out = {}
for i in idx:
    seq = {names[j]: j for j in range(max(0, i - 1), min(len(names), i + 3))}
    out.setdefault("k_step_us", []).append((en[i] - st[i]) / 1e3)
    if names[i - 1] == "k_order":
        out.setdefault("k_order_us", []).append((en[i - 1] - st[i - 1]) / 1e3); out.setdefault("gap_order_to_step_us", []).append((st[i] - en[i - 1]) / 1e3)
    if i + 1 < len(names) and names[i + 1] == "k_prepare":
        out.setdefault("gap_step_to_prepare_us", []).append((st[i + 1] - en[i]) / 1e3); out.setdefault("k_prepare_us", []).append((en[i + 1] - st[i + 1]) / 1e3)
    if i + 2 < len(names) and names[i + 2] == "k_reset_obs":
        out.setdefault("gap_prepare_to_resetobs_us", []).append((st[i + 2] - en[i + 1]) / 1e3); out.setdefault("k_reset_obs_us", []).append((en[i + 2] - st[i + 2]) / 1e3)
    if i + 3 < len(names) and names[i + 3] == "k_order":
        out.setdefault("gap_resetobs_to_next_order_us", []).append((st[i + 3] - en[i + 2]) / 1e3)
step = np.diff([st[i] for i in idx]) / 1e3
print("lockstep control step, kernel trace (us):", {k: round(float(np.mean(v)), 1) for k, v in out.items()}, "step period", round(float(step.mean()), 1))
PY
tail -c 600 gpurun_out/r3e_trace.log
rm -rf gpurun_out/prof_r3e
