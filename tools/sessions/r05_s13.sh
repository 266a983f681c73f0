#!/bin/bash
# Round 5, session 13 (GPU side; 8.7 GPU-minutes left): Stack on top of the seven-envs-per-CU default -- eight per CU (s8), a smaller capacity tier (t7a / t7b) and both.
#   1. lockstep A/B, round-robin, two reps (tools/ab_many.sh)          2. the -m gpu suite on the fastest build (the default when nothing beats it by 2 %)
#   3. staged Stack probe, 64 envs: state sums of default / s8 / chosen (bit-identity; the LDS build of session 12 printed 259.5531005859375)
#   4. full-protocol Stack bench line + rocprofv3 kernel stats on the chosen build   5. PMC passes (VALU / SQ, FETCH_SIZE, WRITE_SIZE) on the chosen build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; T=r05_s13; LIMIT=${SESSION_LIMIT:-470}
left() { echo $((LIMIT - SECONDS)); }
echo "[s13] start, limit $LIMIT s" | tee $O/${T}_log.txt
rm -f $O/${T}_ab_many_stack.txt
bash tools/ab_many.sh $T stack 2 librsim_hip.so librsim_hip_s8.so librsim_hip_t7a.so librsim_hip_s8t7a.so librsim_hip_s8t7b.so
echo "[s13] A/B done at $SECONDS s" | tee -a $O/${T}_log.txt
best=$(python - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r05_s13_ab_many_stack.txt"):
    m = re.match(r"(\S+) stack rep \d+ ms/step ([\d.]+)", l)
    if m: d[m.group(1)].append(float(m.group(2)))
mean = {k: sum(v) / len(v) for k, v in d.items()}
base = mean.get("librsim_hip.so")
cand = {k: v for k, v in mean.items() if k != "librsim_hip.so"}
k = min(cand, key=cand.get) if cand else None
print(k if (k and base and cand[k] < 0.98 * base) else "librsim_hip.so")
PY
)
echo "[s13] chosen build: $best" | tee -a $O/${T}_log.txt
export RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$best
if [ $(left) -gt 120 ]; then
  timeout $(( $(left) - 30 )) python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu_chosen.txt 2>&1
  grep -E "passed|failed" $O/${T}_pytest_gpu_chosen.txt | tail -2 | cut -c1-300 | tee -a $O/${T}_log.txt
fi
echo "[s13] suite done at $SECONDS s" | tee -a $O/${T}_log.txt
for L in librsim_hip.so librsim_hip_s8.so $best; do
  [ $(left) -gt 40 ] && RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$L timeout 60 python tools/stack_probe.py Stack 64 2>&1 | grep "stack_probe" | tail -2 | sed "s|^|$L |" | tee -a $O/${T}_stack_probe.txt
done
echo "[s13] probe done at $SECONDS s" | tee -a $O/${T}_log.txt
if [ $(left) -gt 90 ]; then
  bash tools/gpu_session.sh $T bench:stack:"--steps 100 --warmup 10 --no-cpu-baseline" stats:stack 2>&1 | tail -12 | cut -c1-300 | tee -a $O/${T}_log.txt
fi
echo "[s13] bench + stats done at $SECONDS s" | tee -a $O/${T}_log.txt
if [ $(left) -gt 100 ]; then
  PMC_TIMEOUT=$(( ($(left) - 20) / 3 )) bash tools/gpu_session.sh $T pmc:stack 2>&1 | tail -30 | cut -c1-300 | tee -a $O/${T}_log.txt
fi
echo "[s13] end at $SECONDS s" | tee -a $O/${T}_log.txt
