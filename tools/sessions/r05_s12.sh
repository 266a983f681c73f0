#!/bin/bash
# Round 5, session 12 (GPU side; ~10 GPU-minutes were left for the round, box acquisition included): Stack with six / seven envs per CU (tools/sessions/r05_s12_prep.sh).
# In order of priority, every step under its own timeout and skipped when the session's clock says it no longer fits:
#   1. lockstep A/B, round-robin, two reps: default | s6 | s7 on Stack @4096 (tools/ab_many.sh)
#   2. the -m gpu suite on the faster variant (RSIM_LIB), if it beats the default by more than 3 %
#   3. staged Stack probe (64 envs, ten steps) on default and variant: the printed state sums say whether the variant is bit-identical
#   4. instruction-cache PMC pass of the Lift kernel (tools/pmc_pass.sh ic): SQ_WAIT_INST_ANY is 15 % of the wave cycles and had never been broken down
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; T=r05_s12; LIMIT=${SESSION_LIMIT:-480}
left() { echo $((LIMIT - SECONDS)); }
echo "[s12] start, limit $LIMIT s" | tee $O/${T}_log.txt
rm -f $O/${T}_ab_many_stack.txt
bash tools/ab_many.sh $T stack 2 librsim_hip.so librsim_hip_s6.so librsim_hip_s7.so
echo "[s12] A/B done at $SECONDS s" | tee -a $O/${T}_log.txt
best=$(python - <<'PY'
import re, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/r05_s12_ab_many_stack.txt"):
    m = re.match(r"(\S+) stack rep \d+ ms/step ([\d.]+)", l)
    if m: d[m.group(1)].append(float(m.group(2)))
mean = {k: sum(v) / len(v) for k, v in d.items()}
base = mean.get("librsim_hip.so")
cand = {k: v for k, v in mean.items() if k != "librsim_hip.so"}
if base and cand:
    k = min(cand, key=cand.get)
    print(k if cand[k] < 0.97 * base else "none")
else:
    print("none")
PY
)
echo "[s12] best variant: $best" | tee -a $O/${T}_log.txt
if [ "$best" != "none" ] && [ $(left) -gt 150 ]; then
  RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$best timeout $(( $(left) - 20 )) python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu_variant.txt 2>&1
  tail -3 $O/${T}_pytest_gpu_variant.txt | cut -c1-300 | tee -a $O/${T}_log.txt
fi
echo "[s12] suite done at $SECONDS s" | tee -a $O/${T}_log.txt
if [ $(left) -gt 60 ]; then
  for L in librsim_hip.so ${best/none/librsim_hip_s6.so}; do
    [ $(left) -gt 30 ] && RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$L timeout 60 python tools/stack_probe.py Stack 64 2>&1 | grep "stack_probe" | tail -4 | sed "s|^|$L |" | tee -a $O/${T}_stack_probe.txt
  done
fi
echo "[s12] probe done at $SECONDS s" | tee -a $O/${T}_log.txt
if [ $(left) -gt 100 ]; then
  PMC_TIMEOUT=$(( $(left) - 10 )) bash tools/pmc_pass.sh ${T}_lift ic 2>&1 | tail -12 | tee -a $O/${T}_log.txt
fi
echo "[s12] end at $SECONDS s" | tee -a $O/${T}_log.txt
