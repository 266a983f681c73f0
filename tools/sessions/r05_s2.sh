#!/bin/bash
# round-5 session 2: the adopted build (MGLOBAL, fp64-evaluated polish, qfrc_applied, advisor fixes) through the whole GPU suite; JG256 on top of it and the number of
# polish passes A/B on PickPlace; PickPlace full-size parity on a 128-env sample at 1 and 8 passes; the driver-like default bench line (other_configs included)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out
bash tools/gpu_session.sh r05_b probe tests || exit 3
bash tools/ab_many.sh r05_b pickplace ${REPS:-5} librsim_hip.so librsim_hip_jg256.so librsim_hip.so@RSIM_NEWTON_REFINE=8 librsim_hip.so@RSIM_NEWTON_REFINE=3
echo "=== suite subset on jg256"
RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_jg256.so timeout 600 python -m pytest tests -m gpu -q -x -k "pickplace or PickPlace or pick_place or tendon" > $O/r05_b_pytest_jg256.txt 2>&1; tail -3 $O/r05_b_pytest_jg256.txt | cut -c1-300
for R in 1 3 8; do
  RSIM_NEWTON_REFINE=$R RSIM_PARITY_SAMPLE=128 timeout 900 python -m pytest tests/test_full_size_parity.py -m gpu -q -s -k "pickplace" > $O/r05_b_parity_pickplace_R$R.txt 2>&1
  echo "=== parity, $R polish passes"; grep -E "per env, relative, sorted|objective gap per env|rel dforce per env|passed|failed" $O/r05_b_parity_pickplace_R$R.txt | cut -c1-1800
done
echo "=== driver-like bench"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_b_bench_driver_like.json 2> $O/r05_b_bench_driver_like.err; tail -c 3000 $O/r05_b_bench_driver_like.json; tail -3 $O/r05_b_bench_driver_like.err
