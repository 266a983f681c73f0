cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 0 1 2 3; do
  export RSIM_NEWTON_REFINE=$r
  echo "== refine $r"
  python -m pytest tests -m gpu -q -s -k pickplace_8192 > gpurun_out/r04_u_pp_r$r.txt 2>&1
  grep -E "gripper:|objects:|arm:|oracle fed|passed|failed|^E  " gpurun_out/r04_u_pp_r$r.txt | cut -c1-230 | head -8
done
for r in 0 2 0 2; do export RSIM_NEWTON_REFINE=$r; echo "== refine $r"; bash tools/gpu_session.sh r04_u_r$r quick:pickplace 2>&1 | grep value | cut -c1-100; python -c "
import json; d=json.loads(open('gpurun_out/r04_u_r${r}_quick_pickplace.json').read().strip().splitlines()[-1]); print('diverged', d['config']['diverged_envs'], d['config']['capacity']['max_rows_needed'])"; done
