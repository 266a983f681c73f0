cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_session.sh r04_o tests 2>&1 | tail -12 | cut -c1-220
for rep in 1 2; do for v in 0 1; do
  export RSIM_NEWTON_EXACT=$v
  echo "== exact $v"; bash tools/gpu_session.sh r04_o_x$v quick:lift quick:stack quick:peg 2>&1 | grep value | cut -c1-120
done; done
for v in 0 1; do export RSIM_NEWTON_EXACT=$v; echo "== exact $v"; bash tools/gpu_session.sh r04_o_x$v quick:pickplace 2>&1 | grep value | cut -c1-120; done
