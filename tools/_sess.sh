cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1 2 3; do
  export RSIM_COST_DECAY=$v
  echo "== decay $v"; bash tools/gpu_session.sh r04_r_d$v quick:lift 2>&1 | grep value | cut -c1-100
done; done
for v in 0 2; do export RSIM_COST_DECAY=$v; echo "== decay $v"; bash tools/gpu_session.sh r04_r_d$v quick:stack quick:peg 2>&1 | grep value | cut -c1-100; done
