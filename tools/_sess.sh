cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 5 9 13 17 3 1; do
  export RSIM_TIER_SKIP=$v
  echo "== skip $v"; bash tools/gpu_session.sh r04_j_$v quick:lift 2>&1 | grep value | cut -c1-110
done
