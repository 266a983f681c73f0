#!/bin/bash
# quick lockstep bench of several library builds / settings in ONE session, round-robin:  ab_many.sh <tag> <cfg> <reps> item1 item2 ...
# item = lib.so[@ENV=VALUE[,ENV2=VALUE2]]   (libs relative to robosuite_amd/)
tag=$1; cfg=$2; reps=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
case $cfg in pickplace) extra="--steps 12 --warmup 3 --preroll 60";; lift) extra="--steps 100 --warmup 10";; *) extra="--steps 50 --warmup 5 --preroll 300";; esac
for rep in $(seq $reps); do for item in "$@"; do
  lib=${item%%@*}; envs=""; [[ "$item" == *@* ]] && envs=$(echo "${item#*@}" | tr ',' ' ')
  env $envs RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib timeout 400 python bench.py --config $cfg $extra --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs 2> gpurun_out/${tag}_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$item', '$cfg', 'rep $rep', 'ms/step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], 'overflow', d['config'].get('overflow_envs'))" | tee -a gpurun_out/${tag}_ab_many_$cfg.txt
done; done
