#!/bin/bash
# Round 5, session 17 (GPU side; 1.8 GPU-minutes left): the Stack secondary region of the default command reads 588 - 603 K where the stand-alone protocols read 627 - 647 K
# in the same session.  Same region stand-alone (no parent process holding the headline batch), inside a (shortened) default command, stand-alone again.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
O=gpurun_out; T=r05_s17
one() { timeout 60 python bench.py --secondary-only stack 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 stand-alone stack', round(d['value']), d['ms_per_step'], d['steps'], d['preroll'])" | tee -a $O/${T}_log.txt; }
: > $O/${T}_log.txt
one first
timeout 70 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --preroll 50 --no-open-loop 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for k, v in d['config']['other_configs'].items(): print('child of the default command', k, round(v.get('value', 0)), v.get('ms_per_step'), v.get('error'))" | tee -a $O/${T}_log.txt
[ $SECONDS -lt 75 ] && one second
echo "[s17] end at $SECONDS s" | tee -a $O/${T}_log.txt
