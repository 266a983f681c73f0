#!/bin/bash
# round 3, session q: bench lines with the double-buffered closed-loop leg (library unchanged: the PMC files stay valid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/box_probe.py > gpurun_out/r3q_box_probe.txt 2>&1; rc=$?; cat gpurun_out/r3q_box_probe.txt; if [ $rc -eq 3 ]; then echo 'faulty box: stopping'; exit 3; fi
timeout 600 python bench.py > gpurun_out/r03_z_bench.json 2> gpurun_out/r03_z_bench.err; tail -c 1200 gpurun_out/r03_z_bench.json
for c in stack peg pickplace; do
  case $c in pickplace) extra="--steps 30 --warmup 5 --preroll 100";; *) extra="--steps 100 --warmup 10";; esac
  timeout 900 python bench.py --config $c $extra > gpurun_out/r03_z_bench_$c.json 2> gpurun_out/r03_z_bench_$c.err; tail -c 300 gpurun_out/r03_z_bench_$c.json
done
python - <<'PY'
import json
for c in ("", "_stack", "_peg", "_pickplace"):
    d = json.loads(open(f"gpurun_out/r03_z_bench{c}.json").read().strip().splitlines()[-1]); g = d["config"]
    print(c or "lift", "lockstep %.0f" % d["value"], "open loop %.0f" % g["open_loop"]["value"], "double-buffered %.0f" % g["double_buffered"]["value"], "stale", g["reset_ring"]["bank_stale"], g["double_buffered"]["bank_stale"], "issue", d["roofline"]["issue"] and round(d["roofline"]["issue"]["frac"], 4))
PY
