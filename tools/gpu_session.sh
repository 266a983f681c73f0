#!/bin/bash
# One parametrised GPU-box session script (replaces the single-use tools/gpu_r3_*.sh of round 3).  A session is a list of steps, run in order under
# tight time limits; every step writes gpurun_out/<tag>_<step>...  Boxes differ by +-15 %, so everything that is compared runs in ONE session.
#
#   gpurun --timeout 900 -- 'bash tools/gpu_session.sh r04_a probe tests bench:lift bench:stack:"--steps 50 --warmup 5"'
#
# steps (arguments after ':' are passed on):
#   probe                       tools/box_probe.py; a faulty box ends the session (exit 3)
#   tests[:pytest args]         python -m pytest tests -m gpu -q
#   bench:<cfg>[:args]          python bench.py --config <cfg> args           -> <tag>_bench_<cfg>.json
#   quick:<cfg>[:args]          bench without the cpu / open-loop / double-buffered legs (lockstep figure only) -> <tag>_quick_<cfg>.json
#   stats:<cfg>[:args]          rocprofv3 --kernel-trace --stats of a short quick bench   -> <tag>_kernel_stats_<cfg>.csv
#   pmc:<cfg>[:sets]            tools/pmc_pass.sh (default sets: sq1 hbm1 hbm2), then valu_count / hbm_traffic json keyed to this build
#   tail[:step]                 tools/tail_report.py on robosuite_amd/librsim_hip_prof.so (built by tools/subprof.sh mpr)
#   phase:<task>                tools/phase_profile_task.py <task> (per-phase shares)
#   ab:<libA>:<libB>[:cfg[:args]]  quick bench of two builds, A B A B (libs relative to robosuite_amd/)
#   py:<script>[:args]          python <script> args > <tag>_<script stem>.txt
set -u
tag=${1:-sess}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
bench_extra() { case $1 in pickplace) echo "--steps 30 --warmup 5 --preroll 100";; lift) echo "";; *) echo "--steps 100 --warmup 10";; esac; }
quick_extra() { case $1 in pickplace) echo "--steps 12 --warmup 3 --preroll 60";; lift) echo "--steps 100 --warmup 10";; *) echo "--steps 50 --warmup 5 --preroll 300";; esac; }
line() { python - "$1" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]; r = d["roofline"]
    print(f"{sys.argv[1]}: value {d['value']:.0f} ms/step {d['ms_per_step']:.3f} kernel_ms {r['kernel_ms']:.3f} overflow_envs {c.get('overflow_envs')} capacity {c.get('capacity')} diverged {c.get('diverged_envs')} "
          f"dbuf {(c.get('double_buffered') or {}).get('value')} open {(c.get('open_loop') or {}).get('value')} issue {(r.get('issue') or {}).get('frac') if isinstance(r.get('issue'), dict) else r.get('issue')} traffic {r.get('traffic')}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
EOF
}
for step in "$@"; do
  IFS=: read -r kind a1 a2 a3 a4 <<< "$step"
  echo "=== $step"
  case $kind in
    probe)
      timeout 180 python tools/box_probe.py > $O/${tag}_box_probe.txt 2>&1; rc=$?; head -4 $O/${tag}_box_probe.txt | cut -c1-160
      if [ $rc -ne 0 ]; then echo "probe rc $rc: stopping"; exit 3; fi;;
    tests)
      timeout 1200 python -m pytest tests -m gpu -q ${a1:-} > $O/${tag}_pytest_gpu.txt 2>&1; tail -15 $O/${tag}_pytest_gpu.txt | cut -c1-300;;
    bench)
      timeout 900 python bench.py --config $a1 ${a2:-$(bench_extra $a1)} > $O/${tag}_bench_$a1.json 2> $O/${tag}_bench_$a1.err; line $O/${tag}_bench_$a1.json; tail -3 $O/${tag}_bench_$a1.err | cut -c1-300;;
    quick)
      timeout 400 python bench.py --config $a1 ${a2:-$(quick_extra $a1)} --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/${tag}_quick_$a1.json 2> $O/${tag}_quick_$a1.err; line $O/${tag}_quick_$a1.json; tail -3 $O/${tag}_quick_$a1.err | cut -c1-300;;
    stats)
      rm -rf $O/prof_${tag}_$a1
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_$a1 -o r -- python bench.py --config $a1 ${a2:-$(quick_extra $a1)} --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/${tag}_prof_$a1.log 2>&1
      cp $(find $O/prof_${tag}_$a1 -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats_$a1.csv; head -5 $O/${tag}_kernel_stats_$a1.csv | cut -c1-200; rm -rf $O/prof_${tag}_$a1;;
    pmc)
      export RSIM_CONFIG=$a1 PMC_TIMEOUT=${PMC_TIMEOUT:-150}
      case $a1 in pickplace) export RSIM_BENCH_EXTRA="--preroll 40";; lift) export RSIM_BENCH_EXTRA="";; *) export RSIM_BENCH_EXTRA="--preroll 200";; esac
      sets=${a2:-sq1 hbm1 hbm2}
      KEEP=1 bash tools/pmc_pass.sh ${tag}_$a1 $sets
      [[ " $sets " == *" sq1 "* ]] && python tools/pmc_valu.py $O/${tag}_$a1.sq1 4
      [[ " $sets " == *" hbm1 "* ]] && python tools/pmc_traffic.py $O/${tag}_$a1.hbm1 $O/${tag}_$a1.hbm2 4
      for s in $sets; do rm -rf $O/${tag}_$a1.$s; done
      cp profiles/valu_count*.json profiles/hbm_traffic*.json $O/ 2>/dev/null;;
    tail)
      RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/librsim_hip_prof.so timeout 400 python tools/tail_report.py ${a1:-200} > $O/${tag}_tail_report.txt 2>&1; head -14 $O/${tag}_tail_report.txt | cut -c1-400;;
    phase)
      timeout 300 python tools/phase_profile_task.py $a1 > $O/${tag}_phase_$a1.txt 2>&1; tail -8 $O/${tag}_phase_$a1.txt | cut -c1-400;;
    ab)
      cfg=${a3:-lift}
      for rep in 1 2; do for lib in $a1 $a2; do
        RSIM_LIB=$GRAFT_REPO_ROOT/robosuite_amd/$lib timeout 400 python bench.py --config $cfg ${a4:-$(quick_extra $cfg)} --no-open-loop --no-cpu-baseline --no-double-buffer --no-other-configs > $O/${tag}_ab_${cfg}_${lib%.so}_$rep.json 2> $O/${tag}_ab.err
        line $O/${tag}_ab_${cfg}_${lib%.so}_$rep.json
      done; done;;
    py)
      stem=$(basename ${a1%.py})
      timeout ${PY_TIMEOUT:-600} python $a1 ${a2:-} > $O/${tag}_$stem.txt 2>&1; tail -${PY_TAIL:-30} $O/${tag}_$stem.txt | cut -c1-400;;
    *) echo "unknown step $step";;
  esac
done
