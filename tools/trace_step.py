"""Timeline of the kernels of one lockstep control step from a `rocprofv3 --kernel-trace` of bench.py: every dispatch between two consecutive k_step launches
of the full batch (name, queue, start relative to the step's k_step, duration, gap to the previous kernel's end on the same queue), averaged over the last N steps.
Usage (GPU box): python tools/trace_step.py <rocprof output dir> [N=20]"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
grid = lambda r: int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)   # noqa: E731
short = lambda n: n.split("<")[0].split("(")[0][:28]                     # noqa: E731
main = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_step<") or r["Kernel_Name"].startswith("k_step<")]
gmax = max(grid(rows[i]) for i in main)
main = [i for i in main if grid(rows[i]) == gmax]
steps = main[-(N + 1):]
acc = {}
period = []
for a, b in zip(steps[:-1], steps[1:]):
    t0 = rows[a]["s"]
    period.append(rows[b]["s"] - t0)
    seen = {}
    for r in rows[a:b]:
        k = short(r["Kernel_Name"])
        seen[k] = seen.get(k, 0) + 1
        key = (k, seen[k])
        acc.setdefault(key, []).append(((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r.get("Queue_Id"), grid(r)))
print(f"{len(period)} control steps, period mean {np.mean(period) / 1e3:.1f} us (k_step start to k_step start)")
for key, v in sorted(acc.items(), key=lambda kv: np.mean([x[0] for x in kv[1]])):
    print(f"  {key[0]:28s} #{key[1]}  start {np.mean([x[0] for x in v]):9.1f} us  dur {np.mean([x[1] for x in v]):8.1f} us  queue {v[0][2]}  grid {v[0][3]}  ({len(v)} of {len(period)} steps)")
