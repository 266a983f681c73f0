"""How two alternating half-batches fill the chip: from a `rocprofv3 --kernel-trace` of bench.py, the k_step dispatches of the double-buffered region (the last
2 K control-step launches of the run, 2048-env grids on two queues) -- how long a launch lasts, how much of it overlaps the other half's launch, and the
fraction of the region during which 0 / 1 / 2 control-step kernels are in flight.  For comparison the lockstep region (4096-env grids).
Usage (GPU box): python tools/trace_double_buffer.py <rocprof output dir> <K>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
K = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "k_step<" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grid = lambda r: int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)   # noqa: E731


def region(rs, label):
    s = np.array([int(r["Start_Timestamp"]) for r in rs], dtype=np.int64); e = np.array([int(r["End_Timestamp"]) for r in rs], dtype=np.int64)
    ev = np.concatenate([np.stack([s, np.ones_like(s)], 1), np.stack([e, -np.ones_like(e)], 1)]); ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])[:-1]; dt = np.diff(ev[:, 0]); span = ev[-1, 0] - ev[0, 0]
    share = {k: float(dt[conc == k].sum()) / span for k in (0, 1, 2)}
    q = sorted({r.get("Queue_Id") for r in rs})
    print(f"{label}: {len(rs)} launches of {sorted({grid(r) for r in rs})} threads on queues {q}; launch duration mean {np.mean(e - s) / 1e3:.0f} us; region {span / 1e6:.2f} ms = "
          f"{span / 1e3 / (len(rs) / (2 if len(q) > 1 else 1)):.0f} us per control step of all envs; time with 0 / 1 / 2 control-step kernels in flight: "
          f"{share[0]:.1%} / {share[1]:.1%} / {share[2]:.1%}")


db = rows[-2 * K:]
region(db, "double-buffered")
big = [r for r in rows[:-2 * K] if grid(r) == max(grid(x) for x in rows)]
region(big[-K:], "lockstep")
