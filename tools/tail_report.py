"""What sets the length of a lockstep launch: per-env wavefront durations of ONE control step of the bench workload (4096 Lift envs, all at
episode step `nskip`), their percentiles, the event counts of the slowest envs, and the per-phase profile of the slowest / p99 / median env.
With a -DRSIM_MPRSTAT build (tools/subprof.sh mpr) slots x0..x7 say how the MPR runs end.
Usage (GPU box): [RSIM_LIB=...] python tools/tail_report.py [nskip=200] [B=4096] [task=lift|stack|peg]   (slots per launch: 8 / 8 / 5 resident envs per CU x 256 CUs)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf, peg_in_hole, stack
nskip = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
TASK = sys.argv[3] if len(sys.argv) > 3 else "lift"
STEM, CLS, SLOTS = {"lift": ("lift_panda", lift.LiftBatch, 2048), "stack": ("stack_panda", stack.StackBatch, 2048), "peg": ("peg_baxter_joint_velocity", peg_in_hole.PegBatch, 1280)}[TASK]
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, STEM + ".rsim")); cfg = json.load(open(os.path.join(adir, STEM + ".cfg.json")))
MPR_SLOTS = ("x0 exit pre-test", "x1 exit 1st support", "x2 exit 2nd support", "x3 exit portal discovery", "x4 exit refinement", "x5 contact", "x6 supports of contacts", "x7 supports of late exits", "x8 exit warm start", "x9 portal warm start valid")

tape = None


def run(filter_env):
    env = CLS(flat, cfg, np.arange(B), seed0=0)
    global tape
    if tape is None:
        tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1, action_dim=env.model.action_dim), device="cuda")
    for t in range(nskip): env.step(tape[t])
    env.batch.sync(); env.batch.profile(True); env.batch.profile_env(filter_env)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.ExternalStream(env.batch.stream())
    e0.record(s); env.step(tape[nskip]); e1.record(s); env.batch.sync(); torch.cuda.synchronize()
    return env.batch.wavelog(), env.batch.profile(False), e0.elapsed_time(e1)


w, p, ms = run(-2)   # -2: no env adds to the shared accumulators (undistorted durations)
dur = (w[:, 3].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0
cnt = w[:, 4:8].astype(np.int64)
order = np.argsort(-dur)
print(f"step {nskip}, B={B}: launch {ms*1e3:.0f} us; env wavefront duration us: mean {dur.mean():.0f} p50 {np.percentile(dur,50):.0f} p90 {np.percentile(dur,90):.0f} "
      f"p99 {np.percentile(dur,99):.0f} p99.9 {np.percentile(dur,99.9):.0f} max {dur.max():.0f}; sum/{SLOTS} slots = {dur.sum()/SLOTS:.0f} us")
# the launch as a schedule: 4096 envs on 2048 resident slots (8 per CU), dispatched in decreasing cost of the previous step
t0, t1 = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
st, en, span = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0, (t1.max() - t0.min()) / 100.0
ev = np.concatenate([np.stack([st, np.ones(B)], 1), np.stack([en, -np.ones(B)], 1)]); ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
at = lambda t: int(conc[max(0, np.searchsorted(ev[:, 0], t, side="right") - 1)])
print(f"schedule: span {span:.0f} us; resident envs at 10..100 % of it:", [at(span * k / 10 - 1e-3) for k in range(1, 11)],
      f"; start times us p50 {np.percentile(st, 50):.0f} p75 {np.percentile(st, 75):.0f} p90 {np.percentile(st, 90):.0f} p99 {np.percentile(st, 99):.0f} last {st.max():.0f}"
      f"; envs started in the first 100 us {int((st < 100).sum())} (task {TASK}); mean duration of envs started then {dur[st < 100].mean():.0f}, of the others {dur[st >= 100].mean():.0f} us")
late = np.argsort(-en)[:6]
print("last envs to finish: [env, start us, duration us, dispatch rank]", [[int(e), int(st[e]), int(dur[e]), int(np.argsort(np.argsort(st, kind='stable'), kind='stable')[e])] for e in late])
print("per launch (25 substeps): n_mpr n_support n_newton n_cand -- mean", cnt.mean(0).round(1).tolist(), "p99", np.percentile(cnt, 99, axis=0).round(0).tolist(), "max", cnt.max(0).tolist())
print("corr(dur, n_mpr) %.3f  corr(dur, n_support) %.3f  corr(dur, n_newton) %.3f" % tuple(np.corrcoef(dur, cnt[:, k])[0, 1] for k in (0, 1, 2)))
A = np.stack([np.ones(B), cnt[:, 0], cnt[:, 1], cnt[:, 2], cnt[:, 3]], 1).astype(np.float64)
coef = np.linalg.lstsq(A, dur, rcond=None)[0]
print("least squares: dur_us = %.0f + %.2f n_mpr + %.2f n_support + %.2f n_newton + %.2f n_cand" % tuple(coef))
print("slowest envs: dur_us [n_mpr n_support n_newton n_cand]")
for i in order[:10]: print(f"  env {i}: {dur[i]:.0f}  {cnt[i].tolist()}")
_, pa, _ = run(-1)
ns = max(1, pa["n_sub"])
print("whole batch, per env-substep:", {k[2:]: round(pa[k] / ns, 3) for k in pa if k.startswith("n_") and k != "n_sub"})
xs = {MPR_SLOTS[i]: round(pa[f"x{i}"] / ns, 3) for i in range(9) if pa[f"x{i}"]}
if xs: print("whole batch MPR outcomes per env-substep:", xs)
for which, e in (("slowest", int(order[0])), ("p99", int(order[B // 100])), ("median", int(order[B // 2]))):
    w2, p, _ = run(e)
    nsub = max(1, p["n_sub"])
    cyc = {k: int(v / nsub) for k, v in p.items() if not k.startswith("n_") and not k.startswith("x") and k not in ("boxbox", "mpr", "plane") and v}
    print(f"{which} env {e} ({dur[e]:.0f} us): ticks/substep total {sum(cyc.values())}:", cyc)
    print("    narrow split: boxbox %d mpr %d other %d | per substep:" % (p["boxbox"] / nsub, p["mpr"] / nsub, p["plane"] / nsub),
          {k[2:]: round(p[k] / nsub, 2) for k in p if k.startswith("n_") and k != "n_sub"})
    xs = {MPR_SLOTS[i]: round(p[f"x{i}"] / nsub, 2) for i in range(9) if p[f"x{i}"]}
    if xs and os.environ.get("RSIM_XSLOTS") != "ticks": print("    MPR outcomes per substep:", xs)
    if os.environ.get("RSIM_XSLOTS") == "ticks": print("    sub-phase ticks/substep (x0..x9 of a -DRSIM_SUBPROF build):", {f"x{i}": int(p[f"x{i}"] / nsub) for i in range(10) if p[f"x{i}"]})
