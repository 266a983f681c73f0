"""Do the solo envs (rsim_set_solo_envs) get a SIMD to themselves, and how much faster do they run?  One control step of the bench workload at
episode step `nskip`, once without and once with `nsolo` solo envs (same states: the rollout is deterministic); per-env wavefront start / end /
hardware id from the wave log.  Usage (GPU box): python tools/solo_report.py [nskip=200] [nsolo=64]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
nskip = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nsolo = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = 4096
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1), device="cuda")


def run(solo):
    env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
    for t in range(nskip): env.step(tape[t])
    env.batch.sync(); env.batch.profile(True); env.batch.profile_env(-2); env.batch.set_solo_envs(solo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.ExternalStream(env.batch.stream())
    e0.record(s); env.step(tape[nskip]); e1.record(s); env.batch.sync(); torch.cuda.synchronize()
    w = env.batch.wavelog()
    return w, e0.elapsed_time(e1)


def unpack(w):
    hw, xcc, t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64) & 0xF, w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    tmin = t0.min()
    return key, (t0 - tmin) / 100.0, (t1 - tmin) / 100.0


w0, ms0 = run(0)
w1, ms1 = run(nsolo)
k0, s0, e0 = unpack(w0)
k1, s1, e1 = unpack(w1)
d0, d1 = e0 - s0, e1 - s1
heavy = np.argsort(-d0)[:nsolo]          # the envs the dispatch order puts first (previous step's cost ~ this step's)
print(f"step {nskip}: launch {ms0*1e3:.0f} us without solo envs, {ms1*1e3:.0f} us with {nsolo}; last wavefront ends at {e0.max():.0f} / {e1.max():.0f} us")
print(f"the {nsolo} heaviest envs: duration mean {d0[heavy].mean():.0f} -> {d1[heavy].mean():.0f} us (x{(d0[heavy] / d1[heavy]).mean():.2f}), max {d0[heavy].max():.0f} -> {d1[heavy].max():.0f}; "
      f"start mean {s0[heavy].mean():.0f} -> {s1[heavy].mean():.0f} us, latest start {s1[heavy].max():.0f}")
rest = np.setdiff1d(np.arange(B), heavy)
print(f"all other envs: duration mean {d0[rest].mean():.0f} -> {d1[rest].mean():.0f} us; end p99 {np.percentile(e0[rest], 99):.0f} -> {np.percentile(e1[rest], 99):.0f}, max {e0[rest].max():.0f} -> {e1[rest].max():.0f}")
# exclusivity: for each heavy env in the solo run, the time another wavefront spent on its SIMD while it ran
shared = []
for i in heavy:
    m = (k1 == k1[i]) & (np.arange(B) != i)
    ov = np.clip(np.minimum(e1[m], e1[i]) - np.maximum(s1[m], s1[i]), 0, None).sum()
    shared.append(ov / max(1e-9, d1[i]))
shared = np.array(shared)
print(f"fraction of a solo env's run time with another wavefront on its SIMD: mean {shared.mean():.3f}, max {shared.max():.3f}, envs with > 5 %: {(shared > 0.05).sum()}")
sh0 = []
for i in heavy:
    m = (k0 == k0[i]) & (np.arange(B) != i)
    sh0.append(np.clip(np.minimum(e0[m], e0[i]) - np.maximum(s0[m], s0[i]), 0, None).sum() / max(1e-9, d0[i]))
print(f"(without solo envs the same envs share their SIMD {np.mean(sh0):.3f} of the time)")
order = np.argsort(-e1)[:8]
print("last wavefronts to end with solo envs: env end_us start_us dur_us was_heavy", [(int(i), int(e1[i]), int(s1[i]), int(d1[i]), bool(i in set(heavy.tolist()))) for i in order])
