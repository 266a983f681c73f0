"""Where do the env wavefronts run?  Logs {HW_ID, XCC_ID, start, end} per env for one control step and reports residency."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0, per_env_cube=False)
a = torch.zeros(B, 7, device="cuda").uniform_(-1, 1)
nskip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1), device="cuda")
for t in range(nskip): env.step(tape[t])
a = tape[nskip]
env.batch.sync(); env.batch.profile(True); env.batch.profile_env(0); env.step(a); env.batch.sync()
w = env.batch.wavelog()
hw, xcc, t0, t1 = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64) & 0xF, w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key_cu = ((xcc * 8 + se) * 2 + sh) * 16 + cu
tmin = t0.min(); dur = (t1 - t0) / 100.0
print(f"B={B}: kernel span {(t1.max()-tmin)/100.0:.1f} us; wave duration mean {dur.mean():.1f} us min {dur.min():.1f} max {dur.max():.1f}")
print("distinct XCC", len(np.unique(xcc)), "distinct CUs", len(np.unique(key_cu)), "distinct (CU,SIMD)", len(np.unique(key_cu * 4 + simd)))
# concurrency over time
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)]); ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1]); print("max concurrent waves", conc.max(), "time-avg", float((conc[:-1] * np.diff(ev[:, 0])).sum() / (ev[-1, 0] - ev[0, 0])))
# per-CU max concurrency and per-SIMD
for name, key in (("CU", key_cu), ("SIMD", key_cu * 4 + simd)):
    mx = []
    for k in np.unique(key)[:64]:
        m = key == k
        e = np.concatenate([np.stack([t0[m], np.ones(m.sum(), np.int64)], 1), np.stack([t1[m], -np.ones(m.sum(), np.int64)], 1)]); e = e[np.argsort(e[:, 0], kind="stable")]
        mx.append(np.cumsum(e[:, 1]).max())
    print(f"per-{name} max concurrent waves (first 64): min {min(mx)} max {max(mx)} mean {np.mean(mx):.2f}")
print("waves per SIMD id:", np.bincount(simd, minlength=4))
st = (t0 - tmin) / 100.0
print("start time percentiles (us):", np.percentile(st, [0, 10, 25, 50, 75, 90, 100]).round(1))
print("duration percentiles (us):", np.percentile(dur, [0, 10, 25, 50, 75, 90, 100]).round(1))
k0 = np.unique(key_cu)[0]; mm = key_cu == k0
print("one CU:", sorted([(int(simd[i]), round(float(st[i]), 1), round(float(dur[i]), 1)) for i in np.nonzero(mm)[0]], key=lambda x: x[1])[:12])

cnt = w[:, 4:8].astype(np.int64)
order = np.argsort(-dur)
print("slowest envs: dur_us n_mpr n_support n_newton n_cand (per launch of 25 substeps)")
for i in order[:8]: print(f"  env {i}: {dur[i]:.0f} us  {cnt[i].tolist()}")
print("fastest:"); 
for i in order[-3:]: print(f"  env {i}: {dur[i]:.0f} us  {cnt[i].tolist()}")
print("corr(dur, n_support) =", np.corrcoef(dur, cnt[:, 1])[0, 1].round(3), " corr(dur, n_newton) =", np.corrcoef(dur, cnt[:, 2])[0, 1].round(3))
print("mean counts per launch:", cnt.mean(0).round(1).tolist(), " p99 of n_support:", np.percentile(cnt[:, 1], 99))
