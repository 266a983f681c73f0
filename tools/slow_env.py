"""Per-phase cycle profile of the slowest env of a launch (two identical runs: find it, then profile only it)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B = 4096; nskip = int(sys.argv[1]) if len(sys.argv) > 1 else 150
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1), device="cuda")
def run(filter_env):
    env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
    for t in range(nskip): env.step(tape[t])
    env.batch.sync(); env.batch.profile(True); env.batch.profile_env(filter_env); env.step(tape[nskip]); env.batch.sync()
    w = env.batch.wavelog(); p = env.batch.profile(False)
    return w, p
w, _ = run(-1)
dur = (w[:, 3].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0
for which, e in (("slowest", int(np.argmax(dur))), ("median", int(np.argsort(dur)[B // 2])), ("fastest", int(np.argmin(dur)))):
    w2, p = run(e)
    print(f"{which} env {e}: {dur[e]:.0f} us in run 1; counts {w2[e, 4:8].tolist()}")
    nsub = max(1, p["n_sub"])
    print("   cycles/substep:", {k: int(v / nsub) for k, v in p.items() if not k.startswith("n_") and not k.startswith("x")})
    print("   sub-phase slots:", {k: int(v / nsub) for k, v in p.items() if k.startswith("x") and v})
    print("   counts/substep:", {k: round(v / nsub, 2) for k, v in p.items() if k.startswith("n_")})
