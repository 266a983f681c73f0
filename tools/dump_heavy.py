"""Save the states of the slowest envs of a launch for CPU-side analysis (which pairs reach the narrow phase)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B = 4096; nskip = int(sys.argv[1]) if len(sys.argv) > 1 else 150
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1), device="cuda")
env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
for t in range(nskip): env.step(tape[t])
q0, v0 = env.batch.get("qpos"), env.batch.get("qvel")
env.batch.sync(); env.batch.profile(True); env.batch.profile_env(0); env.step(tape[nskip]); env.batch.sync()
w = env.batch.wavelog()
dur = (w[:, 3].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0
order = np.argsort(-dur)
np.savez(os.path.join(ROOT, "gpurun_out", "heavy_states.npz"), qpos=q0, qvel=v0, dur=dur, counts=w[:, 4:8].astype(np.int64), sizes=env.sizes, order=order)
print("saved; slowest", order[:5], dur[order[:5]])
