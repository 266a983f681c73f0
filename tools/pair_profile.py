"""Which candidate pairs reach the narrow phase, and what they cost in support calls, late in the episode (all 4096 envs, one launch)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B = 4096; nskip = int(sys.argv[1]) if len(sys.argv) > 1 else 200
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1), device="cuda")
env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
for t in range(nskip): env.step(tape[t])
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # restrict the pair counters to one env (-1: all envs)
env.batch.sync(); env.batch.profile(True); env.batch.profile_env(only); env.step(tape[nskip]); env.batch.sync()
if only >= 0: B = 1
vis, sup = env.batch.pairlog()
g1, g2 = flat.arrays["pair_geom1"], flat.arrays["pair_geom2"]
names = flat.names["geom"]
order = np.argsort(-sup)
print(f"step {nskip}: narrow-phase visits per env-substep {vis.sum()/B/25:.2f}, supports per env-substep {sup.sum()/B/25:.2f}")
print("pair                                                              visits/env-step  supports/env-step  supports/visit")
for p in order[:25]:
    print(f"{names[g1[p]]:38s} {names[g2[p]]:38s} {vis[p]/B:8.2f} {sup[p]/B:10.2f} {sup[p]/max(1,vis[p]):8.1f}")
nz = vis > 0
print("pairs ever visited:", int(nz.sum()), "of", len(vis))
order2 = np.argsort(-vis)
print("most visited:")
for p in order2[:12]:
    print(f"{names[g1[p]]:38s} {names[g2[p]]:38s} {vis[p]/B:8.2f} {sup[p]/B:10.2f}")
