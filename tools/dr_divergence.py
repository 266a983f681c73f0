"""PickPlace with a dynamics-randomisation draw before every control step: which envs hit the bad-state guard, in what state, and with which draw?
(GPU box)  Usage: python tools/dr_divergence.py [B] [steps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import pick_place
from tests.util import load_golden
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 35
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
env = pick_place.PickPlaceBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
b = env.batch
b.dr_save_defaults()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
acts = [torch.rand(B, env.model.action_dim, device="cuda", generator=gen) * 2 - 1 for _ in range(T)]
names = flat.names
import json
DR_OVERRIDE = json.loads(os.environ.get("DR_OVERRIDE", "{}"))   # e.g. {"position_size": 0, "quaternion_size": 0}
prev = {}
seen = np.zeros(B, dtype=bool)
for t in range(T):
    prev = {k: b.get(k).copy() for k in ("qpos", "qvel", "ncon", "nefc", "niter")}
    if not os.environ.get("NODR"): b.randomize_dynamics(seed=11, step=t, **DR_OVERRIDE)
    env.step(acts[t])
    d = b.get("diverged") > 0
    new = np.nonzero(d & ~seen)[0]
    seen |= d
    for e in new[:3]:
        v = prev["qvel"][e]
        print(f"step {t}: env {e} hits the guard; before the step: ncon {int(prev['ncon'][e])} nefc {int(prev['nefc'][e])} newton iters {int(prev['niter'][e])} max|qvel| {np.abs(v).max():.2f} at dof {int(np.abs(v).argmax())}")
        for field in ("body_mass", "body_inertia", "dof_damping", "dof_armature", "dof_frictionloss"):
            cur = b.param_get(field, e, 1)[0].ravel(); base = np.asarray(flat.arrays[field], dtype=np.float64).ravel()
            rel = np.where(np.abs(base) > 0, cur / np.where(base == 0, 1, base), np.nan)
            print(f"     {field}: ratio to default min {np.nanmin(rel):.3f} max {np.nanmax(rel):.3f}; smallest value {cur.min():.3e}")
        q = prev["qpos"][e]
        for o in cfg["task"]["placement"]["objects"]:
            a = o["qposadr"]; print(f"     {o['name']}: pos {np.round(q[a:a+3], 3).tolist()}")
        print(f"     gripper qpos {np.round(q[cfg['grip_qpos_idx']], 3).tolist()} qvel {np.round(v[cfg['grip_dof_idx']], 2).tolist()}")
print("envs that hit the guard:", int(seen.sum()), "of", B, "in", T, "steps")
