"""PickPlace @ 8192 with per-step dynamics randomisation: distribution of the solver-metric gap (objective at the kernel's acceleration above the
oracle's minimum on the kernel's own contact geometry, tests/test_full_size_parity.py) over N envs spread across the batch.  The maximum over a
sample is set by one or two ill-conditioned envs and differs run to run; A/B two builds on percentiles.
Usage (GPU box): [RSIM_LIB=...] python tools/pp_gap_stats.py [N=192] [steps=50]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, pick_place
from tests.util import load_golden
from tests.test_full_size_parity import compare_reached_states, spread
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
B = 8192
ids = np.arange(B)
env = pick_place.PickPlaceBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
b = env.batch
b.dr_save_defaults()
tape = torch.tensor(lift.env_actions(ids, T), device="cuda")
for t in range(T):
    b.randomize_dynamics(seed=11, step=t)
    env.step(tape[t])
grip = {i for i in range(flat.ngeom) if (flat.names["geom"][i] or "").startswith("gripper0_")}
arm, fing = np.asarray(cfg["dof_idx"]), np.asarray(cfg["grip_dof_idx"])
groups = {"arm": arm, "gripper": fing, "objects": np.setdiff1d(np.arange(flat.nv), np.concatenate([arm, fing]))}
res = compare_reached_states(flat, b, spread(B, N), ignore_pair=lambda g1, g2: g1 in grip and g2 in grip, dof_groups=groups)
fed = [r for r in res if r["same"] and "g_cost_gap" in r]
gap = np.array([r["g_cost_gap"] for r in fed])
pc = lambda x: " ".join(f"{v:.1e}" for v in np.percentile(x, [50, 90, 99, 100]))   # noqa: E731
print(f"{os.environ.get('RSIM_LIB', 'librsim_hip.so')}: {len(res)} envs, {len(fed)} agree in structure; diverged {int(b.get('diverged').sum())} of {B}")
print("  objective above its minimum, relative (p50 p90 p99 max):", pc(gap), f"; envs above 1e-6: {int((gap > 1e-6).sum())}, above 1e-4: {int((gap > 1e-4).sum())}")
print("  rel dforce (p50 p90 p99 max):", pc([r["g_force"] / max(1.0, r["g_fscale"]) for r in fed]))
for k in groups:
    print(f"  rel dqacc {k} (p50 p90 p99 max):", pc([r["g_groups"][k][0] / max(1.0, r["g_groups"][k][1]) for r in fed]))
print("  Newton iterations of forward() on these envs: mean %.2f max %d" % (b.get("niter")[[r["env"] for r in fed]].mean(), b.get("niter")[[r["env"] for r in fed]].max()))
