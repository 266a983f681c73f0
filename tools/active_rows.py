"""How many constraint rows of a reached state carry a Hessian weight (quadratic or cone state: force != 0) -- the rows H = M + J^T W J actually needs.
Usage (GPU box): python tools/active_rows.py [pickplace|stack] [B] [steps]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, pick_place, stack
from tests.util import load_golden
task = sys.argv[1] if len(sys.argv) > 1 else "pickplace"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 60
if task == "pickplace":
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    env = pick_place.PickPlaceBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
    env.batch.dr_save_defaults()
else:
    g, cfg, flat = load_golden("seed0_full", "stack_panda")
    env = stack.StackBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2)
b = env.batch
tape = torch.tensor(lift.env_actions(np.arange(B), T), device="cuda")
for t in range(T):
    if task == "pickplace": b.randomize_dynamics(seed=11, step=t)
    env.step(tape[t])
b.forward()
efc, nefc, niter = b.get("efc_force"), b.get("nefc"), b.get("niter")
act = np.array([(efc[e][:nefc[e]] != 0).sum() for e in range(B)])
ch_all = (nefc + 3) // 4
ch_act = (act + 3) // 4
print(f"{task} B={B} after {T} steps: rows mean {nefc.mean():.1f} p90 {np.percentile(nefc, 90):.0f} max {nefc.max()};  rows with force != 0: mean {act.mean():.1f} p90 {np.percentile(act, 90):.0f} max {act.max()}")
print(f"   four-row chunks: all {ch_all.mean():.2f}  compacted {ch_act.mean():.2f}  ({100 * (1 - ch_act.sum() / ch_all.sum()):.0f} % fewer);  weighted by Newton iterations: {100 * (1 - (ch_act * np.maximum(niter, 1)).sum() / (ch_all * np.maximum(niter, 1)).sum()):.0f} % fewer")
heavy = np.argsort(-niter)[: max(1, B // 20)]
print(f"   the 5 % of envs with most Newton iterations: rows {nefc[heavy].mean():.1f} active {act[heavy].mean():.1f}  chunks {ch_all[heavy].mean():.2f} -> {ch_act[heavy].mean():.2f}")
