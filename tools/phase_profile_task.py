"""Per-phase cycle shares of the fused kernel for one env of another task (Stack / TwoArmPegInHole / PickPlace), plus event counts per substep.
Usage (GPU box): python tools/phase_profile_task.py PickPlace [B] [skip] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd.vec_env import VecEnv
from tests.util import load_golden
name = sys.argv[1] if len(sys.argv) > 1 else "PickPlace"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 20
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
tag, model = {"Stack": ("seed0_full", "stack_panda"), "TwoArmPegInHole": ("ctl_joint_velocity", "peg_baxter"), "PickPlace": ("seed0_full", "pickplace_iiwa")}[name]
g, cfg, flat = load_golden(tag, model)
env = VecEnv(name, B, flat, cfg, seed=0, horizon=500, bank_episodes=2)
env.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
acts = [torch.rand(B, env.action_dim, device="cuda", generator=gen) * 2 - 1 for _ in range(skip + steps)]
for t in range(skip): env.step(acts[t])
b = env.env.batch
b.sync(); t0 = time.perf_counter()
for t in range(steps): env.step(acts[skip + t])
b.sync(); dt = time.perf_counter() - t0
print(f"{name} B={B}: {1e3 * dt / steps:.2f} ms/step -> {B * steps / dt:.0f} env-steps/s (unprofiled)")
for e in (0, B // 2):
    b.profile(True); b.profile_env(e)
    for t in range(2): env.step(acts[skip + t])
    b.sync(); w = b.wavelog(); p = b.profile(False)
    nsub = max(1, p["n_sub"])
    cyc = {k: v for k, v in p.items() if not k.startswith("n_") and k not in ("boxbox", "mpr", "plane") and not k.startswith("x")}
    if os.environ.get("RSIM_SUBPROF_RAW"):   # RSIM_SUBPROF=3 build: x0 composite | x1 mass loops | x2 copy | x3 factor (crb slot: what is left) ; x4 cvel+cvb | x5 cdd+cacc | x6 body wrench | x7 subtree sums (vel slot: tendons, bias)
        tot0 = sum(cyc.values()) + sum(v for k, v in p.items() if k.startswith("x"))
        print("   raw slots per substep (same units): " + "  ".join(f"{k} {v / nsub:.0f}" for k, v in p.items() if (k.startswith("x") and v) or k in ("crb", "vel", "com", "kin", "makec", "broad", "euler", "solve")) + f"  | total {tot0 / nsub:.0f}")
    xs = {k: v for k, v in p.items() if k.startswith("x") and v}
    if xs:   # -DRSIM_SUBPROF build: x0 rows x1 warm start x2 evaluate + J^T f + gradient x3 Hessian weights x4 H, factor, solve x5 line-search setup x6 line search (x7-x9: OSC)
        print("   solver / controller sub-phases (share of the solve phase): " + "  ".join(f"{k} {100 * v / max(1, p['solve']):.1f}%" for k, v in xs.items()))
    tot = sum(cyc.values())
    print(f"env {e}: " + "  ".join(f"{k} {100 * v / tot:.1f}%" for k, v in cyc.items() if v))
    print("   inside narrow: " + "  ".join(f"{k} {100 * p[k] / tot:.1f}%" for k in ("boxbox", "mpr", "plane")))
    print("   per substep: " + "  ".join(f"{k} {p[k] / nsub:.2f}" for k in ("n_cand", "n_con", "n_efc", "n_newton", "n_ls", "n_boxbox", "n_mpr", "n_support")))
dur = (w[:, 3].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0
print("wave duration us: min %.0f median %.0f p90 %.0f max %.0f" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
