"""A/B of two library builds on another task in ONE GPU session: throughput of each and whether the states they reach are identical.
Usage (GPU box): python tools/ab_task.py PickPlace 1024 30 lib_a.so lib_b.so   (library paths relative to the repo root; each runs in its own process)"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch
    from robosuite_amd.vec_env import VecEnv
    from tests.util import load_golden
    name, B, steps, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    tag, model = {"Stack": ("seed0_full", "stack_panda"), "TwoArmPegInHole": ("ctl_joint_velocity", "peg_baxter"), "PickPlace": ("seed0_full", "pickplace_iiwa"),
                  "Lift": ("seed0_full", "lift_panda")}[name]
    g, cfg, flat = load_golden(tag, model)
    env = VecEnv(name, B, flat, cfg, seed=0, horizon=500, bank_episodes=2)
    env.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = [torch.rand(B, env.action_dim, device="cuda", generator=gen) * 2 - 1 for _ in range(steps)]
    b = env.env.batch
    warm = min(10, steps // 3)
    for t in range(warm): env.step(acts[t])
    b.sync(); t0 = time.perf_counter()
    for t in range(warm, steps): env.step(acts[t])
    b.sync(); dt = time.perf_counter() - t0
    np.savez(out, qpos=b.get("qpos"), qvel=b.get("qvel"), overflow=b.get("overflow"), diverged=b.get("diverged"), nefc=b.get("nefc"))
    print(f"{os.environ.get('RSIM_LIB', 'default')}: {name} B={B}: {1e3 * dt / (steps - warm):.2f} ms/step -> {B * (steps - warm) / dt:.0f} env-steps/s", flush=True)
    sys.exit(0)
name, B, steps = sys.argv[1], sys.argv[2], sys.argv[3]
outs = []
for rep in range(2):
    for i, lib in enumerate(sys.argv[4:]):
        out = f"/tmp/ab_task_{i}.npz"
        subprocess.run([sys.executable, __file__, "--worker", name, B, steps, out], env=dict(os.environ, RSIM_LIB=os.path.join(ROOT, lib)), check=True)
        if rep == 0: outs.append(out)
ref = np.load(outs[0])
for i, o in enumerate(outs[1:], 1):
    d = np.load(o)
    pe = np.abs(d["qpos"].astype(np.float64) - ref["qpos"].astype(np.float64)).reshape(len(ref["qpos"]), -1).max(axis=1)
    ne = ref["nefc"].ravel()
    print(f"per-env max |dqpos|: identical {int((pe == 0).sum())}/{len(pe)}  median {np.median(pe):.2e}  p90 {np.percentile(pe, 90):.2e}  p99 {np.percentile(pe, 99):.2e}  max {pe.max():.2e}"
          f" | nefc max {ne.max()}  envs with nefc > 64: {int((ne > 64).sum())}, of those identical: {int(((ne > 64) & (pe == 0)).sum())}")
    print(f"lib {i} vs lib 0: " + "  ".join(f"{k} max|d| {np.abs(d[k].astype(np.float64) - ref[k].astype(np.float64)).max():.3e} ({'identical' if np.array_equal(d[k], ref[k]) else 'differs'})" for k in ref.files))
