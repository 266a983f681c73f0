"""Throughput of the other BASELINE configurations through the vectorised facade (random full-range actions, on-device episode restart):
Stack/Panda (configs[2] model), TwoArmPegInHole/Baxter JOINT_VELOCITY-class joint controllers (configs[3] model), PickPlace/IIWA+Robotiq140
(configs[4] model).  Not the bench metric (bench.py = configs[1]); a tuning reference for kernel configurations 1-3.
Usage (GPU box): python tools/bench_configs.py [steps]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd.vec_env import VecEnv
from tests.util import load_golden

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = sys.argv[2:]   # optional task names
out = {}
for name, tag, model, B in (("Stack", "seed0_full", "stack_panda", 4096), ("TwoArmPegInHole", "ctl_joint_velocity", "peg_baxter", 2048),
                            ("PickPlace", "seed0_full", "pickplace_iiwa", 2048)):
    if only and name not in only:
        continue
    g, cfg, flat = load_golden(tag, model)
    env = VecEnv(name, B, flat, cfg, seed=0, horizon=500, bank_episodes=2, stream_groups=int(os.environ.get("RSIM_GROUPS", 8)))
    env.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = [torch.rand(B, env.action_dim, device="cuda", generator=gen) * 2 - 1 for _ in range(steps + 5)]
    for t in range(5): env.step(acts[t])
    torch.cuda.synchronize(); env.env.batch.sync(); t0 = time.perf_counter()
    for t in range(5, 5 + steps): env.step(acts[t])
    env.env.batch.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[name] = {"envs": B, "kernel_config": int(env.env.model.kernel_config()[0]), "ms_per_step": 1e3 * dt / steps, "env_steps_per_s": B * steps / dt}
    print(name, out[name], flush=True)
    del env
if not only or "PickPlaceDR" in only:
    # BASELINE configs[4] as stated: 8192 envs, dynamics randomisation re-drawn before every control step (per-env constant blocks rebuilt each time)
    from robosuite_amd import pick_place
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    B = 8192
    env = pick_place.PickPlaceBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
    b = env.batch
    b.dr_save_defaults()
    b.set_stream_groups(int(os.environ.get("RSIM_GROUPS", 8)))
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    acts = [torch.rand(B, env.model.action_dim, device="cuda", generator=gen) * 2 - 1 for _ in range(steps + 5)]
    for t in range(5): b.randomize_dynamics(seed=11, step=t); env.step(acts[t])
    b.sync(); t0 = time.perf_counter()
    for t in range(5, 5 + steps): b.randomize_dynamics(seed=11, step=t); env.step(acts[t])
    b.sync(); dt = time.perf_counter() - t0
    out["PickPlaceDR"] = {"envs": B, "kernel_config": int(env.model.kernel_config()[0]), "ms_per_step": 1e3 * dt / steps, "env_steps_per_s": B * steps / dt,
                          "diverged": int(b.get("diverged").sum()), "overflow_envs": int((b.get("overflow") > 0).sum())}
    print("PickPlaceDR", out["PickPlaceDR"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
