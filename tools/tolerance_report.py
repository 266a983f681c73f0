"""Measured error levels behind the looser assertions of tests/test_hip_parity.py (run on the GPU box; prints maxima, asserts nothing)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import load_golden, make_hip, make_oracle
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-12, np.abs(np.asarray(b)).max()))

# ---- PickPlace fused control step vs oracle loop
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
nq = flat.nq
om, od, oc = make_oracle(flat, cfg); hm, hb = make_hip(flat, cfg, B=2)
s0 = g["states"][0]
od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward(); oc.reset(od)
hb.set("qpos", s0[1:1 + nq][None].repeat(2, 0)); hb.set("qvel", s0[1 + nq:][None].repeat(2, 0)); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0); hb.forward(); hb.ctrl_reset()
fingers = np.zeros(nq, dtype=bool); fingers[7:13] = True; arm = np.zeros(nq, dtype=bool); arm[:7] = True
w = dict(arm=0.0, fingers=0.0, objects=0.0, ctrl=0.0, fv=0.0)
for t in range(len(g["actions"])):
    hb.control_step(torch.tensor(np.repeat(g["actions"][t][None], 2, 0), dtype=torch.float32, device="cuda"), 25); oc.env_step(od, g["actions"][t], 25)
    dq = np.abs(hb.get("qpos")[0] - od.qpos)
    w["arm"] = max(w["arm"], dq[arm].max()); w["fingers"] = max(w["fingers"], dq[fingers].max()); w["objects"] = max(w["objects"], dq[~(arm | fingers)].max())
    w["ctrl"] = max(w["ctrl"], rel(hb.get("ctrl")[0], od.ctrl)); w["fv"] = max(w["fv"], np.abs(hb.get("qvel")[0][7:13] - od.qvel[7:13]).max())
print("PickPlace 20 control steps vs oracle loop: max |dq|", {k: f"{v:.2e}" for k, v in w.items()})

# ---- Baxter random configurations: MPR depth agreement
g, cfg, flat = load_golden("ctl_joint_torque", "peg_baxter")
om, od, _ = make_oracle(flat); hm, hb = make_hip(flat, None, B=2)
rng = np.random.default_rng(7); dd, da, n = [], [], 0
for it in range(40):
    q = flat.qpos0.copy()
    for j in range(flat.njnt):
        lo, hi = flat.jnt_range[j]; q[flat.jnt_qposadr[j]] = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo))
    v = 0.5 * rng.standard_normal(flat.nv)
    od.qpos[:] = q; od.qvel[:] = v; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward()
    if od.ncon == 0 or od.ncon >= hb.maxcon or od.nefc >= hb.maxefc: continue
    hb.set("qpos", q[None].repeat(2, 0)); hb.set("qvel", v[None].repeat(2, 0)); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0); hb.forward()
    if hb.get("ncon")[0] != od.ncon: print("ncon differs at", it); continue
    for a, b in zip(hb.contacts(0), od.contacts()):
        dd.append(abs(a["dist"] - b["dist"]) / max(1e-9, abs(b["dist"]))); n += 1
    da.append(np.abs(hb.get("qacc")[0] - od.qacc).max() / max(1.0, np.abs(od.qacc).max()))
dd = np.array(dd)
print(f"Baxter random poses: {n} contacts; rel depth error: median {np.median(dd):.1e} p90 {np.percentile(dd, 90):.1e} max {dd.max():.1e}; rel qacc error with contacts: max {max(da):.1e} median {np.median(da):.1e}")

# ---- PickPlace long random rollout: bad-state guard
from robosuite_amd.vec_env import VecEnv
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
for B, seed in ((128, 5), (1024, 6)):
    env = VecEnv("PickPlace", B, flat, cfg, seed=0, horizon=100, bank_episodes=2); env.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    for t in range(150): env.step(torch.rand(B, env.action_dim, device="cuda", generator=gen) * 2 - 1)
    print(f"PickPlace B={B} 150 random steps: diverged envs {int((env.env.batch.get('diverged') > 0).sum())}, overflow envs {int((env.env.batch.get('overflow') > 0).sum())}")
