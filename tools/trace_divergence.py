"""Find an env of the PickPlace random rollout that hits the bad-state guard and trace it against the fp64 oracle (GPU box).
1. rollout (B, seed as the edge-case test) recording every env's state before each control step; first (env, step) whose RSIM_DIVERGED rises;
2. oracle replay of that env from its reset with the same action stream: per-control-step deviation up to the failing step;
3. from the last control-step boundary where they agree: substep-by-substep comparison, both sides restarted from the kernel's state,
   controller torques from the oracle controller (teacher forcing), forward quantities compared at every substep."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd.vec_env import VecEnv
from robosuite_amd import pick_place
from tests.util import load_golden, make_hip, make_oracle
B, seed = int(os.environ.get("B", 128)), int(os.environ.get("SEED", 5))
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
nq, nv = flat.nq, flat.nv
env = VecEnv("PickPlace", B, flat, cfg, seed=0, horizon=100, bank_episodes=2); env.reset()
b = env.env.batch
gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
acts, snaps, hit = [], [], None
for t in range(150):
    a = torch.rand(B, env.action_dim, device="cuda", generator=gen) * 2 - 1
    snaps.append({k: b.get(k).copy() for k in ("qpos", "qvel", "qacc_warmstart", "ctrl", "cstate", "ep_step", "ep_index")})
    acts.append(a.cpu().numpy())
    env.step(a)
    d = b.get("diverged")
    if d.sum() > 0:
        hit = (int(np.nonzero(d)[0][0]), t); break
if hit is None:
    print("no env diverged"); sys.exit(0)
e, t = hit
ep0 = t - int(snaps[t]["ep_step"][e])     # first control step of the env's running episode
print(f"env {e} diverges during control step {t} (episode {int(snaps[t]['ep_index'][e])}, episode step {int(snaps[t]['ep_step'][e])})")
# ---- 2. oracle replay of the episode
om, od, oc = make_oracle(flat, cfg)
od.qpos[:] = snaps[ep0]["qpos"][e]; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward(); oc.reset(od)
last_ok = ep0
for s in range(ep0, t):
    oc.env_step(od, acts[s][e].astype(np.float64), 25)
    hq, hv = snaps[s + 1]["qpos"][e], snaps[s + 1]["qvel"][e]
    dq, dv = np.abs(hq - od.qpos), np.abs(hv - od.qvel)
    print(f"  after step {s}: |dq| arm {dq[:7].max():.1e} fingers {dq[7:13].max():.1e} objects {dq[13:].max():.1e}  |dv| max {dv.max():.1e}  oracle ncon {od.ncon} nefc {od.nefc} |qvel|max {np.abs(od.qvel).max():.1f}")
    if dq.max() < 2e-2: last_ok = s + 1
# ---- 3. the control step in which the kernel's state explodes, one substep at a time with identical inputs on both sides:
# kernel state after k substeps (control_step(a, k) from the snapshot: true controller semantics, set_goal once) -> oracle.step() from that very
# state with the ctrl the kernel applied in substep k + 1 -> compare with the kernel's state after k + 1 substeps
tb = next(s for s in range(ep0, t + 1) if not (np.abs(snaps[s + 1]["qvel"][e]).max() < 100.0)) if any(not (np.abs(snaps[s + 1]["qvel"][e]).max() < 100.0) for s in range(ep0, t)) else t
print(f"kernel |qvel| first exceeds 100 during control step {tb}")
hm, hb1 = make_hip(flat, cfg, B=1)
st = snaps[tb]
a1 = torch.tensor(acts[tb][e][None], device="cuda")
def after(k):
    for f in ("qpos", "qvel", "qacc_warmstart", "ctrl", "cstate"): hb1.set(f, st[f][e][None])
    if k: hb1.control_step(a1, k)
    return {f: hb1.get(f)[0].astype(np.float64).copy() for f in ("qpos", "qvel", "qacc_warmstart", "ctrl")}
om2, od2, _ = make_oracle(flat, cfg)
prev = after(0)
for k in range(25):
    cur = after(k + 1)
    od2.qpos[:] = prev["qpos"]; od2.qvel[:] = prev["qvel"]; od2.qacc_warmstart[:] = prev["qacc_warmstart"]; od2.ctrl[:] = cur["ctrl"]
    od2.step()
    dq, dv = np.abs(cur["qpos"] - od2.qpos), np.abs(cur["qvel"] - od2.qvel)
    print(f"  substep {k:2d}: oracle ncon {od2.ncon} nefc {od2.nefc} iter {od2.solver_iter} |dq| {dq.max():.1e} |dv| {dv.max():.1e} (dof {int(dv.argmax())}) |v| kernel {np.abs(cur['qvel']).max():.1e} oracle {np.abs(od2.qvel).max():.1e}")
    if dv.max() > 1.0 or k == 24:
        # forward quantities at the state the bad substep started from
        for f in ("qpos", "qvel", "qacc_warmstart"): hb1.set(f, prev[f][None])
        hb1.set("ctrl", cur["ctrl"][None]); hb1.forward()
        od2.qpos[:] = prev["qpos"]; od2.qvel[:] = prev["qvel"]; od2.qacc_warmstart[:] = prev["qacc_warmstart"]; od2.ctrl[:] = cur["ctrl"]; od2.forward()
        print(f"    forward at that state: ncon {int(hb1.get('ncon')[0])}/{od2.ncon} nefc {int(hb1.get('nefc')[0])}/{od2.nefc} niter {int(hb1.get('niter')[0])}/{od2.solver_iter}")
        da = np.abs(hb1.get("qacc")[0] - od2.qacc)
        print("    |dacc| per dof", np.round(da, 1).tolist()); print("    oracle qacc   ", np.round(np.asarray(od2.qacc), 1).tolist())
        for a_, b_ in zip(hb1.contacts(0), od2.contacts()):
            ang = float(np.degrees(np.arccos(np.clip(np.dot(a_["frame"][0], b_["frame"][0]), -1, 1))))
            print(f"    contact {flat.names['geom'][b_['geom1']]} / {flat.names['geom'][b_['geom2']]}: dist {a_['dist']:.6f} / {b_['dist']:.6f}  normal off {ang:.2f} deg  fn {a_['normal_force']:.2f} / {b_['normal_force']:.2f}")
        print("    bias diff", float(np.abs(hb1.get("qfrc_bias")[0] - od2.qfrc_bias).max()), "passive diff", float(np.abs(hb1.get("qfrc_passive")[0] - od2.qfrc_passive).max()), "M rel diff", float(np.abs(hb1.get("qM")[0].ravel() - od2.qM).max() / np.abs(od2.qM).max()))
        if dv.max() > 1.0: break
    prev = cur
