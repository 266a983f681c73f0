"""Exploratory GPU-vs-oracle comparison (prints errors; the asserting versions live in tests/)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_golden, make_oracle, make_hip
from robosuite_amd import backend

def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-12, np.abs(np.asarray(b)).max()))

g, cfg, flat = load_golden("seed1_full")
# 1. OSC eval
kp = np.array(cfg["kp"])
N = 200
idx = np.linspace(0, len(g["tau"]) - 1, N).astype(int)
packed = np.stack([backend.pack_osc_inputs(g["ep"][i], g["eR"][i], g["ev"][i], g["op"][i], g["oR"][i], g["bv"][i], g["goal_pos"][i], g["goal_ori"][i], g["J"][i], g["M"][i], g["bias"][i], g["q"][i], g["qd"][i], g["q0"][i]) for i in idx])
out = backend.osc_eval(cfg, packed)
print("osc_eval abs err", np.abs(out[:, :7] - g["tau"][idx]).max(), "tau max", np.abs(g["tau"][idx]).max())

# 2. forward quantities at recorded substates
om, od, oc = make_oracle(flat, cfg)
hm, hb = make_hip(flat, cfg, B=4)
for i in (0, 30, 400, 999):
    od.qpos[:] = g["sub_qpos"][i]; od.qvel[:] = g["sub_qvel"][i]; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0
    od.forward()
    hb.set("qpos", g["sub_qpos"][i][None].repeat(4, 0)); hb.set("qvel", g["sub_qvel"][i][None].repeat(4, 0)); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0)
    hb.forward()
    print(f"state {i}: xpos {np.abs(hb.get('xpos')[0].ravel()-od.xpos).max():.2e} xquat {np.abs(hb.get('xquat')[0].ravel()-od.xquat).max():.2e} "
          f"M {rel(hb.get('qM')[0].ravel(), od.qM):.2e} bias {rel(hb.get('qfrc_bias')[0], od.qfrc_bias):.2e} passive {np.abs(hb.get('qfrc_passive')[0]-od.qfrc_passive).max():.2e} "
          f"ncon {hb.get('ncon')[0]} {od.ncon} nefc {hb.get('nefc')[0]} {od.nefc} niter {hb.get('niter')[0]} {od.solver_iter} "
          f"qacc {np.abs(hb.get('qacc')[0]-od.qacc).max():.2e} (|qacc| {np.abs(od.qacc).max():.2e}) fc {np.abs(hb.get('qfrc_constraint')[0]-od.qfrc_constraint).max():.2e}")
    hc, ocn = hb.contacts(0), od.contacts()
    for a, b_ in zip(hc, ocn):
        print("   con", a["geom1"], a["geom2"], f"dist {a['dist']:.3e} {b_['dist']:.3e} pos {np.abs(a['pos']-b_['pos']).max():.1e} fn {a['normal_force']:.4f} {b_['normal_force']:.4f}")
    same = all(np.array_equal(hb.get(k)[0], hb.get(k)[3]) for k in ("qacc", "xpos", "qM"))
    print("   envs identical:", same)

# 3. fused control step vs oracle native loop and vs golden (reference env loop)
import torch
s0 = g["states"][0]
nq, nv = flat.nq, flat.nv
od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0; od.forward(); oc.reset(od)
hb.set("qpos", s0[1:1 + nq][None].repeat(4, 0)); hb.set("qvel", s0[1 + nq:][None].repeat(4, 0)); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0)
hb.forward()
hb.set("qacc_warmstart", hb.get("qacc"))
hb.ctrl_reset()
print("cstate q0", hb.get("cstate")[0][12:19], s0[1:8])
for t in range(len(g["actions"])):
    a = torch.tensor(np.repeat(g["actions"][t][None], 4, 0), dtype=torch.float32, device="cuda")
    hb.control_step(a, 25)
    oc.env_step(od, g["actions"][t], 25)
    hq, hv = hb.get("qpos")[0], hb.get("qvel")[0]
    if t % 5 == 0 or t == len(g["actions"]) - 1:
        print(f"step {t}: vs oracle q {np.abs(hq-od.qpos).max():.2e} v {np.abs(hv-od.qvel).max():.2e} | vs golden q {np.abs(hq-g['states'][t+1][1:1+nq]).max():.2e} v {np.abs(hv-g['states'][t+1][1+nq:]).max():.2e}")
# 4. throughput
for B in (1024, 4096):
    hm2, hb2 = make_hip(flat, cfg, B=B)
    hb2.set("qpos", s0[1:1 + nq][None].repeat(B, 0)); hb2.forward(); hb2.ctrl_reset()
    a = torch.zeros(B, 7, device="cuda").uniform_(-1, 1)
    for _ in range(3): hb2.control_step(a, 25)
    hb2.sync(); t0 = time.time()
    for _ in range(10): hb2.control_step(a, 25)
    hb2.sync(); dt = (time.time() - t0) / 10
    print(f"B={B}: {dt*1e3:.2f} ms per control step -> {B/dt:.0f} env-steps/s")
