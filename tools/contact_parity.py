"""Contact-by-contact comparison of the kernel's narrow phase with the fp64 oracle on reached states of a full-size batch (runs on the GPU box):
depth, normal angle, constraint forces and accelerations per env.  Usage: python tools/contact_parity.py [baxter|pickplace] [steps]"""
import os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift
from tests.util import load_golden
from tests.test_full_size_parity import oracle_for_env
which = sys.argv[1] if len(sys.argv) > 1 else "baxter"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
if which == "baxter":
    from robosuite_amd import peg_in_hole
    g, cfg, flat = load_golden("ctl_joint_velocity", "peg_baxter")
    B = 2048; env = peg_in_hole.PegBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2)
else:
    from robosuite_amd import pick_place
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    B = 2048; env = pick_place.PickPlaceBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2)
tape = torch.tensor(lift.env_actions(np.arange(B), steps, action_dim=env.model.action_dim), device="cuda")
for t in range(steps): env.step(tape[t])
hb = env.batch
q, v, ws, ctrl = hb.get("qpos"), hb.get("qvel"), hb.get("qacc_warmstart"), hb.get("ctrl")
hb.forward()
ncon, nefc, qacc, efc, con = hb.get("ncon"), hb.get("nefc"), hb.get("qacc"), hb.get("efc_force"), hb.get("contact")
have = np.nonzero(ncon > 0)[0][:int(os.environ.get("NMAX", 96))]
stat, struct_bad, percon = [], 0, []
for e in have:
    om, od = oracle_for_env(flat, hb, int(e))
    od.qpos[:] = q[e]; od.qvel[:] = v[e]; od.qacc_warmstart[:] = ws[e]; od.ctrl[:] = ctrl[e]; od.forward()
    if od.ncon != ncon[e] or od.nefc != nefc[e]: struct_bad += 1; continue
    of = np.asarray(od.efc_force)
    fe = np.abs(efc[e][:od.nefc] - of).max() / max(1.0, np.abs(of).max()); qe = np.abs(qacc[e] - od.qacc).max() / max(1.0, np.abs(od.qacc).max())
    worst = 0.0; wd = 0.0
    for c, oc in enumerate(od.contacts()):
        hc = con[e][c]
        ang = float(np.degrees(np.arccos(np.clip(np.dot(hc[4:7], oc["frame"][0]), -1, 1))))
        worst = max(worst, ang); wd = max(wd, abs(hc[0] - oc["dist"]))
        if ang > 0.1 or abs(hc[0] - oc["dist"]) > 1e-5:
            percon.append((int(e), str(flat.names["geom"][oc["geom1"]]), int(flat.geom_type[oc["geom1"]]), str(flat.names["geom"][oc["geom2"]]), int(flat.geom_type[oc["geom2"]]),
                           round(oc["dist"], 6), round(float(hc[0]), 6), round(ang, 2), np.round(oc["frame"][0], 3).tolist(), np.round(hc[4:7], 3).tolist()))
    stat.append((int(e), od.ncon, worst, wd, fe, qe))
print(f"{which}: {len(have)} envs with contacts, structure differs in {struct_bad}")
a = np.array([s[2:] for s in stat])
for th in (0.1, 1.0, 10.0): print(f"  envs with a contact normal > {th} deg off: {(a[:, 0] > th).sum()} of {len(a)}")
print(f"  envs with a depth > 1e-5 m off: {(a[:, 1] > 1e-5).sum()}")
good = a[(a[:, 0] <= 0.1) & (a[:, 1] <= 1e-5)]
print(f"  geometry-agreeing envs: {len(good)}; on those max rel force err {good[:, 2].max():.2e}, max rel qacc err {good[:, 3].max():.2e}")
print("  worst:", sorted(stat, key=lambda s: -s[2])[:8])

pc = collections.Counter((p[1], p[3]) for p in percon)
print("  disagreeing contacts by pair:", pc.most_common(12))
for p in percon[:14]: print("   ", p)
