"""Dump reached Baxter / JOINT_VELOCITY states with contacts (state + the kernel's forward outputs) for offline analysis against the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, peg_in_hole
from tests.util import load_golden
g, cfg, flat = load_golden("ctl_joint_velocity", "peg_baxter")
B = 2048; ids = np.arange(B)
env = peg_in_hole.PegBatch(flat, cfg, ids, seed0=0, horizon=500, bank_episodes=2)
tape = torch.tensor(lift.env_actions(ids, 50, action_dim=env.model.action_dim), device="cuda")
for t in range(50): env.step(tape[t])
hb = env.batch
out = {k: hb.get(k) for k in ("qpos", "qvel", "qacc_warmstart", "ctrl")}
hb.forward()
for k in ("ncon", "nefc", "niter", "qacc", "efc_force", "contact", "qfrc_constraint", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qM"): out["h_" + k] = hb.get(k)
have = np.nonzero(out["h_ncon"] > 0)[0]
out["have"] = have
for k in ("geom_size", "body_mass", "body_inertia", "body_invweight0", "dof_invweight0", "geom_rbound", "body_subtreemass"):
    out["p_" + k] = hb.param_get(k)[have]
sel = {k: (v[have] if (hasattr(v, "shape") and len(v.shape) and v.shape[0] == B) else v) for k, v in out.items()}
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "baxter_states.npz"), **sel)
print("envs with contacts:", len(have))
