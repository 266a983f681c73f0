"""Shared helpers for the test-suite: golden fixtures, oracle and HIP object construction."""
import json
import os

import numpy as np

from robosuite_amd import mjcf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ("seed0_gentle", "seed1_full")


def load_golden(tag, model="lift_panda"):
    g = np.load(os.path.join(GOLD, f"{model}_{tag}.npz"))
    cfg = json.load(open(os.path.join(GOLD, f"{model}_{tag}.cfg.json")))
    flat = mjcf.load_model(os.path.join(GOLD, f"{model}_{tag}.rsim"))
    return g, cfg, flat


def make_oracle(flat, cfg=None):
    from oracle.oracle import OracleController, OracleData, OracleModel

    om = OracleModel(mjcf.to_blob(flat))
    d = OracleData(om)
    c = OracleController(cfg) if cfg is not None else None
    return om, d, c


def make_hip(flat, cfg=None, B=1, per_env=False):
    from robosuite_amd.backend import HipBatch, HipModel

    hm = HipModel(flat)
    if cfg is not None:
        hm.set_controller(cfg)
    return hm, HipBatch(hm, B, 0, per_env)


def scripted_grasp_and_lift(flat, cfg, qpos0, n_steps=80):
    """Closed-loop scripted policy on the CPU oracle (hover -> descend -> close -> lift), the shape of the reference's behavioural
    gripper test (models/grippers/gripper_tester.py:204-226).  Returns the action tape, the oracle's states and the final cube height."""
    om, od, oc = make_oracle(flat, cfg)
    nq = flat.nq
    od.qpos[:] = qpos0; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.forward(); oc.reset(od)
    site, cube = cfg["eef_site"], flat.names["body"].index("cube_main")
    acts, qs, phase, hold = [], [], 0, 0
    for t in range(n_steps):
        e, c = od.site_xpos[3 * site:3 * site + 3].copy(), od.xpos[3 * cube:3 * cube + 3].copy()
        a = np.zeros(7)
        if phase == 0:
            d = c + np.array([0, 0, 0.08]) - e; a[:3] = np.clip(d / 0.05, -1, 1); a[6] = -1
            if np.linalg.norm(d) < 0.01: phase = 1
        elif phase == 1:
            d = c - e; a[:3] = np.clip(d / 0.05, -1, 1); a[6] = -1
            if np.linalg.norm(d) < 0.008: phase = 2
        elif phase == 2:
            a[6] = 1; hold += 1
            if hold > 12: phase = 3
        else:
            a[2] = 0.6; a[6] = 1
        acts.append(a)
        oc.env_step(od, a, 25)
        qs.append(np.concatenate([od.qpos, od.qvel]))
    return np.array(acts), np.array(qs), float(od.xpos[3 * cube + 2]), od
