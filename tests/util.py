"""Shared helpers for the test-suite: golden fixtures, oracle and HIP object construction."""
import json
import os

import numpy as np

from robosuite_amd import mjcf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ("seed0_gentle", "seed1_full")


def load_golden(tag, model="lift_panda"):
    """(arrays, controller / key cfg, compiled model) of fixture `{model}_{tag}` under tests/golden (recorded by tools/gen_golden.py)."""
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


def stack_staged_rewards(flat, cfg, od, table_height=0.8):
    """Test-side restatement of Stack.staged_rewards on oracle data (environments/manipulation/stack.py:268-312)."""
    names = flat.names
    A, B, site = names["body"].index("cubeA_main"), names["body"].index("cubeB_main"), cfg["eef_site"]
    gA, gB = names["geom"].index("cubeA_g0"), names["geom"].index("cubeB_g0")
    lp, rp = names["geom"].index("gripper0_right_finger1_pad_collision"), names["geom"].index("gripper0_right_finger2_pad_collision")
    cA, cB, grip = od.xpos[3 * A:3 * A + 3], od.xpos[3 * B:3 * B + 3], od.site_xpos[3 * site:3 * site + 3]
    pairs = {(c["geom1"], c["geom2"]) for c in od.contacts()} | {(c["geom2"], c["geom1"]) for c in od.contacts()}
    grasp = (gA, lp) in pairs and (gA, rp) in pairs
    r_reach = (1 - np.tanh(10.0 * np.linalg.norm(grip - cA))) * 0.25 + (0.25 if grasp else 0.0)
    lifted = cA[2] > table_height + 0.04
    r_lift = (1.0 + 0.5 * (1 - np.tanh(np.linalg.norm(cA[:2] - cB[:2])))) if lifted else 0.0
    r_stack = 2.0 if (not grasp and r_lift > 0 and (gA, gB) in pairs) else 0.0
    return r_reach, r_lift, r_stack


def scripted_stack(flat, cfg, qpos0, n_steps=170):
    """Closed-loop scripted stacking on the CPU oracle: grasp cubeA, carry it over cubeB, lower, release, retreat.
    Returns the action tape, per-step (reward, success) from stack_staged_rewards and the oracle data."""
    om, od, oc = make_oracle(flat, cfg)
    od.qpos[:] = qpos0; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.forward(); oc.reset(od)
    site = cfg["eef_site"]
    A, B = flat.names["body"].index("cubeA_main"), flat.names["body"].index("cubeB_main")
    acts, rew, phase, hold = [], [], 0, 0
    for t in range(n_steps):
        e, a_, b_ = od.site_xpos[3 * site:3 * site + 3].copy(), od.xpos[3 * A:3 * A + 3].copy(), od.xpos[3 * B:3 * B + 3].copy()
        a = np.zeros(7)
        mv = lambda d: np.clip(d / 0.05, -1, 1)
        if phase == 0:
            d = a_ + np.array([0, 0, 0.08]) - e; a[:3] = mv(d); a[6] = -1
            if np.linalg.norm(d) < 0.01: phase = 1
        elif phase == 1:
            d = a_ - e; a[:3] = mv(d); a[6] = -1
            if np.linalg.norm(d) < 0.008: phase = 2
        elif phase == 2:
            a[6] = 1; hold += 1
            if hold > 12: phase, hold = 3, 0
        elif phase == 3:   # lift
            d = np.array([0, 0, 0.95]) - np.array([0, 0, e[2]]); a[:3] = mv(d); a[6] = 1
            if abs(d[2]) < 0.01: phase = 4
        elif phase == 4:   # carry over cubeB
            d = np.array([b_[0] - a_[0], b_[1] - a_[1], 0.0]); a[:3] = mv(d); a[6] = 1
            if np.linalg.norm(d) < 0.004: phase = 5
        elif phase == 5:   # lower until cubeA rests on cubeB
            d = np.array([b_[0] - a_[0], b_[1] - a_[1], (b_[2] + 0.025 + 0.02 + 0.003) - a_[2]]); a[:3] = mv(d); a[6] = 1
            if abs(d[2]) < 0.004: phase = 6
        elif phase == 6:   # release
            a[6] = -1; hold += 1
            if hold > 12: phase = 7
        else:              # retreat
            a[2] = 0.5; a[6] = -1
        acts.append(a)
        oc.env_step(od, a, 25)
        r = stack_staged_rewards(flat, cfg, od)
        rew.append((max(r) / 2.0, r[2] > 0))
    return np.array(acts), rew, od


def make_oracle_parts(flat, cfg):
    """Multi-arm robots: one oracle controller per part (cfg["parts"]); returns (model, data, [(controller, action_dim), ...])."""
    from oracle.oracle import OracleController, OracleData, OracleModel

    om = OracleModel(mjcf.to_blob(flat))
    d = OracleData(om)
    return om, d, [(OracleController(p), len(p["input_min"])) for p in cfg["parts"]]
