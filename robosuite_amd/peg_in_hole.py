"""TwoArmPegInHole / Baxter (single-robot, no grippers; BASELINE configs[3]) host side for the fused kernel's 64-body configuration.

Task program (observation order, reward) of the reference env restated for the on-device epilogue:
  robot keys   robots/robot.py:334-484 with two arms: joint_pos / cos / sin / vel / acc over all 14 arm joints, then per arm
               {prefix}eef_pos (site), {prefix}eef_quat (body, xyzw), {prefix}eef_quat_site
  object keys  two_arm_peg_in_hole.py:414-475: hole_pos, hole_quat, peg_to_hole (= hole_pos - peg_pos), peg_quat, angle (cos), t, d
  reward       two_arm_peg_in_hole.py:240-290, success :513-521
The reset path of this env draws nothing for the objects (peg and hole are welded to the hands, :488-497); the per-episode peg size
(two_arm_peg_in_hole.py:175-176) is a model edit that is not restated yet: batches use the model they were compiled from.
"""
from __future__ import annotations

import numpy as np


def peg_task(flat, cfg, reward_scale: float = 1.0, reward_shaping: bool = True):
    names = flat.names
    body = names["body"]
    peg, hole = body.index("peg_main"), body.index("hole_main")
    qi, di = cfg["qpos_idx"], cfg["dof_idx"]
    obs = []
    obs += [("qpos", q, 0) for q in qi] + [("cos", q, 0) for q in qi] + [("sin", q, 0) for q in qi]
    obs += [("qvel", d, 0) for d in di] + [("qacc", d, 0) for d in di]
    for arm in ("right", "left"):
        site = names["site"].index(f"gripper0_{arm}_grip_site")
        hand = body.index(f"robot0_{arm}_hand")
        obs += [("site_pos", site, k) for k in range(3)] + [("body_quat", hand, k) for k in range(4)] + [("site_quat", site, k) for k in range(4)]
    obs += [("body_pos", hole, k) for k in range(3)] + [("body_quat", hole, k) for k in range(4)]
    obs += [("body_minus_body", hole, k | (peg << 2)) for k in range(3)]
    obs += [("body_quat", peg, k) for k in range(4)]
    obs += [("peg_cos", 0, 0), ("peg_t", 0, 0), ("peg_d", 0, 0)]
    return dict(obs=obs, task="peg_in_hole", object_body=peg, object2_body=hole, grip_site=0, reward_scale=reward_scale, reward_shaping=reward_shaping)


BAXTER_INIT_QPOS = np.array([0.403, -0.636, 0.114, 1.432, 0.735, 1.205, -0.269, -0.403, -0.636, -0.114, 1.432, -0.735, 1.205, 0.269])  # baxter_robot.py:47-60
PEG_RADIUS = (0.015, 0.03)   # two_arm_peg_in_hole.py:175
PEG_LENGTH = 0.13            # two_arm_peg_in_hole.py:176


def reset_draws(rng: np.random.Generator):
    """One hard-reset block of the env generator, in the reference's order: CylinderObject size (radius U, length U over a zero-width range,
    utils/mjcf_utils.py:470-504 via two_arm_peg_in_hole.py:343-351), then the arm noise N(0,1) x 14 x 0.02 (robots/robot.py:247-259)."""
    radius = rng.uniform(*PEG_RADIUS)
    rng.uniform(PEG_LENGTH, PEG_LENGTH)
    return dict(peg_radius=radius, qpos=BAXTER_INIT_QPOS + rng.standard_normal(14) * 0.02)


def episode_setup(seed0: int, env_ids, block: int = 0):
    out = []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng)
        out.append(d["qpos"])
    return np.array(out)


class PegBatch:
    """B TwoArmPegInHole/Baxter environments on one GPU (64-body kernel configuration, joint-space part controllers).  The peg keeps the radius of
    the model the batch was built from (see the module docstring)."""

    def __init__(self, flat, cfg, env_ids, device: int = 0, seed0: int = 0, horizon: int = 0, bank_episodes: int = 0):
        from .backend import HipBatch, HipModel

        self.flat, self.cfg = flat, cfg
        self.env_ids = np.asarray(env_ids, dtype=np.int64)
        self.B = len(self.env_ids)
        self.model = HipModel(flat)
        self.model.set_controller(cfg)
        self.model.set_task(peg_task(flat, cfg))
        self.batch = HipBatch(self.model, self.B, device, per_env_params=False)
        self.seed0 = seed0
        self.reset()
        if horizon:
            self.batch.set_episode(horizon)
        if bank_episodes:
            qbank = np.stack([episode_setup(seed0, self.env_ids, ep) for ep in range(bank_episodes)], axis=1).astype(np.float32)
            self.batch.set_reset_bank(qbank, [], np.zeros((self.B, bank_episodes, 0), dtype=np.float32))

    def reset(self, block: int = 0):
        qpos = episode_setup(self.seed0, self.env_ids, block)
        b = self.batch
        b.set("qpos", qpos); b.set("qvel", 0.0); b.set("ctrl", 0.0); b.set("time", 0.0); b.set("qacc_warmstart", 0.0)
        b.forward(); b.ctrl_reset()
        self.qpos0 = qpos

    def step(self, actions, n_sub: int = 25):
        self.batch.control_step(actions, n_sub)

    def obs(self):
        return self.batch.tensor("obs")

    def reward(self):
        return self.batch.tensor("reward")

    def success(self):
        return self.batch.tensor("success")
