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
