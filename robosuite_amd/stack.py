"""Batched Stack / Panda / OSC_POSE host logic (BASELINE configs[2]): the per-episode part of the reference's `Stack` environment either
side of the fused control-step kernel (32-dof kernel configuration: arm 7 + fingers 2 + two free cubes 6 + 6).

Restates (does not import) the reference's reset path:
  * arm initial joint noise      robots/robot.py:247-259 (gaussian, magnitude 0.02)
  * cube placement               stack.py:346-357, 403-415 -> utils/placement_samplers.py:221-309 (UniformRandomSampler, x/y in +-0.08, random yaw,
                                 ensure_valid_placement: a candidate closer than the two horizontal radii to an already placed cube is redrawn)
  * cube sizes                   fixed (size_min == size_max, stack.py:324-337); BoxObject gets no rng, so the env generator is not consumed
RNG draw order per hard reset: arm N(0,1) x7 -> cubeA {x U, y U}(+ redraws) -> yaw U -> cubeB {x U, y U}(+ redraws) -> yaw U.
Pinned by tests/test_stack_host.py against reset states recorded from the reference's own code (tests/golden/stack_panda_resets.npz).
"""
from __future__ import annotations

import numpy as np

from .reset_bank import ResetBankMixin

from .lift import PANDA_GRIPPER_INIT_QPOS, PANDA_INIT_QPOS, TABLE_OFFSET, env_actions  # noqa: F401  (same robot, arena and action streams)

CUBE_HALF = {"cubeA": 0.02, "cubeB": 0.025}   # stack.py:324-337
XY_RANGE = 0.08                                # stack.py:350-351
Z_OFFSET = 0.01                                # stack.py:356


def reset_draws(rng: np.random.Generator):
    """One hard-reset block of draws from the env's generator, in the reference's order."""
    arm = PANDA_INIT_QPOS + rng.standard_normal(7) * 0.02
    placed = []   # (x, y, z, half)
    out = {}
    for name in ("cubeA", "cubeB"):
        half = CUBE_HALF[name]
        radius = np.linalg.norm([half, half])
        for _ in range(5000):
            x = rng.uniform(-XY_RANGE, XY_RANGE) + TABLE_OFFSET[0]
            y = rng.uniform(-XY_RANGE, XY_RANGE) + TABLE_OFFSET[1]
            z = Z_OFFSET + TABLE_OFFSET[2] + half
            ok = True
            for (px, py, pz, ph) in placed:
                if np.linalg.norm((x - px, y - py)) <= np.linalg.norm([ph, ph]) + radius and z - pz <= ph + half:
                    ok = False
                    break
            if ok:
                yaw = rng.uniform(0.0, 2.0 * np.pi)
                placed.append((x, y, z, half))
                out[name] = (np.array([x, y, z]), yaw)
                break
        else:
            raise RuntimeError("Cannot place all objects")   # RandomizationError in the reference
    return dict(arm=arm, **out)


def initial_qpos(draw) -> np.ndarray:
    """qpos[23] = [arm x7, finger x2, cubeA xyz + quat wxyz, cubeB xyz + quat wxyz] after Robot.reset + placement (stack.py:403-415)."""
    q = np.zeros(23)
    q[:7] = draw["arm"]
    q[7:9] = PANDA_GRIPPER_INIT_QPOS
    for k, name in enumerate(("cubeA", "cubeB")):
        pos, yaw = draw[name]
        o = 9 + 7 * k
        q[o:o + 3] = pos
        q[o + 3] = np.cos(yaw / 2.0)
        q[o + 6] = np.sin(yaw / 2.0)
    return q


def episode_setup(seed0: int, env_ids, block: int = 0):
    """qpos for the global env ids: env i uses default_rng(seed0 + i); `block` selects the hard-reset block of that generator."""
    qpos = []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng)
        qpos.append(initial_qpos(d))
    return np.array(qpos)


def stack_task(flat, cfg, reward_scale: float = 1.0, reward_shaping: bool = True):
    """Observation program + reward description of Stack/Panda for the on-device epilogue (include/rsim.h rsim_task_desc, task 2).

    Key order = the reference's `_get_observations` order: the robot keys of Lift (robots/robot.py:334-484), then cubeA_pos, cubeA_quat,
    cubeB_pos, cubeB_quat, cubeA_to_cubeB (= cubeB_pos - cubeA_pos), gripper_to_cubeA, gripper_to_cubeB (stack.py:417-461)."""
    names = flat.names
    site = int(cfg["eef_site"])
    eef_body = names["body"].index("robot0_right_hand")
    A, B = names["body"].index("cubeA_main"), names["body"].index("cubeB_main")
    qi, di = cfg["qpos_idx"], cfg["dof_idx"]
    gq, gd = cfg["grip_qpos_idx"], cfg["grip_dof_idx"]
    obs = []
    obs += [("qpos", q, 0) for q in qi] + [("cos", q, 0) for q in qi] + [("sin", q, 0) for q in qi]
    obs += [("qvel", d, 0) for d in di] + [("qacc", d, 0) for d in di]
    obs += [("site_pos", site, k) for k in range(3)] + [("body_quat", eef_body, k) for k in range(4)] + [("site_quat", site, k) for k in range(4)]
    obs += [("qpos", q, 0) for q in gq] + [("qvel", d, 0) for d in gd]
    for body in (A, B):
        obs += [("body_pos", body, k) for k in range(3)] + [("body_quat", body, k) for k in range(4)]
    obs += [("body_minus_body", B, k | (A << 2)) for k in range(3)]
    obs += [("body_minus_site", A, k | (site << 2)) for k in range(3)] + [("body_minus_site", B, k | (site << 2)) for k in range(3)]
    g = names["geom"]
    return dict(obs=obs, task="stack", object_body=A, object2_body=B, grip_site=site, table_height=float(cfg.get("table_height", TABLE_OFFSET[2])),
                lift_margin=0.04, reward_scale=reward_scale, reward_shaping=reward_shaping,
                left_pad_geoms=[g.index("gripper0_right_finger1_pad_collision")], right_pad_geoms=[g.index("gripper0_right_finger2_pad_collision")],
                object_geoms=[g.index("cubeA_g0")], object2_geoms=[g.index("cubeB_g0")])


class StackBatch(ResetBankMixin):
    """B Stack/Panda/OSC_POSE environments resident on one GPU (32-dof configuration of the fused kernel).  `env_ids` are GLOBAL indices."""

    def __init__(self, flat, cfg, env_ids, device: int = 0, seed0: int = 0, horizon: int = 0, bank_episodes: int = 0):
        from .backend import HipBatch, HipModel

        self.flat, self.cfg = flat, cfg
        self.env_ids = np.asarray(env_ids, dtype=np.int64)
        self.B = len(self.env_ids)
        self.model = HipModel(flat)
        self.model.set_controller(cfg)
        self.model.set_task(stack_task(flat, cfg))
        self.batch = HipBatch(self.model, self.B, device, per_env_params=False)   # cube sizes are fixed: one shared model
        self.seed0 = seed0
        self.horizon = horizon
        self.reset()
        if horizon:
            self.batch.set_episode(horizon)
        if bank_episodes:
            self.install_reset_bank(bank_episodes)

    def _bank_patch_offsets(self):
        return []

    _draw_fn = staticmethod(reset_draws)

    def _episode(self, idx, episode):
        return np.array([initial_qpos(d) for d in self.episode_draws(idx, episode)]).reshape(-1, 23)

    def _bank_rows(self, idx, episode):
        return self._episode(idx, episode), np.zeros((len(idx), 0))

    def reset(self, block: int = 0):
        qpos = self._episode(np.arange(self.B), block)
        b = self.batch
        b.set("qpos", qpos); b.set("qvel", 0.0); b.set("ctrl", 0.0); b.set("time", 0.0); b.set("qacc_warmstart", 0.0)
        b.forward()        # MujocoEnv.reset: sim.forward() (base.py:298-303)
        b.ctrl_reset()     # fresh controller objects per reset (robots/robot.py:271)
        self.qpos0 = qpos

    def step(self, actions, n_sub: int = 25):
        self.batch.control_step(actions, n_sub)
        self._bank_tick()

    def obs(self):
        return self.batch.tensor("obs")

    def reward(self):
        return self.batch.tensor("reward")

    def success(self):
        return self.batch.tensor("success")
