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

import functools

import numpy as np

from .reset_bank import ResetBankMixin

from .lift import (PANDA_GRIPPER_INIT_QPOS, PANDA_INIT_QPOS, TABLE_OFFSET, arm_noise, env_actions, grasp_groups, robot_obs_program, sample_objects,  # noqa: F401  (same robot, arena and action streams)
                   task_env_args)

CUBE_HALF = {"cubeA": 0.02, "cubeB": 0.025}   # stack.py:324-337
XY_RANGE = 0.08                                # stack.py:350-351
Z_OFFSET = 0.01                                # stack.py:356


def default_reset_spec():
    """The reset of `suite.make("Stack", robots="Panda")` with the reference's defaults, in the form factory.reset_cfg reads off a live env."""
    objs = [dict(name=n, horizontal_radius=float(np.linalg.norm([h, h])), bottom_z=-h, top_z=h, qposadr=9 + 7 * k, init_quat=None) for k, (n, h) in enumerate(CUBE_HALF.items())]
    return dict(nq=23, arm_init_qpos=[float(x) for x in PANDA_INIT_QPOS], arm_qpos_idx=list(range(7)), noise=dict(type="gaussian", magnitude=0.02),
                grippers=[dict(init_qpos=[float(x) for x in PANDA_GRIPPER_INIT_QPOS], qpos_idx=[7, 8])],
                sampler=dict(x_range=[-XY_RANGE, XY_RANGE], y_range=[-XY_RANGE, XY_RANGE], rotation=None, rotation_axis="z", z_offset=Z_OFFSET,
                             reference_pos=[float(x) for x in TABLE_OFFSET], ensure_object_boundary_in_range=False, ensure_valid_placement=True, objects=objs))


def reset_draws(rng: np.random.Generator, spec=None, aux=None):
    """One hard-reset block of draws from the env's generator, in the reference's order: robot joint noise, then the sampler over its objects
    (placement_samplers.py:221-309).  `spec` = cfg["reset"] (factory.reset_cfg); None = the Panda defaults."""
    spec = default_reset_spec() if spec is None else spec
    arm = arm_noise(rng, spec)
    sm = spec["sampler"]
    placed = sample_objects(aux if (sm.get("own_rng") and aux is not None) else rng, sm, [(o["horizontal_radius"], o["bottom_z"], o["top_z"]) for o in sm["objects"]])
    return dict(arm=arm, objects=placed)


def initial_qpos(draw, spec=None) -> np.ndarray:
    """qpos after Robot.reset + placement (stack.py:403-415): arm joints, gripper init_qpos, each cube's free joint (pos, quat wxyz)."""
    spec = default_reset_spec() if spec is None else spec
    q = np.zeros(int(spec["nq"]))
    q[spec["arm_qpos_idx"]] = draw["arm"]
    for g in spec["grippers"]:
        q[g["qpos_idx"]] = g["init_qpos"]
    for o, (pos, quat) in zip(spec["sampler"]["objects"], draw["objects"]):
        a = int(o["qposadr"])
        q[a:a + 3] = pos
        q[a + 3:a + 7] = quat
    return q


def episode_setup(seed0: int, env_ids, block: int = 0, spec=None):
    """qpos for the global env ids: env i uses default_rng(seed0 + i); `block` selects the hard-reset block of that generator."""
    qpos = []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng, spec)
        qpos.append(initial_qpos(d, spec))
    return np.array(qpos)


def stack_task(flat, cfg, reward_scale=None, reward_shaping=None):
    """Observation program + reward description of Stack for the on-device epilogue (include/rsim.h rsim_task_desc, task 2).

    Key order = the reference's `_get_observations` order: the robot keys of Lift (robots/robot.py:334-484), then cubeA_pos, cubeA_quat,
    cubeB_pos, cubeB_quat, cubeA_to_cubeB (= cubeB_pos - cubeA_pos), gripper_to_cubeA, gripper_to_cubeB (stack.py:417-461).  Reward flavour and scale:
    cfg["env"] (stack.py:161-162, 258-264)."""
    names = flat.names
    site = int(cfg["eef_site"])
    eef_body, lpad, rpad = grasp_groups(flat, cfg)
    A, B = names["body"].index("cubeA_main"), names["body"].index("cubeB_main")
    obs = robot_obs_program(cfg, site, eef_body)
    if "cubeA_pos" in cfg.get("obs_keys", ["cubeA_pos"]):     # use_object_obs (stack.py:417)
        for body in (A, B):
            obs += [("body_pos", body, k) for k in range(3)] + [("body_quat", body, k) for k in range(4)]
        obs += [("body_minus_body", B, k | (A << 2)) for k in range(3)]
        obs += [("body_minus_site", A, k | (site << 2)) for k in range(3)] + [("body_minus_site", B, k | (site << 2)) for k in range(3)]
    if sum(cfg.get("obs_dims", [len(obs)])) != len(obs):
        raise NotImplementedError(f"Stack observation record: the reference env returned {sum(cfg['obs_dims'])} floats under keys {cfg['obs_keys']}, the on-device program has {len(obs)}")
    scale, shaping = task_env_args(cfg, "stack", reward_scale, reward_shaping)
    g = names["geom"]
    return dict(obs=obs, task="stack", object_body=A, object2_body=B, grip_site=site, table_height=float(cfg.get("table_height", TABLE_OFFSET[2])),
                lift_margin=0.04, reward_scale=scale, reward_shaping=shaping, left_pad_geoms=lpad, right_pad_geoms=rpad,
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
        self.spec = cfg.get("reset") or default_reset_spec()     # cfg["reset"]: the reference env's own reset configuration (factory.reset_cfg)
        self.n_sub = int(cfg.get("env", {}).get("n_sub", 25))
        self._draw_fn = functools.partial(reset_draws, spec=self.spec)
        self._draw_aux = bool(self.spec["sampler"].get("own_rng"))
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

    def _episode(self, idx, episode):
        return np.array([initial_qpos(d, self.spec) for d in self.episode_draws(idx, episode)]).reshape(-1, int(self.spec["nq"]))

    def _bank_rows(self, idx, episode):
        return self._episode(idx, episode), np.zeros((len(idx), 0))

    def reset(self, block: int = 0):
        qpos = self._episode(np.arange(self.B), block)
        b = self.batch
        b.set("qpos", qpos); b.set("qvel", 0.0); b.set("ctrl", 0.0); b.set("time", 0.0); b.set("qacc_warmstart", 0.0)
        b.forward()        # MujocoEnv.reset: sim.forward() (base.py:298-303)
        b.ctrl_reset()     # fresh controller objects per reset (robots/robot.py:271)
        self.qpos0 = qpos

    def step(self, actions, n_sub: int = 0):
        self.batch.control_step(actions, n_sub or self.n_sub)
        self._bank_tick()

    def obs(self):
        return self.batch.tensor("obs")

    def reward(self):
        return self.batch.tensor("reward")

    def success(self):
        return self.batch.tensor("success")
