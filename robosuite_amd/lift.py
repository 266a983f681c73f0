"""Batched Lift / Panda / OSC_POSE host logic: the per-episode part of the reference's `Lift` environment that sits
either side of the fused control-step kernel.

Restates (does not import) the reference's reset path for BASELINE configs 1-2:
  * per-episode cube size draw                lift.py:311-318 -> utils/mjcf_utils.py:470-504
  * arm initial joint noise                   robots/robot.py:247-259 (gaussian, magnitude 0.02, robots/robot.py:110-111)
  * cube placement (x, y, yaw)                utils/placement_samplers.py:221-309, lift.py:321-333
  * gripper init qpos                         models/grippers/panda_gripper.py:35-37
RNG draw order per hard reset: size U x3 -> arm N(0,1) x7 -> x U -> y U -> yaw U.
Pinned by tests/test_lift_host.py against fixtures recorded from the reference's own code.
"""
from __future__ import annotations

import numpy as np

from .reset_bank import ResetBankMixin

PANDA_INIT_QPOS = np.array([0, np.pi / 16.0, 0.00, -np.pi / 2.0 - np.pi / 3.0, 0.00, np.pi - 0.2, np.pi / 4])  # panda_robot.py:37
PANDA_GRIPPER_INIT_QPOS = np.array([0.020833, -0.020833])  # panda_gripper.py:36
TABLE_OFFSET = np.array([0.0, 0.0, 0.8])  # lift.py:153
CUBE_DENSITY = 1000.0  # models/objects/generated_objects.py:680-699 (PrimitiveObject default)


def reset_draws(rng: np.random.Generator):
    """One hard-reset block of draws from the env's generator, in the reference's order."""
    size = rng.uniform(0.020, 0.022, 3)  # BoxObject(size_min, size_max), one U per axis via np.array low/high
    arm = PANDA_INIT_QPOS + rng.standard_normal(7) * 0.02
    x = rng.uniform(-0.03, 0.03)
    y = rng.uniform(-0.03, 0.03)
    yaw = rng.uniform(0.0, 2.0 * np.pi)
    return dict(size=size, arm=arm, xy=np.array([x, y]), yaw=yaw)


def initial_qpos(draw) -> np.ndarray:
    """qpos[16] = [arm x7, finger x2, cube xyz, cube quat wxyz] after Robot.reset + placement (lift.py:401-415)."""
    q = np.zeros(16)
    q[:7] = draw["arm"]
    q[7:9] = PANDA_GRIPPER_INIT_QPOS
    half_z = draw["size"][2]
    q[9:11] = TABLE_OFFSET[:2] + draw["xy"]
    q[11] = TABLE_OFFSET[2] + 0.01 + half_z  # z_offset=0.01, minus object bottom_offset (= -half height)
    q[12] = np.cos(draw["yaw"] / 2.0)
    q[15] = np.sin(draw["yaw"] / 2.0)
    return q


def cube_model_rows(flat, sizes: np.ndarray, density: float = CUBE_DENSITY):
    """Model arrays that depend on the per-episode cube half-sizes, for n envs at once.

    The reference rebuilds and recompiles the whole MJCF per reset (base.py:290-295); the cube is a free body, decoupled from the
    arm in the mass matrix, so the affected compiled fields have closed forms (checked against a full recompile in
    tests/test_lift_host.py): geom_size/geom_rbound of cube_g0 (+ its visual twin), body_mass/inertia/subtreemass,
    body_invweight0 and dof_invweight0 of the free joint.
    Returns {field: float64 [n, count_per_env]} ready for rsim_model_param_set.
    """
    sizes = np.asarray(sizes, dtype=np.float64).reshape(-1, 3)
    n = sizes.shape[0]
    cb = flat.name2id("body", "cube_main")
    g0 = flat.name2id("geom", "cube_g0")
    gv = flat.name2id("geom", "cube_g0_vis")
    jadr = int(flat.body_jntadr[cb])
    d0 = int(flat.jnt_dofadr[jadr])
    sx, sy, sz = sizes[:, 0], sizes[:, 1], sizes[:, 2]
    mass = density * 8.0 * sx * sy * sz
    inertia = np.stack([mass / 3.0 * (sy**2 + sz**2), mass / 3.0 * (sx**2 + sz**2), mass / 3.0 * (sx**2 + sy**2)], axis=1)

    def tile(a):
        return np.repeat(np.asarray(a, dtype=np.float64)[None], n, axis=0).copy()

    geom_size = tile(flat.geom_size)
    geom_size[:, g0] = sizes
    geom_size[:, gv] = sizes
    geom_rbound = tile(flat.geom_rbound)
    geom_rbound[:, g0] = np.linalg.norm(sizes, axis=1)
    geom_rbound[:, gv] = geom_rbound[:, g0]
    body_mass = tile(flat.body_mass)
    old_mass = flat.body_mass[cb]
    body_mass[:, cb] = mass
    body_inertia = tile(flat.body_inertia)
    body_inertia[:, cb] = inertia
    sub = tile(flat.body_subtreemass)
    sub[:, cb] = mass
    sub[:, 0] += mass - old_mass
    biw = tile(flat.body_invweight0)
    biw[:, cb, 0] = 1.0 / mass
    biw[:, cb, 1] = np.mean(1.0 / inertia, axis=1)
    diw = tile(flat.dof_invweight0)
    diw[:, d0:d0 + 3] = (1.0 / mass)[:, None]
    diw[:, d0 + 3:d0 + 6] = np.mean(1.0 / inertia, axis=1)[:, None]
    return {
        "geom_size": geom_size.reshape(n, -1), "geom_rbound": geom_rbound.reshape(n, -1), "body_mass": body_mass.reshape(n, -1),
        "body_inertia": body_inertia.reshape(n, -1), "body_subtreemass": sub.reshape(n, -1), "body_invweight0": biw.reshape(n, -1),
        "dof_invweight0": diw.reshape(n, -1),
    }


def episode_setup(seed0: int, env_ids, block: int = 0):
    """Draws for the global env ids `env_ids`: env i uses default_rng(seed0 + i); `block` selects which
    hard-reset block of that generator (0 = the state after make(), 1 = after the first user reset(), ...)."""
    sizes, qpos = [], []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng)
        sizes.append(d["size"])
        qpos.append(initial_qpos(d))
    return np.array(sizes), np.array(qpos)


def env_actions(env_ids, n_steps: int, scale: float = 1.0, action_dim: int = 7):
    """Per-env action streams a_t ~ scale * U(-1,1)^7 from default_rng(10**6 + i); returns [n_steps, n, action_dim] float32."""
    out = np.empty((n_steps, len(env_ids), action_dim), dtype=np.float32)
    for k, i in enumerate(env_ids):
        out[:, k, :] = scale * np.random.default_rng(10**6 + int(i)).uniform(-1, 1, (n_steps, action_dim))
    return out


def lift_task(flat, cfg, reward_scale: float = 1.0, reward_shaping: bool = True):
    """Observation program + reward description of Lift/Panda for the on-device epilogue (include/rsim.h rsim_task_desc).

    Key order = the reference's `_get_observations` order for use_object_obs=True, use_camera_obs=False:
    robot0_joint_pos, _cos, _sin, joint_vel, joint_acc, eef_pos, eef_quat (BODY robot0_right_hand, xyzw), eef_quat_site, gripper_qpos,
    gripper_qvel (robots/robot.py:334-484), cube_pos, cube_quat, gripper_to_cube_pos (lift.py:356-399)."""
    names = flat.names
    site = int(cfg["eef_site"])                      # gripper0_right_grip_site
    eef_body = names["body"].index("robot0_right_hand")
    cube_body = names["body"].index("cube_main")
    qi, di = cfg["qpos_idx"], cfg["dof_idx"]
    gq, gd = cfg["grip_qpos_idx"], cfg["grip_dof_idx"]
    obs = []
    obs += [("qpos", q, 0) for q in qi] + [("cos", q, 0) for q in qi] + [("sin", q, 0) for q in qi]
    obs += [("qvel", d, 0) for d in di] + [("qacc", d, 0) for d in di]
    obs += [("site_pos", site, k) for k in range(3)] + [("body_quat", eef_body, k) for k in range(4)] + [("site_quat", site, k) for k in range(4)]
    obs += [("qpos", q, 0) for q in gq] + [("qvel", d, 0) for d in gd]
    obs += [("body_pos", cube_body, k) for k in range(3)] + [("body_quat", cube_body, k) for k in range(4)]
    obs += [("body_minus_site", cube_body, k | (site << 2)) for k in range(3)]
    g = names["geom"]
    return dict(obs=obs, task="lift", object_body=cube_body, grip_site=site, table_height=float(TABLE_OFFSET[2]), lift_margin=0.04,
                reward_scale=reward_scale, reward_shaping=reward_shaping,
                left_pad_geoms=[g.index("gripper0_right_finger1_pad_collision")], right_pad_geoms=[g.index("gripper0_right_finger2_pad_collision")],
                object_geoms=[g.index("cube_g0")])


class LiftBatch(ResetBankMixin):
    """B Lift/Panda/OSC_POSE environments resident on one GPU, stepped by the fused HIP control-step kernel.

    `env_ids` are GLOBAL env indices (results do not depend on how envs are sharded over GPUs)."""

    def __init__(self, flat, cfg, env_ids, device: int = 0, seed0: int = 0, per_env_cube: bool = True, horizon: int = 0, bank_episodes: int = 0):
        from .backend import HipBatch, HipModel

        self.flat, self.cfg = flat, cfg
        self.env_ids = np.asarray(env_ids, dtype=np.int64)
        self.B = len(self.env_ids)
        self.model = HipModel(flat)
        self.model.set_controller(cfg)
        self.model.set_task(lift_task(flat, cfg))
        self.batch = HipBatch(self.model, self.B, device, per_env_params=per_env_cube)
        self.per_env_cube = per_env_cube
        self.seed0 = seed0
        self.horizon = horizon
        self.reset()
        if horizon:
            self.batch.set_episode(horizon)
        if bank_episodes:
            # hard resets drawn ahead per env (block k of each env's generator = its episode k, the reference's draw order): when an env reaches
            # the horizon the kernel re-initialises it in place (MujocoEnv.reset with hard_reset, base.py:277-347); reset_bank.py keeps the ring filled
            self.install_reset_bank(bank_episodes)

    def _bank_slots(self):
        if not hasattr(self, "_slots"):
            slots = []  # (field, element, float-table offset) of every entry that changes with the per-episode cube size
            if self.per_env_cube:
                base = {k: np.asarray(self.flat.arrays[k], dtype=np.float64).ravel() for k in cube_model_rows(self.flat, self.sizes[:1])}
                probe = cube_model_rows(self.flat, np.array([[0.0201, 0.0213, 0.0207]]))
                for k, rows in probe.items():
                    for e in np.nonzero(np.abs(rows[0] - base[k]) > 0)[0]:
                        off = self.batch.param_offset(k, int(e))
                        if off >= 0:
                            slots.append((k, int(e), off))
            self._slots = slots
        return self._slots

    def _bank_patch_offsets(self):
        return [o for _, _, o in self._bank_slots()]

    _draw_fn = staticmethod(reset_draws)

    def _episode(self, idx, episode):
        """(cube sizes, qpos) of episode `episode` for the LOCAL env indices idx; equals episode_setup(seed0, env_ids[idx], episode), one block per call."""
        d = self.episode_draws(idx, episode)
        return np.array([x["size"] for x in d]).reshape(-1, 3), np.array([initial_qpos(x) for x in d]).reshape(-1, 16)

    def _bank_rows(self, idx, episode):
        sizes, qpos = self._episode(idx, episode)
        rows = cube_model_rows(self.flat, sizes) if self._bank_slots() else {}
        patch = np.stack([rows[k][:, e] for k, e, _ in self._bank_slots()], axis=1) if self._bank_slots() else np.zeros((len(idx), 0))
        return qpos, patch

    def reset(self, block: int = 0):
        sizes, qpos = self._episode(np.arange(self.B), block)
        b = self.batch
        if self.per_env_cube:
            for field, rows in cube_model_rows(self.flat, sizes).items():
                b.param_set(field, rows)
        b.set("qpos", qpos)
        b.set("qvel", 0.0)
        b.set("ctrl", 0.0)
        b.set("time", 0.0)
        b.set("qacc_warmstart", 0.0)
        b.forward()                                   # MujocoEnv.reset: sim.forward() (base.py:298-303); warm start stays 0 (fresh MjData)
        b.ctrl_reset()                                # fresh controller objects per reset (robots/robot.py:271)
        self.sizes, self.qpos0 = sizes, qpos

    def step(self, actions, n_sub: int = 25):
        """One env.step for every env: fused physics + controllers + observation / reward epilogue (results stay on the device)."""
        self.batch.control_step(actions, n_sub)
        self._bank_tick()

    def obs(self):
        return self.batch.tensor("obs")

    def reward(self):
        return self.batch.tensor("reward")

    def success(self):
        return self.batch.tensor("success")


def LiftVecEnv(n_envs: int, device: int = 0, seed: int = 0, horizon: int = 500, env_ids=None, bank_episodes: int = 4, flat=None, cfg=None):
    """Vectorised `suite.make("Lift", robots="Panda", ...)` on the packaged model: `vec_env.VecEnv("Lift", ...)` (one facade for all tasks)."""
    import json
    import os

    from . import mjcf
    from .vec_env import VecEnv

    if flat is None:
        adir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
        flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim"))
        cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
    return VecEnv("Lift", n_envs, flat, cfg, device=device, seed=seed, horizon=horizon, env_ids=env_ids, bank_episodes=bank_episodes)
