"""TwoArmPegInHole / Baxter (single-robot, no grippers; BASELINE configs[3]) host side for the fused kernel's 64-body configuration.

Task program (observation order, reward) of the reference env restated for the on-device epilogue:
  robot keys   robots/robot.py:334-484 with two arms: joint_pos / cos / sin / vel / acc over all 14 arm joints, then per arm
               {prefix}eef_pos (site), {prefix}eef_quat (body, xyzw), {prefix}eef_quat_site
  object keys  two_arm_peg_in_hole.py:414-475: hole_pos, hole_quat, peg_to_hole (= hole_pos - peg_pos), peg_quat, angle (cos), t, d
  reward       two_arm_peg_in_hole.py:240-290, success :513-521
The reset path of this env places no objects (peg and hole are welded to the hands, :488-497); the per-episode peg radius
(two_arm_peg_in_hole.py:175-176, 343-351) is a model edit, restated in closed form by `peg_model_rows` and applied per env / per episode.
"""
from __future__ import annotations

import functools

import numpy as np

from .reset_bank import ResetBankMixin  # noqa: E402


def peg_task(flat, cfg, reward_scale=None, reward_shaping=None):
    from .lift import task_env_args

    reward_scale, reward_shaping = task_env_args(cfg, "peg_in_hole", reward_scale, reward_shaping)     # cfg["env"]: two_arm_peg_in_hole.py:163-164, 284-288
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


def default_reset_spec():
    """The reset of TwoArmPegInHole / Baxter (single-robot, no grippers) with the reference's defaults, in the form factory.reset_cfg reads off a live env."""
    return dict(nq=14, arm_init_qpos=[float(x) for x in BAXTER_INIT_QPOS], arm_qpos_idx=list(range(14)), noise=dict(type="gaussian", magnitude=0.02), grippers=[],
                peg=dict(radius=list(PEG_RADIUS), length=PEG_LENGTH))


def reset_draws(rng: np.random.Generator, spec=None):
    """One hard-reset block of the env generator, in the reference's order: CylinderObject size (radius U, length U over a zero-width range,
    utils/mjcf_utils.py:470-504 via two_arm_peg_in_hole.py:343-351), then the arm noise (robots/robot.py:247-259).  `spec` = cfg["reset"]."""
    from .lift import arm_noise

    spec = default_reset_spec() if spec is None else spec
    radius = rng.uniform(*spec["peg"]["radius"])
    rng.uniform(spec["peg"]["length"], spec["peg"]["length"])
    q = np.zeros(int(spec["nq"]))
    q[spec["arm_qpos_idx"]] = arm_noise(rng, spec)
    return dict(peg_radius=radius, qpos=q)


def episode_setup(seed0: int, env_ids, block: int = 0, with_radius: bool = False, spec=None):
    out, rad = [], []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng, spec)
        out.append(d["qpos"]); rad.append(d["peg_radius"])
    return (np.array(out), np.array(rad)) if with_radius else np.array(out)


class PegBatch(ResetBankMixin):
    """B TwoArmPegInHole/Baxter environments on one GPU (64-body kernel configuration, joint-space part controllers).  With `per_env_peg`
    (default) every env carries its own peg radius, redrawn per episode like the reference's hard reset (closed-form model rows,
    `peg_model_rows`); without it all envs keep the radius of the model the batch was built from."""

    def __init__(self, flat, cfg, env_ids, device: int = 0, seed0: int = 0, horizon: int = 0, bank_episodes: int = 0, per_env_peg: bool = True):
        from .backend import HipBatch, HipModel

        self.flat, self.cfg = flat, cfg
        self.env_ids = np.asarray(env_ids, dtype=np.int64)
        self.B = len(self.env_ids)
        self.model = HipModel(flat)
        self.model.set_controller(cfg)
        self.model.set_task(peg_task(flat, cfg))
        self.spec = cfg.get("reset") or default_reset_spec()
        self.n_sub = int(cfg.get("env", {}).get("n_sub", 25))
        self._draw_fn = functools.partial(reset_draws, spec=self.spec)
        self.per_env_peg = per_env_peg
        self.batch = HipBatch(self.model, self.B, device, per_env_params=per_env_peg)
        self.seed0 = seed0
        self.horizon = horizon
        self.reset()
        if horizon:
            self.batch.set_episode(horizon)
        if bank_episodes:
            self.install_reset_bank(bank_episodes)   # qpos plus, with per-env pegs, the float-table slots that depend on the radius (reset_bank.py)

    def _bank_slots(self):
        if not hasattr(self, "_slots"):
            slots = []
            if self.per_env_peg:
                base = peg_model_rows(self.flat, [0.02])
                probe = peg_model_rows(self.flat, [0.0271])
                for k in base:
                    for e in np.nonzero(np.abs(probe[k][0] - base[k][0]) > 0)[0]:
                        off = self.batch.param_offset(k, int(e))
                        if off >= 0:
                            slots.append((k, int(e), off))
            self._slots = slots
        return self._slots

    def _bank_patch_offsets(self):
        return [o for _, _, o in self._bank_slots()]

    def _episode(self, idx, episode):
        d = self.episode_draws(idx, episode)
        return np.array([x["qpos"] for x in d]).reshape(-1, int(self.spec["nq"])), np.array([x["peg_radius"] for x in d])

    def _bank_rows(self, idx, episode):
        qpos, radii = self._episode(idx, episode)
        if not self._bank_slots():
            return qpos, np.zeros((len(idx), 0))
        rows = peg_model_rows(self.flat, radii)
        return qpos, np.stack([rows[k][:, e] for k, e, _ in self._bank_slots()], axis=1)

    def reset(self, block: int = 0):
        qpos, radii = self._episode(np.arange(self.B), block)
        b = self.batch
        if self.per_env_peg:
            for field, rows in peg_model_rows(self.flat, radii).items():
                b.param_set(field, rows)
        b.set("qpos", qpos); b.set("qvel", 0.0); b.set("ctrl", 0.0); b.set("time", 0.0); b.set("qacc_warmstart", 0.0)
        b.forward(); b.ctrl_reset()
        self.qpos0, self.radii = qpos, radii

    def step(self, actions, n_sub: int = 0):
        self.batch.control_step(actions, n_sub or self.n_sub)
        self._bank_tick()

    def obs(self):
        return self.batch.tensor("obs")

    def reward(self):
        return self.batch.tensor("reward")

    def success(self):
        return self.batch.tensor("success")


def peg_model_rows(flat, radii, density: float = 1000.0):
    """Model arrays that depend on the per-episode peg radius (two_arm_peg_in_hole.py:343-351), for n envs at once.

    The reference recompiles the MJCF on every hard reset.  The peg is a cylinder welded to the right hand, so the affected compiled fields are:
    geom_size / geom_rbound of the peg geom (+ its visual twin), body_mass / body_inertia of the peg body, body_subtreemass along its ancestors,
    and -- because the arm's mass matrix at qpos0 changes -- dof_invweight0 and body_invweight0 of everything in that kinematic tree
    (mj_setConst [3P]: diag / Jacobian traces of M^-1).  M(r) = M(r0) + C(r) - C(r0) with C the peg's projected spatial inertia, so all envs are
    handled with batched numpy (checked against a full recompile in tests/test_peg_host.py)."""
    from . import mjcf

    radii = np.asarray(radii, dtype=np.float64).ravel()
    n = len(radii)
    nv, nbody = flat.nv, flat.nbody
    pb = flat.name2id("body", "peg_main")
    g0, gv = flat.name2id("geom", "peg_g0"), flat.name2id("geom", "peg_g0_vis")
    r0, h = float(flat.geom_size[g0][0]), float(flat.geom_size[g0][1])
    M0, (xpos, xquat, xmat, xipos, ximat, xanchor, xaxis) = mjcf.mass_matrix_np(flat, flat.qpos0)

    def cyl(r):
        m = density * np.pi * r * r * 2 * h
        return m, np.stack([m * (3 * r * r + 4 * h * h) / 12.0, m * (3 * r * r + 4 * h * h) / 12.0, m * r * r / 2.0], axis=-1)

    jp, jr = mjcf.body_jacobian_np(flat, xpos, xmat, xanchor, xaxis, pb, xipos[pb])
    R = np.asarray(ximat[pb]).reshape(3, 3)

    def project(m, I):       # C = m jp^T jp + jr^T (R diag(I) R^T) jr, batched over envs
        Iw = np.einsum("ab,nb,cb->nac", R, I, R)
        return m[:, None, None] * (jp.T @ jp)[None] + np.einsum("ai,nab,bj->nij", jr, Iw, jr)

    m_new, I_new = cyl(radii)
    m_old, I_old = cyl(np.array([r0]))
    M = M0[None] + project(m_new, I_new) - project(m_old, I_old)
    Minv = np.linalg.inv(M)

    def tile(a):
        return np.repeat(np.asarray(a, dtype=np.float64)[None], n, axis=0).copy()

    geom_size, geom_rbound = tile(flat.geom_size), tile(flat.geom_rbound)
    for g in (g0, gv):
        geom_size[:, g, 0] = radii
        geom_rbound[:, g] = np.sqrt(radii**2 + h * h)
    body_mass, body_inertia, sub = tile(flat.body_mass), tile(flat.body_inertia), tile(flat.body_subtreemass)
    body_mass[:, pb] = m_new
    body_inertia[:, pb] = I_new
    b = pb
    while True:
        sub[:, b] += m_new - m_old[0]
        if b == 0:
            break
        b = int(flat.body_parentid[b])
    diw = np.einsum("nii->ni", Minv).copy()                       # every Baxter joint is a hinge: dof_invweight0 = diag(M^-1)
    for j in range(flat.njnt):
        if flat.jnt_type[j] not in (2, 3):
            raise NotImplementedError("peg_model_rows: only hinge / slide joints")
    biw = tile(flat.body_invweight0)
    for bb in range(1, nbody):
        if flat.body_weldid[bb] == 0:
            continue
        bjp, bjr = mjcf.body_jacobian_np(flat, xpos, xmat, xanchor, xaxis, bb, xipos[bb])
        biw[:, bb, 0] = np.einsum("ai,nij,aj->n", bjp, Minv, bjp) / 3.0
        biw[:, bb, 1] = np.einsum("ai,nij,aj->n", bjr, Minv, bjr) / 3.0
    return {"geom_size": geom_size.reshape(n, -1), "geom_rbound": geom_rbound.reshape(n, -1), "body_mass": body_mass.reshape(n, -1),
            "body_inertia": body_inertia.reshape(n, -1), "body_subtreemass": sub.reshape(n, -1), "body_invweight0": biw.reshape(n, -1),
            "dof_invweight0": diw.reshape(n, -1)}
