"""PickPlace / IIWA + Robotiq140 (BASELINE configs[4]) host side for the fused kernel's 64-body x 48-dof configuration: the task program.

  robot keys   robots/robot.py:334-484 (one arm; the Robotiq gripper has six finger joints)
  object keys  pick_place.py:585-668, per object: {obj}_to_robot0_eef_pos, {obj}_to_robot0_eef_quat, {obj}_pos, {obj}_quat.  The relative
               sensors precede the object's own pos / quat sensors in the Observable order and read them from the observation cache, so they
               see the object pose of the PREVIOUS control step against the current gripper pose (and zeros in the record reset() returns);
               the kernel reproduces that (include/rsim.h RSIM_OBS_REL_POS).
  reward       pick_place.py:274-429, success :737-762
  single-object mode 2 (PickPlaceMilk / Bread / Cereal / Can, pick_place.py:800-847; cfg["task"]["single_object_mode"] = 2, ["object_id"]): the
               observation record holds the chosen object's sensors only (:612-624), reset() moves the other three objects out of the scene
               (clear_objects, base.py:591-602: free joint at (10, 10, 10), from where they fall for the rest of the episode -- they stay in the
               reward's sums exactly as in the reference), the reward is not divided by 4 and success = the object is in its bin.
  single-object mode 1 (PickPlaceSingle, pick_place.py:717-722, 800-807): as mode 2, but every reset draws the object -- rng.choice over a Python SET
               of the object names, i.e. one rng.integers(0, 4) whose meaning depends on the process's string-hash order.  cfg["task"]["mode1_order"]
               (object indices in the order the reference process iterated the set; tools/gen_golden.py records it under PYTHONHASHSEED=0) fixes
               that meaning, identity if absent.  The observation record holds the episode's object (the kernel selects it per env:
               include/rsim.h RSIM_TASK_OBJECT, observation entries with a = -1) followed by the `obj_id` observable (:626-635).
The visual-object bodies (pick_place.py:455-513, 703-706: static bodies without collision geoms) take no part in the dynamics and are not restated.
"""
from __future__ import annotations


def pick_place_task(flat, cfg, reward_scale=None, reward_shaping=None):
    names, t = flat.names, cfg["task"]
    e = cfg.get("env", {})      # what the reference constructor was given (pick_place.py:177-178, 305-310); absent: the benchmark's (1.0, dense)
    reward_shaping = bool(e.get("reward_shaping", True)) if reward_shaping is None else bool(reward_shaping)
    if reward_scale is None:
        reward_scale = e.get("reward_scale", 1.0)
        if reward_scale is None:   # reward_scale=None: the raw sum, not divided by four either (pick_place.py:307-310); the epilogue divides by 4 in the all-objects mode
            reward_scale = 1.0 if int(t.get("single_object_mode", 0)) else 4.0
    body, geom, site = names["body"], names["geom"], names["site"]
    gsite = site.index(t["grip_site"])
    eef_body = body.index(t["eef_body"])
    qi, di = cfg["qpos_idx"], cfg["dof_idx"]
    gq, gd = cfg["grip_qpos_idx"], cfg["grip_dof_idx"]
    obs = []
    obs += [("qpos", q, 0) for q in qi] + [("cos", q, 0) for q in qi] + [("sin", q, 0) for q in qi]
    obs += [("qvel", d, 0) for d in di] + [("qacc", d, 0) for d in di]
    obs += [("site_pos", gsite, k) for k in range(3)] + [("body_quat", eef_body, k) for k in range(4)] + [("site_quat", gsite, k) for k in range(4)]
    obs += [("qpos", q, 0) for q in gq] + [("qvel", d, 0) for d in gd]
    obj_body = [body.index(b) for b in t["object_bodies"]]
    single, only = int(t.get("single_object_mode", 0)), int(t.get("object_id", -1))
    pos_slot = []
    if single == 1:                   # the episode's object, whichever it is (a = -1), then its id
        obs += [("rel_pos", -1, k) for k in range(3)] + [("rel_quat", -1, k) for k in range(4)]
        pos_slot = [len(obs)] * len(obj_body)
        obs += [("body_pos", -1, k) for k in range(3)] + [("body_quat", -1, k) for k in range(4)] + [("task_object", 0, 0)]
    for i, ob in enumerate(obj_body if single != 1 else []):
        if single and i != only:      # pick_place.py:616-624: the other objects' sensors are disabled
            pos_slot.append(0)
            continue
        obs += [("rel_pos", i, k) for k in range(3)] + [("rel_quat", i, k) for k in range(4)]
        pos_slot.append(len(obs))
        obs += [("body_pos", ob, k) for k in range(3)] + [("body_quat", ob, k) for k in range(4)]
    return dict(obs=obs, task="pick_place", grip_site=gsite, eef_body=eef_body, obj_body=obj_body, pos_slot=pos_slot, single_object_mode=single,
                obj_geoms=[[geom.index(g) for g in gs] for gs in t["object_geoms"]], bin2_pos=t["bin2_pos"], bin_size=t["bin_size"][:2],
                bin_target=[r[:2] for r in t["target_bin_placements"]], left_pad_geoms=[geom.index(g) for g in t["left_pad"]],
                right_pad_geoms=[geom.index(g) for g in t["right_pad"]], reward_scale=reward_scale, reward_shaping=reward_shaping)


import functools  # noqa: E402

import numpy as np  # noqa: E402

from .reset_bank import ResetBankMixin  # noqa: E402


def reset_draws(rng: np.random.Generator, placement: dict, choose: bool = False):
    """One hard-reset block of the env generator in the reference's order (pick_place.py:670-700 -> robots/robot.py:247-259,
    utils/placement_samplers.py:221-309, 383-470): arm noise N(0,1) x 7 x 0.02; CollisionObjectSampler over the objects in list order
    (x, y uniform inside bin 1 shrunk by the object's horizontal radius, redrawn while it overlaps an already placed object, then a uniform yaw);
    then one x and one y draw over a zero-width range per visual object (their rotation is fixed); `choose` (single-object mode 1): then the
    object of the episode, rng.choice over the four names = one rng.integers(0, 4) (pick_place.py:717-718)."""
    noise = placement.get("noise") or dict(type="gaussian", magnitude=0.02)     # robots/robot.py:107-113 (`initialization_noise`), 247-259
    if noise["type"] == "gaussian":
        z = rng.standard_normal(len(placement["arm_init_qpos"]))
    elif noise["type"] == "uniform":
        z = rng.uniform(-1.0, 1.0, len(placement["arm_init_qpos"]))
    else:
        raise ValueError("Error: Invalid noise type specified. Options are 'gaussian' or 'uniform'.")
    arm = np.array(placement["arm_init_qpos"]) + z * noise["magnitude"]
    bx, by, bz = placement["bin1_pos"]
    xh, yh = placement["x_half"], placement["y_half"]
    placed, out = [], []
    for o in placement["objects"]:
        r, bot = o["horizontal_radius"], o["bottom_z"]
        for _ in range(5000):
            # low + (high - low) * rng.random() is Generator.uniform(low, high) without its argument checks: the same double, a third of the time
            x = (-xh + r) + ((xh - r) - (-xh + r)) * rng.random() + bx
            y = (-yh + r) + ((yh - r) - (-yh + r)) * rng.random() + by
            z = placement["z_offset"] + bz - bot
            ok = True
            for (px, py, pz, pr, ptop) in placed:
                if np.linalg.norm((x - px, y - py)) <= pr + r and z - pz <= ptop - bot:
                    ok = False
                    break
            if ok:
                yaw = 0.0 + (2.0 * np.pi - 0.0) * rng.random()
                placed.append((x, y, z, r, o["top_z"]))
                out.append((np.array([x, y, z]), yaw))
                break
        else:
            raise RuntimeError("Cannot place all objects")
    for _ in placement["objects"]:        # the four visual twins: x and y over [c, c]
        rng.random(); rng.random()
    return dict(arm=arm, objects=out, pick=int(rng.integers(0, len(placement["objects"]))) if choose else -1)


def active_object(task: dict, draw) -> int:
    """Object index the episode uses: all (-1), the drawn one (mode 1, through the recorded set order), the fixed one (mode 2)."""
    mode = int(task.get("single_object_mode", 0))
    if mode == 1:
        order = task.get("mode1_order") or list(range(len(task["placement"]["objects"])))
        return int(order[draw["pick"]])
    return int(task.get("object_id", -1)) if mode == 2 else -1


def initial_qpos(draw, placement: dict, nq: int, only: int = -1) -> np.ndarray:
    """only >= 0 (single-object mode 2): every object is placed by the sampler as usual, then all but object `only` are moved out of the scene
    (pick_place.py:723-727 -> base.py:591-602)."""
    q = np.zeros(nq)
    q[placement["arm_qpos_idx"]] = draw["arm"]
    q[placement["gripper_qpos_idx"]] = placement["gripper_init_qpos"]
    for i, (o, (pos, yaw)) in enumerate(zip(placement["objects"], draw["objects"])):
        a = o["qposadr"]
        if only >= 0 and i != only:
            q[a:a + 7] = (10.0, 10.0, 10.0, 1.0, 0.0, 0.0, 0.0)
            continue
        q[a:a + 3] = pos
        q[a + 3], q[a + 6] = np.cos(yaw / 2.0), np.sin(yaw / 2.0)
    return q


def episode_setup(cfg, nq: int, seed0: int, env_ids, block: int = 0):
    t = cfg["task"]
    pl = t["placement"]
    out = []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng, pl, int(t.get("single_object_mode", 0)) == 1)
        out.append(initial_qpos(d, pl, nq, active_object(t, d)))
    return np.array(out)


class PickPlaceBatch(ResetBankMixin):
    """B PickPlace/IIWA+Robotiq140 environments on one GPU (64-body x 48-dof kernel configuration).  `env_ids` are GLOBAL indices."""

    def __init__(self, flat, cfg, env_ids, device: int = 0, seed0: int = 0, horizon: int = 0, bank_episodes: int = 0, per_env_params: bool = False):
        from .backend import HipBatch, HipModel

        self.flat, self.cfg = flat, cfg
        self.env_ids = np.asarray(env_ids, dtype=np.int64)
        self.B = len(self.env_ids)
        self.model = HipModel(flat)
        self.model.set_controller(cfg)
        self.model.set_task(pick_place_task(flat, cfg))
        self.n_sub = int(cfg.get("env", {}).get("n_sub", 25))
        # no bound method here: env -> EpisodeStreams -> bound method -> env would be a reference cycle that keeps the batch's device memory (and the
        # upkeep thread) alive until the cyclic collector runs
        self._draw_fn = functools.partial(reset_draws, placement=cfg["task"]["placement"], choose=int(cfg["task"].get("single_object_mode", 0)) == 1)
        self.batch = HipBatch(self.model, self.B, device, per_env_params=per_env_params)   # per-env float tables: dynamics randomisation
        self.seed0 = seed0
        self.horizon = horizon
        self.reset()
        if horizon:
            self.batch.set_episode(horizon)
        if bank_episodes:
            self.install_reset_bank(bank_episodes)

    @property
    def _mode1(self):
        return int(self.cfg["task"].get("single_object_mode", 0)) == 1

    def _bank_patch_offsets(self):
        return [-1] if self._mode1 else []        # RSIM_PATCH_TASK_OBJECT: the row's extra column is the episode's object

    def _episode(self, idx, episode):
        """(qpos rows, object of the episode per env: -1 outside the single-object modes)"""
        t = self.cfg["task"]
        draws = self.episode_draws(idx, episode)
        obj = np.array([active_object(t, d) for d in draws], dtype=np.int64)
        return np.array([initial_qpos(d, t["placement"], self.flat.nq, int(o)) for d, o in zip(draws, obj)]).reshape(-1, self.flat.nq), obj

    def _bank_rows(self, idx, episode):
        q, obj = self._episode(idx, episode)
        return q, (obj[:, None].astype(np.float64) if self._mode1 else np.zeros((len(idx), 0)))

    def reset(self, block: int = 0):
        qpos, obj = self._episode(np.arange(self.B), block)
        b = self.batch
        if self._mode1:
            b.set("task_object", obj)
        b.set("qpos", qpos); b.set("qvel", 0.0); b.set("ctrl", 0.0); b.set("time", 0.0); b.set("qacc_warmstart", 0.0)
        b.forward(); b.ctrl_reset()
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
