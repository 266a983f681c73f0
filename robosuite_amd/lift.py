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

import functools

import numpy as np

from .reset_bank import ResetBankMixin

PANDA_INIT_QPOS = np.array([0, np.pi / 16.0, 0.00, -np.pi / 2.0 - np.pi / 3.0, 0.00, np.pi - 0.2, np.pi / 4])  # panda_robot.py:37
PANDA_GRIPPER_INIT_QPOS = np.array([0.020833, -0.020833])  # panda_gripper.py:36
TABLE_OFFSET = np.array([0.0, 0.0, 0.8])  # lift.py:153
CUBE_DENSITY = 1000.0  # models/objects/generated_objects.py:680-699 (PrimitiveObject default)


def default_reset_spec():
    """The reset of `suite.make("Lift", robots="Panda")` with the reference's defaults, as the data `factory.reset_cfg` reads off a live env
    (cfg["reset"]): robot init_qpos + noise (robots/robot.py:107-113, 247-259), gripper init_qpos, cube size range (lift.py:311-318), the
    UniformRandomSampler of lift.py:321-333."""
    return dict(nq=16, arm_init_qpos=[float(x) for x in PANDA_INIT_QPOS], arm_qpos_idx=list(range(7)), noise=dict(type="gaussian", magnitude=0.02),
                grippers=[dict(init_qpos=[float(x) for x in PANDA_GRIPPER_INIT_QPOS], qpos_idx=[7, 8])],
                cube=dict(size_min=[0.020] * 3, size_max=[0.022] * 3, density=CUBE_DENSITY),
                sampler=dict(x_range=[-0.03, 0.03], y_range=[-0.03, 0.03], rotation=None, rotation_axis="z", z_offset=0.01,
                             reference_pos=[float(x) for x in TABLE_OFFSET], ensure_object_boundary_in_range=False, ensure_valid_placement=True,
                             objects=[dict(name="cube", horizontal_radius=None, bottom_z=None, top_z=None, qposadr=9, init_quat=None)]))


_PREPARED = {}          # id(spec) -> (spec, content key, arrays); at most _PREPARED_MAX entries, least recently used first out
_PREPARED_MAX = 32


def _spec_key(spec):
    """The values the prepared arrays are built from: an in-place edit of the spec after its first draw changes the key and the arrays are rebuilt."""
    c = spec.get("cube")
    return (tuple(spec["arm_init_qpos"]), None if c is None else (tuple(c["size_min"]), tuple(c["size_max"])))


def prepared(spec):
    """The spec's lists as numpy arrays, built once per spec object (the reset-ring upkeep draws thousands of episodes per second of rollout: no per-draw
    list -> array conversions).  Kept beside the spec, not inside it: the spec is part of a cfg that stays JSON-serialisable (factory.extract, tools/gen_golden.py
    dump it) and is shared between the envs built from it (round-4 advisor finding).  Bounded and content-checked (round-5 advisor finding): an entry holds its
    spec (so the id cannot be reused while it is cached), is rebuilt when the values it was built from have been edited in place, and the cache keeps the
    _PREPARED_MAX most recently used specs -- envs and specs built over and over (tests, per-reset rebuilds) no longer pin every spec for the life of the process."""
    k = id(spec)
    hit = _PREPARED.get(k)
    key = _spec_key(spec)
    if hit is None or hit[0] is not spec or hit[1] != key:
        p = dict(arm=np.array(spec["arm_init_qpos"], dtype=np.float64))
        if "cube" in spec:
            p["size_min"], p["size_max"] = np.array(spec["cube"]["size_min"], dtype=np.float64), np.array(spec["cube"]["size_max"], dtype=np.float64)
        hit = (spec, key, p)
    else:
        del _PREPARED[k]          # re-inserted below: dicts keep insertion order, the first key is the least recently used
    _PREPARED[k] = hit
    while len(_PREPARED) > _PREPARED_MAX:
        del _PREPARED[next(iter(_PREPARED))]
    return hit[2]


def arm_noise(rng: np.random.Generator, spec) -> np.ndarray:
    """Robot.reset's joint noise (robots/robot.py:247-259).  The draw is made even at magnitude 0 (`initialization_noise=None`), as there."""
    arm, noise = prepared(spec)["arm"], spec["noise"]
    if noise["type"] == "gaussian":
        z = rng.standard_normal(len(arm))
    elif noise["type"] == "uniform":
        z = rng.uniform(-1.0, 1.0, len(arm))
    else:
        raise ValueError("Error: Invalid noise type specified. Options are 'gaussian' or 'uniform'.")
    return arm + z * noise["magnitude"]


def sample_quat(rng: np.random.Generator, sampler) -> np.ndarray:
    """UniformRandomSampler._sample_quat (placement_samplers.py:191-219): rotation None = uniform yaw over the full circle, a pair = uniform inside it,
    a number = fixed (no draw)."""
    rot = sampler["rotation"]
    if rot is None:
        ang = rng.uniform(high=2 * np.pi, low=0)
    elif isinstance(rot, (list, tuple)):
        ang = rng.uniform(high=max(rot), low=min(rot))
    else:
        ang = rot
    k = {"x": 1, "y": 2, "z": 3}[sampler["rotation_axis"]]
    q = np.zeros(4)
    q[0], q[k] = np.cos(ang / 2), np.sin(ang / 2)
    return q


def quat_multiply_raw(quaternion1, quaternion0):
    """utils/transform_utils.py:67-93 on the raw 4-arrays it is handed.  The sampler passes its (w, x, y, z) draw and the object's init_quat to this
    (x, y, z, w) routine and uses the float32 result as (w, x, y, z) again (placement_samplers.py:287-289); restated as it is."""
    x0, y0, z0, w0 = quaternion0
    x1, y1, z1, w1 = quaternion1
    return np.array((x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0, -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0, x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
                     -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0), dtype=np.float32)


def sample_objects(rng: np.random.Generator, sampler, geometry):
    """UniformRandomSampler.sample (placement_samplers.py:221-309) over the sampler's objects in order.  geometry[i] = (horizontal_radius, bottom_z,
    top_z) of object i.  Returns [(pos3, quat wxyz)]."""
    ref = sampler["reference_pos"]
    placed, out = [], []
    for o, (radius, bottom, top) in zip(sampler["objects"], geometry):
        for _ in range(5000):
            lo, hi = sampler["x_range"]
            if sampler["ensure_object_boundary_in_range"]:
                lo, hi = lo + radius, hi - radius
            x = rng.uniform(high=hi, low=lo) + ref[0]
            lo, hi = sampler["y_range"]
            if sampler["ensure_object_boundary_in_range"]:
                lo, hi = lo + radius, hi - radius
            y = rng.uniform(high=hi, low=lo) + ref[1]
            z = sampler["z_offset"] + ref[2] - bottom
            ok = True
            if sampler["ensure_valid_placement"]:
                for (px, py, pz, pr, ptop) in placed:
                    if float(np.linalg.norm((x - px, y - py))) <= pr + radius and z - pz <= ptop - bottom:
                        ok = False
                        break
            if ok:
                quat = sample_quat(rng, sampler)
                if o.get("init_quat") is not None:
                    quat = quat_multiply_raw(quat, np.array(o["init_quat"]))
                placed.append((x, y, z, radius, top))
                out.append((np.array([x, y, z]), quat))
                break
        else:
            raise RuntimeError("Cannot place all objects ):")   # RandomizationError in the reference
    return out


def reset_draws(rng: np.random.Generator, spec=None, aux=None):
    """One hard-reset block of draws from the env's generator, in the reference's order: cube size (one U per axis, mjcf_utils.py:470-504), robot
    joint noise, then the placement sampler (x U, y U, rotation).  `spec` = cfg["reset"] (factory.reset_cfg); None = the Panda defaults.
    `aux`: the generator a sampler with `own_rng` draws from (a user's placement_initializer built without rng= owns one, placement_samplers.py:44-47)."""
    spec = default_reset_spec() if spec is None else spec
    p = prepared(spec)
    size = rng.uniform(p["size_min"], p["size_max"])  # BoxObject(size_min, size_max): one U per axis
    arm = arm_noise(rng, spec)
    srng = aux if (spec["sampler"].get("own_rng") and aux is not None) else rng
    sx, sy, sz = float(size[0]), float(size[1]), float(size[2])
    (pos, quat), = sample_objects(srng, spec["sampler"], [((sx * sx + sy * sy) ** 0.5, -sz, sz)])   # BoxObject: horizontal_radius (= |size[:2]|), bottom / top offsets
    return dict(size=size, arm=arm, pos=pos, quat=quat)


def reset_draws_fast(rng: np.random.Generator, spec):
    """reset_draws for the usual shape of the Lift reset (one object, the env's own generator, no init_quat), with the generator's scalar / vector
    calls written out: low + (high - low) * rng.random() is what Generator.uniform computes (numpy/random/_generator.pyx: loc + scale * next_double),
    without its argument checking -- bit for bit the same draws (tests/test_lift_host.py), a third of the time.  The reset ring draws ~8 episodes per
    control step of a 4096-env rollout, on the host, beside the control steps."""
    p = prepared(spec)
    lo, hi = p["size_min"], p["size_max"]
    size = lo + (hi - lo) * rng.random(3)
    noise, sm = spec["noise"], spec["sampler"]
    n = len(p["arm"])
    z = rng.standard_normal(n) if noise["type"] == "gaussian" else (-1.0 + 2.0 * rng.random(n))
    arm = p["arm"] + z * noise["magnitude"]
    sx, sy, sz = float(size[0]), float(size[1]), float(size[2])
    ref, (xlo, xhi), (ylo, yhi) = sm["reference_pos"], sm["x_range"], sm["y_range"]
    if sm["ensure_object_boundary_in_range"]:
        r = (sx * sx + sy * sy) ** 0.5
        xlo, xhi, ylo, yhi = xlo + r, xhi - r, ylo + r, yhi - r
    x = xlo + (xhi - xlo) * rng.random() + ref[0]
    y = ylo + (yhi - ylo) * rng.random() + ref[1]
    z0 = sm["z_offset"] + ref[2] - (-sz)
    rot = sm["rotation"]
    if rot is None:
        ang = 0 + (2 * np.pi - 0) * rng.random()
    elif isinstance(rot, (list, tuple)):
        ang = min(rot) + (max(rot) - min(rot)) * rng.random()
    else:
        ang = rot
    quat = np.zeros(4)
    quat[0], quat[{"x": 1, "y": 2, "z": 3}[sm["rotation_axis"]]] = np.cos(ang / 2), np.sin(ang / 2)
    return dict(size=size, arm=arm, pos=np.array([x, y, z0]), quat=quat)


def fast_path_ok(spec) -> bool:
    sm = spec["sampler"]
    return len(sm["objects"]) == 1 and not sm.get("own_rng") and sm["objects"][0].get("init_quat") is None and spec["noise"]["type"] in ("gaussian", "uniform")


def initial_qpos(draw, spec=None) -> np.ndarray:
    """qpos after Robot.reset + placement (lift.py:401-415): arm joints, gripper init_qpos, the cube's free joint (pos, quat wxyz)."""
    spec = default_reset_spec() if spec is None else spec
    q = np.zeros(int(spec["nq"]))
    q[spec["arm_qpos_idx"]] = draw["arm"]
    for g in spec["grippers"]:
        q[g["qpos_idx"]] = g["init_qpos"]
    a = int(spec["sampler"]["objects"][0]["qposadr"])
    q[a:a + 3] = draw["pos"]
    q[a + 3:a + 7] = draw["quat"]
    return q


def cube_model_entries(flat, sizes: np.ndarray, density: float = CUBE_DENSITY):
    """The compiled-model ENTRIES that depend on the cube half-sizes, for n envs at once: {field: (flat element indices, values [n, len(indices)])} -- the
    closed forms of cube_model_rows without tiling the whole arrays (the reset-ring upkeep needs these few numbers per episode, nothing else)."""
    sizes = np.asarray(sizes, dtype=np.float64).reshape(-1, 3)
    cb, g0, gv = flat.name2id("body", "cube_main"), flat.name2id("geom", "cube_g0"), flat.name2id("geom", "cube_g0_vis")
    d0 = int(flat.jnt_dofadr[int(flat.body_jntadr[cb])])
    sx, sy, sz = sizes[:, 0], sizes[:, 1], sizes[:, 2]
    mass = density * 8.0 * sx * sy * sz
    inertia = np.stack([mass / 3.0 * (sy**2 + sz**2), mass / 3.0 * (sx**2 + sz**2), mass / 3.0 * (sx**2 + sy**2)], axis=1)
    rb = np.linalg.norm(sizes, axis=1)
    minv, iinv = 1.0 / mass, np.mean(1.0 / inertia, axis=1)
    sub_world, old_mass = float(np.asarray(flat.body_subtreemass).ravel()[0]), float(np.asarray(flat.body_mass).ravel()[cb])
    return {
        "geom_size": (np.array([3 * g0, 3 * g0 + 1, 3 * g0 + 2, 3 * gv, 3 * gv + 1, 3 * gv + 2]), np.concatenate([sizes, sizes], axis=1)),
        "geom_rbound": (np.array([g0, gv]), np.stack([rb, rb], axis=1)),
        "body_mass": (np.array([cb]), mass[:, None]),
        "body_inertia": (np.array([3 * cb, 3 * cb + 1, 3 * cb + 2]), inertia),
        "body_subtreemass": (np.array([0, cb]), np.stack([sub_world + (mass - old_mass), mass], axis=1)),
        "body_invweight0": (np.array([2 * cb, 2 * cb + 1]), np.stack([minv, iinv], axis=1)),
        "dof_invweight0": (np.arange(d0, d0 + 6), np.stack([minv, minv, minv, iinv, iinv, iinv], axis=1)),
    }


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


def episode_setup(seed0: int, env_ids, block: int = 0, spec=None):
    """Draws for the global env ids `env_ids`: env i uses default_rng(seed0 + i); `block` selects which
    hard-reset block of that generator (0 = the state after make(), 1 = after the first user reset(), ...)."""
    sizes, qpos = [], []
    for i in env_ids:
        rng = np.random.default_rng(seed0 + int(i))
        for _ in range(block + 1):
            d = reset_draws(rng, spec)
        sizes.append(d["size"])
        qpos.append(initial_qpos(d, spec))
    return np.array(sizes), np.array(qpos)


def env_actions(env_ids, n_steps: int, scale: float = 1.0, action_dim: int = 7):
    """Per-env action streams a_t ~ scale * U(-1,1)^7 from default_rng(10**6 + i); returns [n_steps, n, action_dim] float32."""
    out = np.empty((n_steps, len(env_ids), action_dim), dtype=np.float32)
    for k, i in enumerate(env_ids):
        out[:, k, :] = scale * np.random.default_rng(10**6 + int(i)).uniform(-1, 1, (n_steps, action_dim))
    return out


REWARD_NORM = {"lift": 2.25, "stack": 2.0, "peg_in_hole": 5.0}   # lift.py:270-271, stack.py:263-264, two_arm_peg_in_hole.py:287-288: reward *= reward_scale / norm


def task_env_args(cfg, task: str, reward_scale=None, reward_shaping=None):
    """(reward_scale, reward_shaping) of the on-device epilogue: explicit arguments, else cfg["env"] (what the reference constructor was given,
    factory.env_cfg), else the benchmark's (1.0, dense).  reward_scale = None in the reference means "no normalisation" (the raw sum): the epilogue
    multiplies by reward_scale / norm, so it is handed the norm."""
    e = cfg.get("env", {})
    shaping = bool(e.get("reward_shaping", True)) if reward_shaping is None else bool(reward_shaping)
    scale = e.get("reward_scale", 1.0) if reward_scale is None and "reward_scale" in e else (1.0 if reward_scale is None else reward_scale)
    if scale is None:
        scale = REWARD_NORM[task]
    return float(scale), shaping


def robot_obs_program(cfg, site, eef_body):
    """Observation entries of the single-arm robot keys in `_get_observations` order (robots/robot.py:334-484)."""
    qi, di = cfg["qpos_idx"], cfg["dof_idx"]
    gq, gd = cfg["grip_qpos_idx"], cfg["grip_dof_idx"]
    obs = []
    obs += [("qpos", q, 0) for q in qi] + [("cos", q, 0) for q in qi] + [("sin", q, 0) for q in qi]
    obs += [("qvel", d, 0) for d in di] + [("qacc", d, 0) for d in di]
    obs += [("site_pos", site, k) for k in range(3)] + [("body_quat", eef_body, k) for k in range(4)] + [("site_quat", site, k) for k in range(4)]
    obs += [("qpos", q, 0) for q in gq] + [("qvel", d, 0) for d in gd]
    return obs


def grasp_groups(flat, cfg):
    """(eef body id, left pad geom ids, right pad geom ids) of ManipulationEnv._check_grasp: from cfg["grasp"] (factory.grasp_cfg: the gripper's
    important_geoms), else the Panda gripper's names."""
    names = flat.names
    g = cfg.get("grasp") or dict(left_pad=["gripper0_right_finger1_pad_collision"], right_pad=["gripper0_right_finger2_pad_collision"], eef_body="robot0_right_hand")
    geom = names["geom"]
    return names["body"].index(g["eef_body"]), [geom.index(n) for n in g["left_pad"]], [geom.index(n) for n in g["right_pad"]]


def lift_task(flat, cfg, reward_scale=None, reward_shaping=None):
    """Observation program + reward description of Lift for the on-device epilogue (include/rsim.h rsim_task_desc).

    Key order = the reference's `_get_observations` order for use_camera_obs=False:
    robot0_joint_pos, _cos, _sin, joint_vel, joint_acc, eef_pos, eef_quat (BODY robot0_right_hand, xyzw), eef_quat_site, gripper_qpos,
    gripper_qvel (robots/robot.py:334-484), and with use_object_obs cube_pos, cube_quat, gripper_to_cube_pos (lift.py:356-399).  Reward flavour and
    scale: cfg["env"] (what the reference's constructor was given; lift.py:158-159, 256-271)."""
    names = flat.names
    site = int(cfg["eef_site"])                      # gripper0_right_grip_site
    eef_body, lpad, rpad = grasp_groups(flat, cfg)
    cube_body = names["body"].index("cube_main")
    obs = robot_obs_program(cfg, site, eef_body)
    if "cube_pos" in cfg.get("obs_keys", ["cube_pos"]):       # use_object_obs (lift.py:356)
        obs += [("body_pos", cube_body, k) for k in range(3)] + [("body_quat", cube_body, k) for k in range(4)]
        obs += [("body_minus_site", cube_body, k | (site << 2)) for k in range(3)]
    if sum(cfg.get("obs_dims", [len(obs)])) != len(obs):
        raise NotImplementedError(f"Lift observation record: the reference env returned {sum(cfg['obs_dims'])} floats under keys {cfg['obs_keys']}, the on-device program has {len(obs)}")
    scale, shaping = task_env_args(cfg, "lift", reward_scale, reward_shaping)
    return dict(obs=obs, task="lift", object_body=cube_body, grip_site=site, table_height=float(cfg.get("table_height", TABLE_OFFSET[2])), lift_margin=0.04,
                reward_scale=scale, reward_shaping=shaping, left_pad_geoms=lpad, right_pad_geoms=rpad, object_geoms=[names["geom"].index("cube_g0")])


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
        self.spec = cfg.get("reset") or default_reset_spec()     # cfg["reset"]: the reference env's own reset configuration (factory.reset_cfg)
        self.n_sub = int(cfg.get("env", {}).get("n_sub", 25))    # control_timestep / model_timestep (base.py:212-218)
        self._draw_fn = functools.partial(reset_draws_fast if fast_path_ok(self.spec) else reset_draws, spec=self.spec)
        self._draw_aux = bool(self.spec["sampler"].get("own_rng"))
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
                dens = float(self.spec["cube"]["density"])
                base = {k: np.asarray(self.flat.arrays[k], dtype=np.float64).ravel() for k in cube_model_rows(self.flat, self.sizes[:1], dens)}
                probe = cube_model_rows(self.flat, np.array([[0.0201, 0.0213, 0.0207]]), dens)
                for k, rows in probe.items():
                    for e in np.nonzero(np.abs(rows[0] - base[k]) > 0)[0]:
                        off = self.batch.param_offset(k, int(e))
                        if off >= 0:
                            slots.append((k, int(e), off))
            self._slots = slots
        return self._slots

    def _bank_patch_offsets(self):
        return [o for _, _, o in self._bank_slots()]

    def _episode(self, idx, episode):
        """(cube sizes, qpos) of episode `episode` for the LOCAL env indices idx; equals episode_setup(seed0, env_ids[idx], episode, spec), one block per call."""
        d = self.episode_draws(idx, episode)
        return np.array([x["size"] for x in d]).reshape(-1, 3), np.array([initial_qpos(x, self.spec) for x in d]).reshape(-1, int(self.spec["nq"]))

    def _bank_rows(self, idx, episode):
        sizes, qpos = self._episode(idx, episode)
        slots = self._bank_slots()
        if not slots:
            return qpos, np.zeros((len(idx), 0))
        ent = cube_model_entries(self.flat, sizes, float(self.spec["cube"]["density"]))
        if not hasattr(self, "_slot_src"):      # slot -> (field, column of that field's entry values)
            self._slot_src = [(k, int(np.nonzero(ent[k][0] == e)[0][0])) for k, e, _ in slots]
        return qpos, np.stack([ent[k][1][:, c] for k, c in self._slot_src], axis=1)

    def reset(self, block: int = 0):
        sizes, qpos = self._episode(np.arange(self.B), block)
        b = self.batch
        if self.per_env_cube:
            for field, rows in cube_model_rows(self.flat, sizes, float(self.spec["cube"]["density"])).items():
                b.param_set(field, rows)
        b.set("qpos", qpos)
        b.set("qvel", 0.0)
        b.set("ctrl", 0.0)
        b.set("time", 0.0)
        b.set("qacc_warmstart", 0.0)
        b.forward()                                   # MujocoEnv.reset: sim.forward() (base.py:298-303); warm start stays 0 (fresh MjData)
        b.ctrl_reset()                                # fresh controller objects per reset (robots/robot.py:271)
        self.sizes, self.qpos0 = sizes, qpos

    def step(self, actions, n_sub: int = 0):
        """One env.step for every env: fused physics + controllers + observation / reward epilogue (results stay on the device).  n_sub = 0: the
        env's own control_timestep / model_timestep (cfg["env"]["n_sub"]; 25 at the default control_freq of 20 Hz)."""
        self.batch.control_step(actions, n_sub or self.n_sub)
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
