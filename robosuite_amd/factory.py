"""`robosuite_amd.make(...)` (this module: robosuite_amd/factory.py): the `suite.make()`-shaped entry to the batched path (reference: environments/base.py:23-42).

    import robosuite_amd
    env = robosuite_amd.make("Stack", robots="Panda", n_envs=4096)                      # a VecEnv on cuda:0
    env = robosuite_amd.make("TwoArmPegInHole", robots="Baxter", n_envs=2048, controller_configs=cc,
                             env_configuration="single-robot", gripper_types=None)      # kwargs go to the reference's constructor

What the fused kernel needs from a robosuite env is (a) the compiled model and (b) a handful of index tables and gains that live on the
reference's Python objects (controller parts, gripper, task constants).  Two sources:

  * the reference itself, when `robosuite` is importable (or `reference_path` / $ROBOSUITE_PATH points at a checkout): the unmodified env class is
    constructed over this package's `mujoco`-shaped shim -- its own model assembly (models/, L0), its own composite-controller factory
    (controllers/composite/composite_controller_factory.py:73-138) and its own reset produce the MJCF string and the controller objects -- and
    `extract()` reads (flat model, cfg) off the live objects.  The shim runs on `KinematicsBackend` for this: host-only frames, Jacobians and mass
    matrix, enough for constructors that call `sim.forward()` and `Controller.update()`; it cannot step (the hot path has no CPU fallback);
  * the assets shipped in `robosuite_amd/assets/` for the four BASELINE configurations, produced by exactly that path (`python -m
    robosuite_amd.factory --ship`), so that `make()` works on a machine without the reference checkout (the GPU box).

`tools/gen_golden.py` uses the same `extract` helpers when it records fixtures, so a `make()`-built env and a fixture-built one agree bitwise.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

from . import mjcf

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

# (env name, robot, arm part-controller type) -> asset stem.  The controller type of a default config is the robot's own default
# (controllers/config/robots/default_<robot>.json: OSC_POSE for the arms used here).
SHIPPED = {
    ("Lift", "Panda", "OSC_POSE"): "lift_panda",
    ("Stack", "Panda", "OSC_POSE"): "stack_panda",
    ("TwoArmPegInHole", "Baxter", "JOINT_VELOCITY"): "peg_baxter_joint_velocity",
    ("PickPlace", "IIWA", "OSC_POSE"): "pickplace_iiwa",
}
# constructor kwargs the shipped configurations were built with, beyond the benchmark defaults below
SHIPPED_KWARGS = {"peg_baxter_joint_velocity": dict(env_configuration="single-robot", gripper_types=None)}
BENCH_KWARGS = dict(has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, use_object_obs=True, reward_shaping=True, control_freq=20,
                    horizon=500, ignore_done=True)

# format_action direction tables (panda_gripper.py:55-57, robotiq_140_gripper.py:66-68, robotiq_85_gripper.py:65-67)
GRIPPER_SIGNS = {"PandaGripper": [-1.0, 1.0], "Robotiq140Gripper": [1.0, -1.0], "Robotiq85Gripper": [1.0, 1.0],
                 "JacoThreeFingerGripper": [-1.0, -1.0, -1.0]}   # jaco_three_finger_gripper.py:57-71: current_action - speed * sign(action)


def gripper_signs(gripper):
    """Direction table of `gripper.format_action` (models/grippers/*.py: current_action = clip(current_action + d * speed * sign(action), -1, 1), the
    array handed to the actuators): known classes from the table above, any other single-action gripper of that family by PROBING its format_action
    with +1 from rest -- the returned vector is speed * d.  Grippers whose format_action is not of that form (the dexterous hands) are refused."""
    name = type(gripper).__name__
    if name in GRIPPER_SIGNS:
        return list(GRIPPER_SIGNS[name])
    if int(getattr(gripper, "dof", 0)) != 1 or not hasattr(gripper, "speed"):
        raise NotImplementedError(f"gripper {name}: only single-action grippers with the clip(current + d * speed * sign(a)) law have an in-kernel GRIP part")
    saved = np.array(gripper.current_action, dtype=np.float64).copy()
    try:
        gripper.current_action = np.zeros_like(saved)
        plus = np.array(gripper.format_action(np.array([1.0])), dtype=np.float64)
        gripper.current_action = np.zeros_like(saved)
        minus = np.array(gripper.format_action(np.array([-1.0])), dtype=np.float64)
    finally:
        gripper.current_action = saved
    d = plus / float(gripper.speed)
    if not (np.allclose(np.abs(d), 1.0) and np.allclose(minus, -plus)):
        raise NotImplementedError(f"gripper {name}: format_action is not clip(current + d * speed * sign(a)) (probe gave {plus}, {minus})")
    return [float(x) for x in np.round(d)]


# ------------------------------------------------------------------------------------------------------------------------------------------
# shim backend for model / configuration extraction
# ------------------------------------------------------------------------------------------------------------------------------------------
class KinematicsBackend:
    """robosuite_amd.shim backend without dynamics: frames, site / body Jacobians and the joint-space inertia from plain numpy
    (mjcf.kinematics_np / body_jacobian_np / mass_matrix_np, the compile-time helpers).  The reference's constructors and reset path call
    `sim.forward()` and `Controller.update()` (controllers/parts/controller.py:129-132, 199-232); they get consistent values here.  Stepping raises:
    this is how `make()` reads a model and its controller configuration off the reference's objects, not a way to simulate."""

    def __init__(self, flat: mjcf.FlatModel):
        self.flat = m = flat
        f64 = lambda n: np.zeros(int(n), dtype=np.float64)   # noqa: E731
        nmocap = int(m.arrays["nmocap"][0]) if "nmocap" in m.arrays else 0
        self.d = {"qpos": np.array(m.qpos0, dtype=np.float64).ravel().copy(), "qvel": f64(m.nv), "qacc": f64(m.nv), "qacc_warmstart": f64(m.nv), "ctrl": f64(m.nu),
                  "qfrc_applied": f64(m.nv), "mocap_pos": f64(3 * nmocap), "mocap_quat": f64(4 * nmocap), "xpos": f64(3 * m.nbody), "xquat": f64(4 * m.nbody),
                  "xmat": f64(9 * m.nbody), "xipos": f64(3 * m.nbody), "ximat": f64(9 * m.nbody), "geom_xpos": f64(3 * m.ngeom), "geom_xmat": f64(9 * m.ngeom),
                  "site_xpos": f64(3 * m.nsite), "site_xmat": f64(9 * m.nsite), "subtree_com": f64(3 * m.nbody), "qM": f64(m.nv * m.nv), "qfrc_bias": f64(m.nv),
                  "qfrc_passive": f64(m.nv), "qfrc_actuator": f64(m.nv), "qfrc_constraint": f64(m.nv), "actuator_force": f64(m.nu),
                  "sensordata": f64(m.arrays["sensor_dim"].sum() if m.nsensor else 0), "time": f64(1)}
        self._kin = None

    def model_array(self, name):
        return None

    def sync_model(self):
        pass

    def data_array(self, name):
        return self.d[name]

    def forward(self):
        m, d = self.flat, self.d
        M, kin = mjcf.mass_matrix_np(m, d["qpos"])
        xpos, xquat, xmat, xipos, ximat, xanchor, xaxis = self._kin = kin
        d["xpos"][:], d["xquat"][:], d["xmat"][:], d["xipos"][:], d["ximat"][:] = xpos.ravel(), xquat.ravel(), xmat.reshape(-1), xipos.ravel(), ximat.reshape(-1)
        gb, sb = m.geom_bodyid, m.site_bodyid
        d["geom_xpos"][:] = (xpos[gb] + np.einsum("gij,gj->gi", xmat[gb], m.geom_pos)).ravel()
        d["geom_xmat"][:] = np.stack([mjcf.quat2mat(mjcf.quat_mul(xquat[gb[i]], m.geom_quat[i])) for i in range(m.ngeom)]).reshape(-1)
        if m.nsite:
            d["site_xpos"][:] = (xpos[sb] + np.einsum("sij,sj->si", xmat[sb], m.site_pos)).ravel()
            d["site_xmat"][:] = np.stack([mjcf.quat2mat(mjcf.quat_mul(xquat[sb[i]], m.site_quat[i])) for i in range(m.nsite)]).reshape(-1)
        d["qM"][:] = M.ravel()

    def _no(self, *a, **k):
        raise RuntimeError("KinematicsBackend cannot step: it exists to read a model and its controller configuration off the reference's objects "
                           "(robosuite_amd.make); simulate with the VecEnv make() returns, or over robosuite_amd.hip_shim_backend.HipShimBackend")

    step = step1 = step2 = _no

    def reset(self):
        d = self.d
        d["qpos"][:] = np.asarray(self.flat.qpos0).ravel()
        for k in ("qvel", "qacc", "qacc_warmstart", "ctrl", "qfrc_applied", "time"):
            d[k][:] = 0

    def jac(self, kind, idx):
        if self._kin is None:
            self.forward()
        m = self.flat
        xpos, xquat, xmat, xipos, ximat, xanchor, xaxis = self._kin
        if kind == "site":
            body, point = int(m.site_bodyid[idx]), self.d["site_xpos"].reshape(-1, 3)[idx]
        elif kind == "body":
            body, point = int(idx), xpos[idx]
        else:
            body, point = int(m.geom_bodyid[idx]), self.d["geom_xpos"].reshape(-1, 3)[idx]
        return mjcf.body_jacobian_np(m, xpos, xmat, xanchor, xaxis, body, np.asarray(point))

    def full_M(self):
        return self.d["qM"].reshape(self.flat.nv, self.flat.nv).copy()

    ncon = 0

    def contacts(self):
        return []


# ------------------------------------------------------------------------------------------------------------------------------------------
# (flat model, cfg) from a live reference env
# ------------------------------------------------------------------------------------------------------------------------------------------
def controller_cfg(env):
    """Index tables / gains of the default single-arm configuration (OSC_POSE arm part + GRIP gripper part) from the reference objects."""
    robot = env.robots[0]
    osc = robot.part_controllers["right"]
    sim = env.sim
    return dict(
        qpos_idx=[int(i) for i in osc.qpos_index], dof_idx=[int(i) for i in osc.qvel_index],
        act_idx=[int(i) for i in robot._ref_actuators_indexes_dict["right"]],
        eef_site=int(sim.model.site_name2id(osc.ref_name)),
        base_site=int(sim.model.site_name2id(f"{osc.naming_prefix}{osc.part_name}_center")),
        kp=[float(x) for x in osc.kp], damping_ratio=1.0,
        input_min=[float(x) for x in osc.input_min], input_max=[float(x) for x in osc.input_max],
        output_min=[float(x) for x in osc.output_min], output_max=[float(x) for x in osc.output_max],
        uncouple=int(osc.uncoupling),
        grip_act=[int(i) for i in robot._ref_actuators_indexes_dict["right_gripper"]],
        grip_sign=[-1.0, 1.0], grip_speed=float(robot.gripper["right"].speed),
        grip_qpos_idx=[int(i) for i in robot._ref_gripper_joint_pos_indexes["right"]],
        grip_dof_idx=[int(i) for i in robot._ref_gripper_joint_vel_indexes["right"]],
    )


def controller_cfg_generic(env, ctype):
    """The same for any in-kernel arm part type (backend.CTRL_TYPES) on a single-arm robot with a gripper."""
    robot = env.robots[0]
    ctl = robot.part_controllers["right"]
    sim = env.sim
    base = dict(
        type=ctype,
        qpos_idx=[int(i) for i in ctl.qpos_index], dof_idx=[int(i) for i in ctl.qvel_index],
        act_idx=[int(i) for i in robot._ref_actuators_indexes_dict["right"]],
        eef_site=int(sim.model.site_name2id(ctl.ref_name)),
        base_site=int(sim.model.site_name2id(f"{ctl.naming_prefix}{ctl.part_name}_center")),
        input_min=[float(x) for x in ctl.input_min], input_max=[float(x) for x in ctl.input_max],
        output_min=[float(x) for x in ctl.output_min], output_max=[float(x) for x in ctl.output_max],
        grip_act=[int(i) for i in robot._ref_actuators_indexes_dict["right_gripper"]],
        grip_sign=[-1.0, 1.0], grip_speed=float(robot.gripper["right"].speed),
        grip_qpos_idx=[int(i) for i in robot._ref_gripper_joint_pos_indexes["right"]],
        grip_dof_idx=[int(i) for i in robot._ref_gripper_joint_vel_indexes["right"]],
    )
    if ctype in ("JOINT_POSITION", "OSC_POSITION", "OSC_POSE"):
        base["kp"] = [float(x) for x in np.atleast_1d(ctl.kp)]   # variable-impedance recordings overwrite this with the constructor value
        base["kd"] = [float(x) for x in np.atleast_1d(ctl.kd)]
        base["damping_ratio"] = 1.0
    if ctype in ("OSC_POSITION", "OSC_POSE"):
        base["uncouple"] = int(ctl.uncoupling)
    if ctype == "JOINT_TORQUE":
        base["torque_limits"] = [[float(x) for x in ctl.torque_limits[0]], [float(x) for x in ctl.torque_limits[1]]]
    if ctype in ("JOINT_POSITION", "JOINT_TORQUE"):
        base["use_torque_compensation"] = int(getattr(ctl, "use_torque_compensation", True))
    return base


def two_arm_cfg(env, ctype, keys, obs):
    """One part controller per arm of a bimanual robot without grippers (composite_controller.py:70-95; BASELINE configs[3])."""
    sim, robot = env.sim, env.robots[0]
    parts = []
    for arm in robot.arms:
        ctl = robot.part_controllers[arm]
        pc = dict(type=ctype, qpos_idx=[int(i) for i in ctl.qpos_index], dof_idx=[int(i) for i in ctl.qvel_index],
                  act_idx=[int(i) for i in robot._ref_actuators_indexes_dict[arm]], eef_site=0, base_site=0,
                  input_min=[float(x) for x in ctl.input_min], input_max=[float(x) for x in ctl.input_max],
                  output_min=[float(x) for x in ctl.output_min], output_max=[float(x) for x in ctl.output_max],
                  grip_act=[], grip_sign=[], grip_speed=0.0, damping_ratio=1.0)
        if ctype in ("JOINT_POSITION", "JOINT_VELOCITY"):
            pc["kp"] = [float(x) for x in np.atleast_1d(ctl.kp)]
        if ctype in ("OSC_POSE", "OSC_POSITION"):   # one OSC object per arm (Baxter's default), each with its own eef / base ("<arm>_center") sites
            pc.update(kp=[float(x) for x in np.atleast_1d(ctl.kp)], kd=[float(x) for x in np.atleast_1d(ctl.kd)], uncouple=int(ctl.uncoupling),
                      eef_site=int(sim.model.site_name2id(ctl.ref_name)),
                      base_site=int(sim.model.site_name2id(f"{ctl.naming_prefix}{ctl.part_name}_center")))
        if ctype == "JOINT_VELOCITY" and ctl.velocity_limits is not None:
            lo, hi = np.broadcast_to(ctl.velocity_limits[0], (len(pc["qpos_idx"]),)), np.broadcast_to(ctl.velocity_limits[1], (len(pc["qpos_idx"]),))
            pc["velocity_limits"] = [[float(x) for x in lo], [float(x) for x in hi]]
        if ctype == "JOINT_TORQUE":
            pc["torque_limits"] = [[float(x) for x in ctl.torque_limits[0]], [float(x) for x in ctl.torque_limits[1]]]
        parts.append(pc)
    cat = lambda k: sum((p[k] for p in parts), [])   # noqa: E731
    cfg = dict(type=ctype, parts=parts, qpos_idx=cat("qpos_idx"), dof_idx=cat("dof_idx"), act_idx=cat("act_idx"), input_min=cat("input_min"),
               input_max=cat("input_max"), output_min=cat("output_min"), output_max=cat("output_max"), grip_act=[], grip_sign=[], grip_speed=0.0,
               damping_ratio=1.0, part_of=sum(([k] * len(p["qpos_idx"]) for k, p in enumerate(parts)), []), obs_keys=keys,
               obs_dims=[int(np.atleast_1d(obs[k]).size) for k in keys])
    if ctype in ("JOINT_POSITION", "JOINT_VELOCITY"):
        cfg["kp"] = cat("kp")
    if ctype == "JOINT_VELOCITY" and "velocity_limits" in parts[0]:
        cfg["velocity_limits"] = [sum((p["velocity_limits"][0] for p in parts), []), sum((p["velocity_limits"][1] for p in parts), [])]
    if ctype == "JOINT_TORQUE":
        cfg["torque_limits"] = [sum((p["torque_limits"][0] for p in parts), []), sum((p["torque_limits"][1] for p in parts), [])]
    return cfg


def pickplace_task_cfg(env):
    """Task constants of PickPlace (pick_place.py:188-199, 560-583) and of its reset path (:431-483, placement_samplers.py:221-309)."""
    sim = env.sim
    g = env.robots[0].gripper["right"]
    task = dict(objects=[o.name for o in env.objects], object_bodies=[o.root_body for o in env.objects],
                object_geoms=[list(o.contact_geoms) for o in env.objects], bin2_pos=[float(x) for x in env.bin2_pos],
                bin_size=[float(x) for x in env.bin_size], target_bin_placements=[[float(x) for x in r] for r in env.target_bin_placements],
                left_pad=list(g.important_geoms["left_fingerpad"]), right_pad=list(g.important_geoms["right_fingerpad"]),
                eef_body=env.robots[0].robot_model.eef_name["right"], grip_site=g.important_sites["grip_site"])
    if env.single_object_mode:
        task.update(single_object_mode=int(env.single_object_mode), object_id=int(env.object_id))
    if env.single_object_mode == 1:
        # _reset_internal draws with rng.choice(list(obj_names)), obj_names a SET of the object names (pick_place.py:716-722): what draw k means is
        # this process's set order (string hashing), so it is recorded with the configuration
        ids = []
        for name in list({obj.name for obj in env.objects}):
            ids.append(next(i for typ, i in env.object_to_id.items() if typ.lower() in name.lower()))
        task["mode1_order"] = ids
    task["placement"] = dict(
        bin1_pos=[float(x) for x in env.bin1_pos], z_offset=float(env.z_offset), z_rotation=env.z_rotation,
        x_half=float(env.model.mujoco_arena.table_full_size[0] / 2 - 0.05), y_half=float(env.model.mujoco_arena.table_full_size[1] / 2 - 0.05),
        objects=[dict(name=o.name, horizontal_radius=float(o.horizontal_radius), bottom_z=float(o.bottom_offset[-1]), top_z=float(o.top_offset[-1]),
                      qposadr=int(sim.model.get_joint_qpos_addr(o.joints[0])[0])) for o in env.objects],
        noise=dict(type=str(env.robots[0].initialization_noise["type"]), magnitude=float(env.robots[0].initialization_noise["magnitude"])),
        arm_init_qpos=[float(x) for x in env.robots[0].init_qpos], gripper_init_qpos=[float(x) for x in g.init_qpos],
        arm_qpos_idx=[int(i) for i in env.robots[0]._ref_joint_pos_indexes], gripper_qpos_idx=[int(i) for i in env.robots[0]._ref_gripper_joint_pos_indexes["right"]])
    return task


def env_cfg(env):
    """The constructor arguments that change what a control step MEANS and that live on the env object, not in the model: reward flavour
    (lift.py:158-159, 256-271; reward_scale = None: no normalisation), the substeps of a control step (base.py:212-218: control_timestep /
    model_timestep) and the horizon."""
    n_sub = env.control_timestep / env.model_timestep
    if abs(n_sub - round(n_sub)) > 1e-9 or round(n_sub) < 1:
        raise NotImplementedError(f"control_freq {env.control_freq}: control_timestep / model_timestep = {n_sub} is not a whole number of substeps")
    return dict(reward_shaping=bool(getattr(env, "reward_shaping", False)), reward_scale=(None if getattr(env, "reward_scale", 1.0) is None else float(env.reward_scale)),
                control_freq=float(env.control_freq), n_sub=int(round(n_sub)), horizon=int(env.horizon), ignore_done=bool(env.ignore_done), hard_reset=bool(env.hard_reset))


def _sampler_cfg(sampler, sim, env_rng=None):
    """A UniformRandomSampler as data (utils/placement_samplers.py:87-309).  Anything else (a SequentialCompositeSampler the user passed, a custom
    class) has its own draw order and is refused -- silently replacing it with the default would be the bug this function exists to prevent."""
    from robosuite.utils.placement_samplers import UniformRandomSampler

    if type(sampler) is not UniformRandomSampler:
        raise NotImplementedError(f"placement_initializer of type {type(sampler).__name__}: only UniformRandomSampler is restated on the host side (lift.py / stack.py)")
    rot = sampler.rotation
    objs = []
    for o in sampler.mujoco_objects:
        objs.append(dict(name=o.name, horizontal_radius=float(o.horizontal_radius), bottom_z=float(o.bottom_offset[-1]), top_z=float(o.top_offset[-1]),
                         qposadr=int(sim.model.get_joint_qpos_addr(o.joints[0])[0]), init_quat=([float(x) for x in o.init_quat] if hasattr(o, "init_quat") else None)))
    return dict(x_range=[float(x) for x in sampler.x_range], y_range=[float(x) for x in sampler.y_range],
                rotation=(None if rot is None else ([float(x) for x in rot] if hasattr(rot, "__iter__") else float(rot))), rotation_axis=str(sampler.rotation_axis),
                z_offset=float(sampler.z_offset), reference_pos=[float(x) for x in sampler.reference_pos],
                ensure_object_boundary_in_range=bool(sampler.ensure_object_boundary_in_range), ensure_valid_placement=bool(sampler.ensure_valid_placement), objects=objs,
                # a sampler the user built without rng= draws from an unseeded generator of its own (placement_samplers.py:44-47), not from the env's stream
                own_rng=bool(env_rng is not None and sampler.rng is not env_rng))


def reset_cfg(env):
    """What the env's hard reset draws and where it writes it (robots/robot.py:107-113, 247-259: init_qpos + noise; lift.py:311-333, stack.py:324-357:
    object sizes and the placement sampler; two_arm_peg_in_hole.py:343-351: peg radius), read off the live objects so that constructor kwargs
    (`initialization_noise`, `placement_initializer`, another robot's init_qpos) reach the host-side restatement."""
    robot, sim = env.robots[0], env.sim
    noise = robot.initialization_noise
    spec = dict(nq=int(sim.model.nq), arm_init_qpos=[float(x) for x in robot.init_qpos], arm_qpos_idx=[int(i) for i in robot._ref_joint_pos_indexes],
                noise=dict(type=str(noise["type"]), magnitude=float(noise["magnitude"])), grippers=[])
    if noise["type"] not in ("gaussian", "uniform"):
        raise ValueError("Error: Invalid noise type specified. Options are 'gaussian' or 'uniform'.")   # robots/robot.py:257
    for arm in robot.arms:
        if robot.has_gripper[arm]:
            spec["grippers"].append(dict(init_qpos=[float(x) for x in robot.gripper[arm].init_qpos], qpos_idx=[int(i) for i in robot._ref_gripper_joint_pos_indexes[arm]]))
    name = type(env).__name__
    if name == "Lift":
        c = env.cube
        # the size range is a literal of Lift._load_model (lift.py:311-318: BoxObject(size_min=[0.020] * 3, size_max=[0.022] * 3)), not a constructor
        # argument, and BoxObject keeps only the drawn size; the density is the object's
        spec["cube"] = dict(size_min=[0.020] * 3, size_max=[0.022] * 3, density=float(getattr(c, "density", 1000.0)))
        if not np.all((np.array(c.size) >= 0.020 - 1e-12) & (np.array(c.size) <= 0.022 + 1e-12)):
            raise NotImplementedError(f"Lift cube of size {c.size}: outside the size range this restatement of lift.py:311-318 draws from")
        spec["sampler"] = _sampler_cfg(env.placement_initializer, sim, env.rng)
    elif name == "Stack":
        spec["sampler"] = _sampler_cfg(env.placement_initializer, sim, env.rng)
    elif name == "TwoArmPegInHole":
        spec["peg"] = dict(radius=[float(x) for x in env.peg_radius], length=float(env.peg_length))
    return spec


def grasp_cfg(env):
    """Geom groups of ManipulationEnv._check_grasp (manipulation_env.py:361-362) and the eef names of the single-arm observation record."""
    robot = env.robots[0]
    if len(robot.arms) != 1 or not robot.has_gripper[robot.arms[0]]:
        return None
    g = robot.gripper[robot.arms[0]]
    return dict(left_pad=list(g.important_geoms["left_fingerpad"]), right_pad=list(g.important_geoms["right_fingerpad"]),
                grip_site=g.important_sites["grip_site"], eef_body=robot.robot_model.eef_name[robot.arms[0]])


def patch_joint_velocity_defect():
    """JointVelocityController cannot be constructed in the surveyed snapshot: joint_vel.py:118 assigns `self.torque_compensation = ...`
    although `torque_compensation` is a read-only property of Controller (controller.py:303-311), and run_controller then tests the truth
    value of that 7-vector (joint_vel.py:186).  SURVEY.md section 8 (config 4) resolves the defect as use_torque_compensation = True.
    This patch does exactly that and nothing else: the property accepts (and ignores) the assignment and returns qfrc_bias[qvel_index] wrapped
    in an object whose truth value is True and which adds to an ndarray as a plain ndarray; set_goal / run_controller / RingBuffer / the
    saturation logic stay the reference's own code."""
    from robosuite.controllers.parts.generic.joint_vel import JointVelocityController

    class _Compensation:
        """qfrc_bias[qvel_index] with truth value True; adds to an ndarray as a plain ndarray (so later comparisons keep numpy semantics)."""

        def __init__(self, v):
            self.v = np.array(v, dtype=np.float64)

        def __bool__(self):
            return True

        def __array__(self, dtype=None, copy=None):
            return self.v if dtype is None else self.v.astype(dtype)

    def getter(self):
        return _Compensation(self.sim.data.qfrc_bias[self.qvel_index])

    JointVelocityController.torque_compensation = property(getter, lambda self, value: None)


def arm_controller_type(env) -> str:
    """`name` of the arm part controller(s) the reference built (controller.py `name` property: "OSC_POSE", "JOINT_VELOCITY", ...)."""
    robot = env.robots[0]
    names = {robot.part_controllers[a].name for a in robot.arms}
    if len(names) != 1:
        raise NotImplementedError(f"arms under different part-controller types ({sorted(names)}) have no in-kernel implementation")
    return names.pop()


def extract(env, obs=None):
    """(compiled model, cfg dict) of a constructed reference env: what `VecEnv(name, n, flat, cfg)` takes.  `obs`: an observation dict of this env
    (default: env.reset()) -- the key order and sizes of its per-key record become cfg["obs_keys"] / ["obs_dims"]."""
    from .backend import CTRL_TYPES, IMPEDANCE_MODES

    obs = env.reset() if obs is None else obs
    sim, robot = env.sim, env.robots[0]
    flat = sim.model._model._flat
    keys = [k for k in obs.keys() if not k.endswith("-state")]
    ctype = arm_controller_type(env)
    if ctype not in CTRL_TYPES:
        raise NotImplementedError(f"arm part controller {ctype!r} has no in-kernel implementation (have {sorted(CTRL_TYPES)}); it still runs through the B = 1 shim")
    if len(robot.arms) == 2:
        if any(robot.has_gripper[a] for a in robot.arms) if isinstance(robot.has_gripper, dict) else robot.has_gripper:
            raise NotImplementedError("bimanual robots are carried without grippers (gripper_types=None)")
        cfg = two_arm_cfg(env, ctype, keys, obs)
        cfg["env"], cfg["reset"] = env_cfg(env), reset_cfg(env)
        return flat, cfg
    ctl = robot.part_controllers["right"]
    mode = getattr(ctl, "impedance_mode", "fixed")
    interp = getattr(ctl, "interpolator", None) or getattr(ctl, "interpolator_pos", None)
    plain = ctype == "OSC_POSE" and mode == "fixed" and interp is None
    cfg = controller_cfg(env) if plain else controller_cfg_generic(env, ctype)
    if mode != "fixed":
        if mode not in IMPEDANCE_MODES:
            raise NotImplementedError(f"impedance mode {mode!r}")
        cfg["impedance_mode"] = mode
        cfg["kp_limits"] = [[float(x) for x in ctl.kp_min], [float(x) for x in ctl.kp_max]]
        cfg["damping_ratio_limits"] = [[float(x) for x in ctl.damping_ratio_min], [float(x) for x in ctl.damping_ratio_max]]
        cfg["input_min"], cfg["input_max"] = [float(x) for x in ctl.input_min], [float(x) for x in ctl.input_max]
    if interp is not None:
        cfg["interp_steps"] = int(interp.total_steps)      # ceil(ramp_ratio * controller_freq / policy_freq), traj_utils.py:55-57
    cfg["grip_sign"] = gripper_signs(robot.gripper["right"])
    name = type(env).__name__
    cfg["env"], cfg["reset"], cfg["grasp"] = env_cfg(env), reset_cfg(env), grasp_cfg(env)
    if name.startswith("PickPlace"):
        cfg["task"] = pickplace_task_cfg(env)
    cfg["obs_keys"] = keys
    cfg["obs_dims"] = [int(np.atleast_1d(obs[k]).size) for k in keys]
    if name.startswith("PickPlace") and env.single_object_mode == 1:
        # the object keys of the record are those of whichever object this episode drew (`Can_pos` ...): same slots every episode, named neutrally
        used = env.obj_to_use
        cfg["obs_keys"] = [k.replace(used + "_", "obj_", 1) if k.startswith(used + "_") else k for k in keys]
    if name in ("Stack", "Lift"):
        cfg["table_height"] = float(env.model.mujoco_arena.table_offset[2])     # lift.py:441, stack.py:276: the height _check_success / staged_rewards measure from
    return flat, cfg


# ------------------------------------------------------------------------------------------------------------------------------------------
# the entry point
# ------------------------------------------------------------------------------------------------------------------------------------------
def _import_reference(reference_path=None):
    """The reference package over this project's `mujoco`-shaped shim (KinematicsBackend), or None when no checkout is reachable."""
    from . import shim

    if "robosuite" in sys.modules and getattr(shim, "_BACKEND_FACTORY", None) is not None:
        return sys.modules["robosuite"]      # already imported over some shim backend (tests, tools/gen_golden.py): use it as it is
    for p in (reference_path, os.environ.get("ROBOSUITE_PATH"), "/root/reference"):
        if p and os.path.isdir(os.path.join(p, "robosuite")) and p not in sys.path:
            sys.path.insert(0, p)
            break
    try:
        import mujoco  # noqa: F401  (a real MuJoCo: the reference runs natively and needs no shim to be read)
    except ImportError:
        shim.install(KinematicsBackend)
    try:
        import robosuite
    except ImportError:
        return None
    return robosuite


def controller_type_of(controller_configs, robot: str) -> str:
    """Arm part-controller type a `controller_configs` dict asks for (composite config: body_parts -> right -> type), the robot default otherwise."""
    if controller_configs is None:
        return "OSC_POSE"
    parts = controller_configs.get("body_parts", controller_configs.get("body_parts_controller_configs", {}))
    for arm in ("right", "left"):
        if arm in parts and "type" in parts[arm]:
            return str(parts[arm]["type"])
    return str(controller_configs.get("type", "OSC_POSE"))


def from_reference(env_name, robots="Panda", controller_configs=None, seed=0, reference_path=None, defaults=None, **kwargs):
    """(flat, cfg) by constructing the reference's own env class (needs a reachable robosuite checkout).  `defaults`: constructor kwargs under the
    caller's (None = BENCH_KWARGS, what the shipped assets and the fixtures were recorded with)."""
    suite = _import_reference(reference_path)
    if suite is None:
        raise ImportError("robosuite is not importable and no checkout was found (reference_path=, $ROBOSUITE_PATH)")
    if controller_type_of(controller_configs, str(robots)) == "JOINT_VELOCITY":
        patch_joint_velocity_defect()
    kw = {**(BENCH_KWARGS if defaults is None else defaults), **kwargs}
    env = suite.make(env_name, robots=robots, controller_configs=controller_configs, seed=seed, **kw)
    return extract(env)


def load_shipped(stem):
    return mjcf.load_model(os.path.join(ASSETS, stem + ".rsim")), json.load(open(os.path.join(ASSETS, stem + ".cfg.json")))


# ---- what make() does with the reference constructor's kwargs ---------------------------------------------------------------------------------
# Rendering: the batched path has none.  These may be passed with their "off" values (or any value where nothing is rendered anyway); asking
# for a renderer or camera observations raises.
RENDER_OFF = dict(has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False)
RENDER_IGNORED = {"render_camera", "render_collision_mesh", "render_visual_mesh", "render_gpu_device_id", "camera_names", "camera_heights", "camera_widths",
                  "camera_depths", "camera_segmentations", "renderer", "renderer_config"}
# Consumed on the host side of the boundary: extract() reads them off the live env into cfg["env"] / cfg["reset"] (or, for a shipped
# configuration, make() patches the shipped cfg): reward flavour, robot joint noise, substeps per control step.
HOST_KWARGS = ("reward_shaping", "reward_scale", "initialization_noise", "control_freq")
# The reference's own defaults for those (environments/manipulation/lift.py:150-166 and the other task constructors; robots/robot.py:107-111)
REFERENCE_DEFAULTS = dict(reward_shaping=False, reward_scale=1.0, initialization_noise="default", control_freq=20, use_object_obs=True, hard_reset=True,
                          lite_physics=True, ignore_done=False)


def _noise_cfg(noise):
    """robots/robot.py:107-113"""
    if noise is None:
        return dict(type="gaussian", magnitude=0.0)
    if noise == "default":
        return dict(type="gaussian", magnitude=0.02)
    if noise["type"] not in ("gaussian", "uniform"):
        raise ValueError("Error: Invalid noise type specified. Options are 'gaussian' or 'uniform'.")
    return dict(type=str(noise["type"]), magnitude=float(noise["magnitude"]) if noise["magnitude"] else 0.0)


def apply_host_kwargs(flat, cfg, host):
    """A shipped configuration under other host-side constructor arguments: the same edits extract() would have read off a live env."""
    cfg = json.loads(json.dumps(cfg))
    env = cfg.setdefault("env", {})
    if "reward_shaping" in host:
        env["reward_shaping"] = bool(host["reward_shaping"])
    if "reward_scale" in host:
        env["reward_scale"] = None if host["reward_scale"] is None else float(host["reward_scale"])
    if "control_freq" in host:
        h = float(np.asarray(flat.arrays["timestep"]).ravel()[0])
        n_sub = (1.0 / float(host["control_freq"])) / h          # base.py:212-218: control_timestep = 1 / control_freq
        if abs(n_sub - round(n_sub)) > 1e-9 or round(n_sub) < 1:
            raise ValueError(f"control_freq {host['control_freq']}: {n_sub} substeps of {h} s per control step is not a whole number")
        env["control_freq"], env["n_sub"] = float(host["control_freq"]), int(round(n_sub))
    if "initialization_noise" in host:
        n = _noise_cfg(host["initialization_noise"])
        if "reset" in cfg:
            cfg["reset"]["noise"] = n
        if "placement" in cfg.get("task", {}):
            cfg["task"]["placement"]["noise"] = n
    return cfg


def make(env_name, robots="Panda", n_envs=1, controller_configs=None, seed=0, horizon=1000, device=0, bank_episodes=4, stream_groups=1, env_ids=None,
         source="auto", reference_path=None, alternating=False, **kwargs):
    """Batched counterpart of `robosuite.make(env_name, robots=..., controller_configs=..., **kwargs)` (environments/base.py:23-42): a
    `vec_env.VecEnv` of `n_envs` environments on `cuda:device`, env i seeded by `seed + i`.

    kwargs are the reference constructor's, with the reference's defaults (sparse reward: reward_shaping=False, reward_scale=1.0, control_freq=20,
    horizon=1000, initialization_noise="default", ...).  What happens to each:
      * model kwargs (robots, gripper_types, env_configuration, table_full_size, table_friction, single_object_mode, ...) reach the reference's
        own model assembly; the compiled model carries them;
      * reward_shaping / reward_scale / initialization_noise / placement_initializer / control_freq / use_object_obs are read off the constructed env
        into cfg["env"] / cfg["reset"] / the observation program and honoured by the host-side reset and the on-device epilogue;
      * horizon: the vectorised env reports `done` there and restarts the env on the device (gym auto-reset), whatever `ignore_done` says;
      * rendering kwargs: there is no renderer; has_renderer / has_offscreen_renderer / use_camera_obs = True raise, the rest is ignored;
      * hard_reset=False raises (every restart is a hard reset: sizes and placements are redrawn, base.py:277-347);
      * anything else raises TypeError -- no kwarg is dropped silently.

    source: "assets" = only the shipped BASELINE configurations (with host-side kwargs patched into their cfg); "reference" = always construct the
    reference's env class and read the model and the configuration off it; "auto" (default) = shipped assets when (env_name, robots, controller
    type, model kwargs) name one of them, the reference otherwise.
    alternating=True: a `vec_env.AlternatingVecEnv` (two half-batches stepped alternately: closed-loop compatible, fills the drain of a lockstep launch)."""
    from .vec_env import AlternatingVecEnv, VecEnv

    if not isinstance(robots, str):
        if len(robots) != 1:
            raise NotImplementedError("one robot per env (the two-arm tasks are carried as single-robot bimanual configurations)")
        robots = robots[0]
    for k, off in RENDER_OFF.items():
        if kwargs.get(k, off) != off:
            raise NotImplementedError(f"{k}={kwargs[k]!r}: the batched path has no renderer and no camera observations (render from a reference env over robosuite_amd.shim)")
    if not kwargs.get("hard_reset", True):
        raise NotImplementedError("hard_reset=False: every restart of a batched env is a hard reset (object sizes and placements are redrawn)")
    model_kw = {k: v for k, v in kwargs.items() if k not in RENDER_OFF and k not in RENDER_IGNORED and k not in HOST_KWARGS and k not in ("hard_reset", "ignore_done", "lite_physics")}
    host_kw = {k: kwargs.get(k, REFERENCE_DEFAULTS[k]) for k in HOST_KWARGS}
    key = (env_name, robots, controller_type_of(controller_configs, robots))
    stem = SHIPPED.get(key)
    extra = {k: v for k, v in model_kw.items() if REFERENCE_DEFAULTS.get(k, object()) != v}
    shipped_ok = stem is not None and extra == SHIPPED_KWARGS.get(stem, {}) and os.path.exists(os.path.join(ASSETS, stem + ".rsim"))
    if source == "assets" or (source == "auto" and shipped_ok):
        if not shipped_ok:
            raise ValueError(f"no shipped assets for {key} with kwargs {extra}; shipped: {sorted(SHIPPED)} (use source='reference' with a robosuite checkout)")
        flat, cfg = load_shipped(stem)
        cfg = apply_host_kwargs(flat, cfg, host_kw)
    else:
        # the reference constructor sees every kwarg (an unknown one raises TypeError there, as in suite.make); extract() reads the result
        ref_kw = {**RENDER_OFF, **{k: v for k, v in kwargs.items() if k not in RENDER_OFF}, "horizon": horizon}
        flat, cfg = from_reference(env_name, robots, controller_configs, seed=seed, reference_path=reference_path, defaults={}, **ref_kw)
    if alternating:
        return AlternatingVecEnv(env_name, n_envs, flat, cfg, device=device, seed=seed, horizon=horizon, env_ids=env_ids, bank_episodes=bank_episodes)
    return VecEnv(env_name, n_envs, flat, cfg, device=device, seed=seed, horizon=horizon, env_ids=env_ids, bank_episodes=bank_episodes, stream_groups=stream_groups)


def ship_assets(reference_path=None):
    """(Re)generate robosuite_amd/assets/ for the BASELINE configurations from the reference (build container only)."""
    suite = _import_reference(reference_path)
    from robosuite.controllers import load_part_controller_config
    from robosuite.controllers.composite.composite_controller_factory import refactor_composite_controller_config

    for (env_name, robot, ctype), stem in SHIPPED.items():
        cc = None
        if ctype != "OSC_POSE":
            cc = refactor_composite_controller_config(load_part_controller_config(default_controller=ctype), robot, ["right", "left"])
        flat, cfg = from_reference(env_name, robot, cc, seed=1 if stem == "lift_panda" else 0, **SHIPPED_KWARGS.get(stem, {}))
        mjcf.save_model(flat, os.path.join(ASSETS, stem + ".rsim"))
        with open(os.path.join(ASSETS, stem + ".cfg.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        print(stem, "nq", flat.nq, "nv", flat.nv, "nbody", flat.nbody, "obs", sum(cfg["obs_dims"]))
    return suite


if __name__ == "__main__":
    if "--ship" in sys.argv:
        ship_assets()
