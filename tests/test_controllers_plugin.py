"""Batched Controller plugin protocol (robosuite_amd/controllers.py): user part controllers on [B, ...] device tensors, evaluated once per substep
between rsim_step1 and rsim_step2 of the whole batch -- the batched form of controllers/parts/controller.py:35-44, 140-147."""
import numpy as np
import pytest

from tests.util import load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _lift_joint_torque(B, horizon=0, bank=0):
    from robosuite_amd import lift
    g, cfg, flat = load_golden("ctl_joint_torque")
    env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=4, horizon=horizon, bank_episodes=bank)
    return g, cfg, flat, env


def _parts(env, cfg, flat):
    from robosuite_amd.controllers import BatchState, Part, TorchGripController, TorchJointTorqueController
    st = BatchState(env.batch)
    cr = np.asarray(flat.actuator_ctrlrange)
    arm = TorchJointTorqueController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]),
                                     input_max=cfg["input_max"], input_min=cfg["input_min"], output_max=cfg["output_max"], output_min=cfg["output_min"],
                                     torque_limits=cfg["torque_limits"])
    grip = TorchGripController(st, dict(joints=cfg["grip_qpos_idx"], qpos=cfg["grip_qpos_idx"], qvel=cfg["grip_dof_idx"]),
                               (cr[cfg["grip_act"], 0], cr[cfg["grip_act"], 1]), signs=cfg["grip_sign"], speed=cfg["grip_speed"])
    n = len(cfg["qpos_idx"])
    return st, [Part(arm, slice(0, n), cfg["act_idx"]), Part(grip, slice(n, n + 1), cfg["grip_act"])]


def test_torch_joint_torque_and_grip_plugins_equal_the_in_kernel_controllers():
    """The reference's JointTorqueController + SimpleGripController re-expressed as BatchedController plugins and driven substep by substep through
    rsim_step1 / rsim_step2 / rsim_step2_last, against the same laws inside the fused kernel (one launch per control step): same states, observation
    records, rewards and horizon flags -- through an on-device episode restart."""
    from robosuite_amd import lift
    from robosuite_amd.controllers import HostControlledEnv
    B, T, H = 6, 9, 5
    g, cfg, flat, fused = _lift_joint_torque(B, horizon=H, bank=3)
    _, _, _, hosted = _lift_joint_torque(B, horizon=H, bank=3)
    st, parts = _parts(hosted, cfg, flat)
    env = HostControlledEnv(hosted, parts)
    assert env.action_dim == fused.model.action_dim == 8
    acts = torch.tensor(lift.env_actions(np.arange(B), T, action_dim=8), device="cuda")
    for t in range(T):
        fused.step(acts[t])
        obs, rew, done, info = env.step(acts[t])
        for k in ("done", "ep_step", "ep_index"):
            assert np.array_equal(fused.batch.get(k), hosted.batch.get(k)), (t, k)
        # same fp32 laws; the kernel's compiler fuses multiply-adds that torch issues as two operations, so the torques agree to the last bits
        # rather than exactly: states to 2e-6 / 2e-5, records and rewards to 1e-5 over the nine control steps
        for k, tol in (("qpos", 2e-6), ("qvel", 2e-5), ("ctrl", 1e-5), ("obs", 2e-5), ("reward", 1e-6)):
            a, b = fused.batch.get(k), hosted.batch.get(k)
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), (t, k, np.abs(a - b).max())
    assert fused.batch.get("ep_index").tolist() == [1] * B and int(hosted.batch.get("bank_stale").sum()) == 0
    assert torch.allclose(obs, fused.obs(), atol=1e-4) and torch.allclose(rew, fused.reward(), atol=1e-6)


def test_batch_state_jacobians_and_site_frames_match_the_c_abi():
    """BatchState.site_jacobian / site_pose (what a user OSC-type plugin builds on) against rsim_jac_site and the oracle's site frames."""
    from tests.util import make_oracle
    B = 5
    g, cfg, flat, env = _lift_joint_torque(B)
    st, _ = _parts(env, cfg, flat)
    acts = torch.zeros(B, 8, device="cuda").uniform_(-1, 1)
    env.step(acts)
    env.batch.step1()
    site = flat.names["site"].index("gripper0_right_grip_site")
    jp, jr = st.site_jacobian(site)
    pos, mat = st.site_pose(site)
    om, od, _ = make_oracle(flat)
    for e in range(B):
        hp, hr = env.batch.jac_site(e, site)
        assert np.abs(jp[e].cpu().numpy() - hp).max() < 2e-6 and np.abs(jr[e].cpu().numpy() - hr).max() < 2e-6
    od.qpos[:] = env.batch.get("qpos")[0]; od.qvel[:] = env.batch.get("qvel")[0]; od.forward()
    assert np.abs(pos[0].cpu().numpy() - od.site_xpos[3 * site:3 * site + 3]).max() < 5e-6
    assert np.abs(mat[0].cpu().numpy().ravel() - od.site_xmat[9 * site:9 * site + 9]).max() < 5e-6


def test_a_user_defined_controller_runs_on_the_batch():
    """A law the kernel does not have -- joint-space PD towards a posture with the mass matrix and gravity compensation, plus a task-space damping
    term through the site Jacobian -- written as a BatchedController: it holds 64 arms at the posture against gravity."""
    from robosuite_amd.controllers import BatchedController, HostControlledEnv, Part
    B = 64
    g, cfg, flat, task = _lift_joint_torque(B)
    st, parts = _parts(task, cfg, flat)
    site = flat.names["site"].index("gripper0_right_grip_site")

    class PostureController(BatchedController):
        name = "POSTURE_PD"

        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.target = None

        def set_goal(self, action):
            if self.target is None:
                self.target = self.joint_pos.clone()
            self.target = self.target + 0.02 * action

        def run_controller(self):
            des = 100.0 * (self.target - self.joint_pos) - 20.0 * self.joint_vel
            jp, _ = self.state.site_jacobian(site)
            J = jp[:, :, self.qvel_index]
            damp = -torch.einsum("bki,bk->bi", J, 5.0 * torch.einsum("bki,bi->bk", J, self.joint_vel))
            return torch.einsum("bij,bj->bi", self.mass_matrix, des) + damp + self.torque_compensation

    arm = PostureController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (parts[0].controller.actuator_min.cpu().numpy(), parts[0].controller.actuator_max.cpu().numpy()))
    env = HostControlledEnv(task, [Part(arm, slice(0, 7), cfg["act_idx"]), parts[1]])
    q0 = task.batch.get("qpos")[:, :7].copy()
    a = torch.zeros(B, 8, device="cuda")
    for t in range(10):
        obs, rew, done, info = env.step(a)
    q = task.batch.get("qpos")[:, :7]
    assert np.isfinite(q).all() and np.abs(q - q0).max() < 5e-3 and np.abs(task.batch.get("qvel")[:, :7]).max() < 5e-2


def test_torch_osc_plugin_tracks_the_in_kernel_osc_controller():
    """TorchOSCController (the reference's OperationalSpaceController re-expressed for the batched protocol, pinned on CPU against the reference's
    recorded calls: tests/test_controllers_host.py) driven through rsim_step1 / rsim_step2 on device tensors -- frames, site Jacobians, mass matrix
    and bias from BatchState -- against the OSC law inside the fused kernel, through an on-device episode restart (goals and initial_joint of a
    restarted env are re-captured from its new state)."""
    from robosuite_amd import lift
    from robosuite_amd.controllers import BatchState, HostControlledEnv, Part, TorchGripController, TorchOSCController
    from tests.util import load_golden
    g, cfg, flat = load_golden("seed1_full")
    B, T, H = 6, 8, 5
    ids = np.arange(B)
    fused = lift.LiftBatch(flat, cfg, ids, seed0=4, horizon=H, bank_episodes=3)
    hosted = lift.LiftBatch(flat, cfg, ids, seed0=4, horizon=H, bank_episodes=3)
    st = BatchState(hosted.batch)
    cr = np.asarray(flat.actuator_ctrlrange)
    arm = TorchOSCController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]),
                             cfg["eef_site"], cfg["base_site"], kp=cfg["kp"], damping_ratio=cfg["damping_ratio"], input_max=cfg["input_max"], input_min=cfg["input_min"],
                             output_max=cfg["output_max"], output_min=cfg["output_min"], uncouple_pos_ori=bool(cfg["uncouple"]))
    grip = TorchGripController(st, dict(joints=cfg["grip_qpos_idx"], qpos=cfg["grip_qpos_idx"], qvel=cfg["grip_dof_idx"]),
                               (cr[cfg["grip_act"], 0], cr[cfg["grip_act"], 1]), signs=cfg["grip_sign"], speed=cfg["grip_speed"])
    env = HostControlledEnv(hosted, [Part(arm, slice(0, 6), cfg["act_idx"]), Part(grip, slice(6, 7), cfg["grip_act"])])
    env.reset()
    fused.reset()
    acts = torch.tensor(0.5 * lift.env_actions(ids, T), device="cuda")
    worst = {}
    for t in range(T):
        fused.step(acts[t])
        env.step(acts[t])
        for k in ("done", "ep_step", "ep_index"):
            assert np.array_equal(fused.batch.get(k), hosted.batch.get(k)), (t, k)
        for k, tol in (("qpos", 2e-6), ("qvel", 2e-5), ("ctrl", 1e-4), ("reward", 1e-6)):       # measured 8e-8, 9e-7, 6e-6, 3e-8
            a, b = fused.batch.get(k), hosted.batch.get(k)
            e = np.abs(a - b).max() / max(1.0, np.abs(a).max())
            worst[k] = max(worst.get(k, 0.0), float(e))
            assert e <= tol, (t, k, e)
    print("torch OSC plugin vs in-kernel OSC, worst relative deviations over", T, "control steps:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert fused.batch.get("ep_index").tolist() == [1] * B


def test_torch_joint_position_plugin_tracks_the_in_kernel_controller():
    """TorchJointPositionController through rsim_step1 / rsim_step2 against JOINT_POSITION inside the fused kernel, through an on-device episode restart
    (the goal of a restarted env is re-captured from its new joint positions)."""
    from robosuite_amd import lift
    from robosuite_amd.controllers import BatchState, HostControlledEnv, Part, TorchGripController, TorchJointPositionController
    from tests.util import load_golden
    g, cfg, flat = load_golden("ctl_joint_position")
    B, T, H = 6, 8, 5
    ids = np.arange(B)
    fused = lift.LiftBatch(flat, cfg, ids, seed0=4, horizon=H, bank_episodes=3)
    hosted = lift.LiftBatch(flat, cfg, ids, seed0=4, horizon=H, bank_episodes=3)
    st = BatchState(hosted.batch)
    cr = np.asarray(flat.actuator_ctrlrange)
    arm = TorchJointPositionController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]),
                                       input_max=cfg["input_max"], input_min=cfg["input_min"], output_max=cfg["output_max"], output_min=cfg["output_min"],
                                       kp=cfg["kp"], damping_ratio=cfg["damping_ratio"])
    grip = TorchGripController(st, dict(joints=cfg["grip_qpos_idx"], qpos=cfg["grip_qpos_idx"], qvel=cfg["grip_dof_idx"]),
                               (cr[cfg["grip_act"], 0], cr[cfg["grip_act"], 1]), signs=cfg["grip_sign"], speed=cfg["grip_speed"])
    env = HostControlledEnv(hosted, [Part(arm, slice(0, 7), cfg["act_idx"]), Part(grip, slice(7, 8), cfg["grip_act"])])
    env.reset(); fused.reset()
    acts = torch.tensor(lift.env_actions(ids, T, action_dim=8), device="cuda")
    worst = {}
    for t in range(T):
        fused.step(acts[t]); env.step(acts[t])
        for k in ("done", "ep_step", "ep_index"):
            assert np.array_equal(fused.batch.get(k), hosted.batch.get(k)), (t, k)
        for k, tol in (("qpos", 2e-6), ("qvel", 2e-5), ("ctrl", 1e-5), ("reward", 1e-6)):          # measured 8e-10, 3e-7, 1e-7, 0
            a, b = fused.batch.get(k), hosted.batch.get(k)
            e = np.abs(a - b).max() / max(1.0, np.abs(a).max())
            worst[k] = max(worst.get(k, 0.0), float(e))
            assert e <= tol, (t, k, e)
    print("torch JOINT_POSITION plugin vs in-kernel, worst relative deviations:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert fused.batch.get("ep_index").tolist() == [1] * B


def test_pickplace_record_of_a_host_controlled_step_is_a_step_record_not_a_reset_record():
    """rsim_step2_last on PickPlace: the `{obj}_to_robot0_eef_pos / _quat` entries of the observation record (RSIM_OBS_REL_POS / REL_QUAT) are those of a
    control step -- the object pose of the previous record against the current gripper pose -- as rsim_control_step writes them; they read zero only in
    the record reset() returns.  (Round 3 derived "reset record" from "no in-kernel controller ran" and zeroed them under plugin controllers.)  The torch
    OSC + GRIP plugins against the in-kernel ones on the IIWA + Robotiq140 model, observation records compared entry by entry."""
    from robosuite_amd import pick_place
    from robosuite_amd.controllers import BatchState, HostControlledEnv, Part, TorchGripController, TorchOSCController
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    B, T = 4, 3
    ids = np.arange(B)
    fused = pick_place.PickPlaceBatch(flat, cfg, ids, seed0=2)
    hosted = pick_place.PickPlaceBatch(flat, cfg, ids, seed0=2)
    st = BatchState(hosted.batch)
    cr = np.asarray(flat.actuator_ctrlrange)
    arm = TorchOSCController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]),
                             cfg["eef_site"], cfg["base_site"], kp=cfg["kp"], damping_ratio=cfg["damping_ratio"], input_max=cfg["input_max"], input_min=cfg["input_min"],
                             output_max=cfg["output_max"], output_min=cfg["output_min"], uncouple_pos_ori=bool(cfg["uncouple"]))
    grip = TorchGripController(st, dict(joints=cfg["grip_qpos_idx"], qpos=cfg["grip_qpos_idx"], qvel=cfg["grip_dof_idx"]),
                               (cr[cfg["grip_act"], 0], cr[cfg["grip_act"], 1]), signs=cfg["grip_sign"], speed=cfg["grip_speed"])
    env = HostControlledEnv(hosted, [Part(arm, slice(0, 6), cfg["act_idx"]), Part(grip, slice(6, 7), cfg["grip_act"])])
    env.reset(); fused.reset()
    fused.batch.observe(); hosted.batch.observe()
    dims = np.cumsum([0] + cfg["obs_dims"])
    rel = [k for k, key in enumerate(cfg["obs_keys"]) if key.endswith("_to_robot0_eef_pos")]
    assert len(rel) == 4 and all(np.all(hosted.batch.get("obs")[:, dims[k]:dims[k + 1]] == 0) for k in rel)      # the reset record: empty observation cache
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    for t in range(T):
        a = 0.3 * (torch.rand(B, 7, device="cuda", generator=gen) * 2 - 1)
        fused.step(a); env.step(a)
        of, oh = fused.batch.get("obs"), hosted.batch.get("obs")
        for k in rel:
            assert np.abs(oh[:, dims[k]:dims[k + 1]]).max() > 0.05, (t, cfg["obs_keys"][k])            # objects are decimetres away from the gripper
        for k, key in enumerate(cfg["obs_keys"]):
            x, y = of[:, dims[k]:dims[k + 1]], oh[:, dims[k]:dims[k + 1]]
            if key.endswith("quat") or key.endswith("quat_site"):
                y = y * np.sign(np.sum(x * y, axis=1, keepdims=True))
            tol = 0.5 if key.endswith("joint_acc") else (2e-2 if key.endswith("vel") else 2e-3)          # the Robotiq's undamped finger links track loosely
            assert np.abs(x - y).max() < tol, (t, key, np.abs(x - y).max())
