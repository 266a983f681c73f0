"""The torch plugin controllers of robosuite_amd/controllers.py on CPU tensors against what the REFERENCE's classes did: tests/golden/lift_panda_ctl_joint_torque
holds, for every run_controller() call of the reference's JointTorqueController inside the env loop, the state it saw, its goal and the torques it
returned; `ctrl` holds what reached the actuators (arm torques and SimpleGripController's position targets).  (The GPU test
tests/test_controllers_plugin.py drives the same classes on device tensors against the in-kernel controllers.)"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from robosuite_amd.controllers import BatchedController, TorchGripController, TorchJointTorqueController  # noqa: E402
from tests.util import load_golden, make_oracle  # noqa: E402


class CpuState:
    """The attributes of BatchState the joint-space controllers read, as CPU tensors [B, ...]."""

    def __init__(self, B, nq, nv):
        self.B, self.device = B, torch.device("cpu")
        self.qpos, self.qvel, self.qfrc_bias = torch.zeros(B, nq), torch.zeros(B, nv), torch.zeros(B, nv)
        self.qM = torch.zeros(B, nv, nv)


def _controllers(cfg, flat, st):
    cr = np.asarray(flat.actuator_ctrlrange)
    arm = TorchJointTorqueController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]),
                                     input_max=cfg["input_max"], input_min=cfg["input_min"], output_max=cfg["output_max"], output_min=cfg["output_min"],
                                     torque_limits=cfg["torque_limits"])
    grip = TorchGripController(st, dict(joints=cfg["grip_qpos_idx"], qpos=cfg["grip_qpos_idx"], qvel=cfg["grip_dof_idx"]),
                               (cr[cfg["grip_act"], 0], cr[cfg["grip_act"], 1]), signs=cfg["grip_sign"], speed=cfg["grip_speed"])
    return arm, grip


def test_torch_joint_torque_and_grip_controllers_reproduce_the_reference_classes_call_by_call():
    g, cfg, flat = load_golden("ctl_joint_torque")
    om, od, _ = make_oracle(flat)
    n_sub = len(g["sub_qpos"]) // len(g["actions"])
    B = 3                                                      # three copies of the env: the batch dimension must not mix rows
    st = CpuState(B, flat.nq, flat.nv)
    arm, grip = _controllers(cfg, flat, st)
    arm.reset_goal(); grip.reset_goal()
    worst = 0.0
    for t, a in enumerate(g["actions"]):
        act = torch.tensor(np.repeat(a[None], B, 0), dtype=torch.float32)
        arm.set_goal(act[:, :7]); grip.set_goal(act[:, 7:8])
        for s in range(n_sub):
            k = t * n_sub + s
            od.qpos[:] = g["sub_qpos"][k]; od.qvel[:] = g["sub_qvel"][k]; od.forward()
            st.qpos[:] = torch.tensor(g["sub_qpos"][k], dtype=torch.float32); st.qvel[:] = torch.tensor(g["sub_qvel"][k], dtype=torch.float32)
            st.qfrc_bias[:] = torch.tensor(np.array(od.qfrc_bias), dtype=torch.float32)
            assert np.abs(arm.goal_torque[0].numpy() - g["sub_goal_right"][k]).max() < 1e-6
            tau = arm.run_controller()
            assert torch.equal(tau[0], tau[B - 1])
            e = np.abs(tau[0].numpy() - g["sub_tau_right"][k]).max() / max(1.0, np.abs(g["sub_tau_right"][k]).max())
            worst = max(worst, float(e))
        # what reached the actuators at the end of the control step (fixed_base_robot.py:143-153: clip to the ctrl range)
        ctrl = np.concatenate([arm.clip_torques(tau)[0].numpy(), grip.clip_torques(grip.run_controller())[0].numpy()])
        ref = g["ctrl"][t][list(cfg["act_idx"]) + list(cfg["grip_act"])]
        assert np.abs(ctrl - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (t, ctrl, ref)
    assert worst < 2e-6, worst                                  # fp32 against the reference's float64


def test_scale_action_and_reset_masks_follow_the_reference_contract():
    """Controller.scale_action (controller.py:149-168): clip to the input range, affine map onto the output range; reset_goal(mask) touches only the
    masked envs (a finished env gets fresh controller objects, robots/robot.py:271)."""
    g, cfg, flat = load_golden("ctl_joint_torque")
    st = CpuState(4, flat.nq, flat.nv)
    arm, grip = _controllers(cfg, flat, st)
    a = torch.tensor([[2.0, -2.0, 0.0, 0.5, -0.5, 1.0, -1.0]] * 4)
    arm.set_goal(a)
    lo, hi = np.asarray(cfg["output_min"]), np.asarray(cfg["output_max"])
    want = np.clip(a[0].numpy(), -1, 1) * 0.5 * (hi - lo) + 0.5 * (hi + lo)
    assert np.allclose(arm.goal_torque[0].numpy(), want, atol=1e-7)
    mask = torch.tensor([True, False, False, True])
    arm.reset_goal(mask)
    assert (arm.goal_torque[0] == 0).all() and (arm.goal_torque[3] == 0).all() and np.allclose(arm.goal_torque[1].numpy(), want, atol=1e-7)
    for _ in range(7):
        grip.set_goal(torch.ones(4, 1))
    assert np.allclose(grip.current_action.numpy(), np.tile(np.asarray(cfg["grip_sign"]), (4, 1)))      # saturates at +-1 after five steps of 0.2
    grip.reset_goal(mask)
    assert (grip.current_action[0] == 0).all() and (grip.current_action[1] != 0).all()
    with pytest.raises(NotImplementedError):
        BatchedController(st, dict(joints=[0], qpos=[0], qvel=[0]), ([-1.0], [1.0])).run_controller()


class RecordedState(CpuState):
    """A state whose site frames are whatever the test puts there (the recorded inputs of the reference's controller calls)."""

    def __init__(self, B, nq, nv):
        super().__init__(B, nq, nv)
        self.poses = {}

    def site_pose(self, site):
        return self.poses[site]


@pytest.mark.parametrize("tag", ("seed0_gentle", "seed1_full"))
def test_torch_osc_plugin_reproduces_the_reference_controller_on_its_recorded_inputs(tag):
    """TorchOSCController against the reference's OperationalSpaceController: tests/golden/lift_panda_<tag> holds, for each of its 1000 run_controller()
    calls, every input the class read (frames, velocities, Jacobian, mass matrix, bias, joint state, goals) and the torques it returned -- the batch
    dimension of the plugin is used as "one env per recorded call".  Then set_goal(): the goals the reference held after each policy step from the
    frames it saw."""
    from robosuite_amd.controllers import TorchOSCController
    g, cfg, flat = load_golden(tag)
    n = len(g["tau"])
    st = RecordedState(n, flat.nq, flat.nv)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)   # noqa: E731
    cr = np.asarray(flat.actuator_ctrlrange)
    jidx = dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"])
    c = TorchOSCController(st, jidx, (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]), cfg["eef_site"], cfg["base_site"], kp=cfg["kp"], damping_ratio=cfg["damping_ratio"],
                           input_max=cfg["input_max"], input_min=cfg["input_min"], output_max=cfg["output_max"], output_min=cfg["output_min"],
                           uncouple_pos_ori=bool(cfg["uncouple"]), nullspace_kp=cfg.get("nullspace_kp", 10.0))
    c.goal_pos, c.goal_ori, c.initial_joint = f32(g["goal_pos"]), f32(g["goal_ori"]), f32(g["q0"])
    tau = c.torques_from(f32(g["ep"]), f32(g["eR"]), f32(g["ev"]), f32(g["op"]), f32(g["oR"]), f32(g["bv"]), f32(g["J"]), f32(g["M"]), f32(g["bias"]), f32(g["q"]), f32(g["qd"]))
    err = np.abs(tau.numpy() - g["tau"]).max(axis=1) / np.maximum(1.0, np.abs(g["tau"]).max(axis=1))
    assert err.max() < 2e-4, (err.max(), int(err.argmax()))     # fp32 torch against the reference's float64 (the in-kernel law is held to the same bound)
    # set_goal at every policy step: the state the reference saw is that of the first controller call of the step
    n_sub = n // len(g["actions"])
    k0 = np.arange(len(g["actions"])) * n_sub
    st2 = RecordedState(len(k0), flat.nq, flat.nv)
    c2 = TorchOSCController(st2, jidx, (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]), cfg["eef_site"], cfg["base_site"], kp=cfg["kp"], damping_ratio=cfg["damping_ratio"],
                            input_max=cfg["input_max"], input_min=cfg["input_min"], output_max=cfg["output_max"], output_min=cfg["output_min"])
    st2.poses = {cfg["eef_site"]: (f32(g["ep"][k0]), f32(g["eR"][k0])), cfg["base_site"]: (f32(g["op"][k0]), f32(g["oR"][k0]))}
    c2.set_goal(f32(g["actions"][:, :6]))
    assert np.abs(c2.goal_pos.numpy() - g["goal_pos"][k0]).max() < 2e-6 and np.abs(c2.goal_ori.numpy() - g["goal_ori"][k0]).max() < 2e-6


def test_torch_joint_position_controller_reproduces_the_reference_class_call_by_call():
    """TorchJointPositionController against every run_controller() call the reference's JointPositionController made in the env loop of
    tests/golden/lift_panda_ctl_joint_position (state it saw, goal it held, torques it returned); the mass matrix and the bias come from the oracle
    at the recorded state."""
    from robosuite_amd.controllers import TorchJointPositionController
    g, cfg, flat = load_golden("ctl_joint_position")
    om, od, _ = make_oracle(flat)
    n_sub = len(g["sub_qpos"]) // len(g["actions"])
    st = CpuState(2, flat.nq, flat.nv)
    cr = np.asarray(flat.actuator_ctrlrange)
    c = TorchJointPositionController(st, dict(joints=cfg["qpos_idx"], qpos=cfg["qpos_idx"], qvel=cfg["dof_idx"]), (cr[cfg["act_idx"], 0], cr[cfg["act_idx"], 1]),
                                     input_max=cfg["input_max"], input_min=cfg["input_min"], output_max=cfg["output_max"], output_min=cfg["output_min"],
                                     kp=cfg["kp"], damping_ratio=cfg["damping_ratio"])
    assert np.allclose(c.kd.numpy(), cfg["kd"], rtol=1e-6)
    worst = 0.0
    for t, a in enumerate(g["actions"]):
        for s in range(n_sub):
            k = t * n_sub + s
            od.qpos[:] = g["sub_qpos"][k]; od.qvel[:] = g["sub_qvel"][k]; od.forward()
            st.qpos[:] = torch.tensor(g["sub_qpos"][k], dtype=torch.float32); st.qvel[:] = torch.tensor(g["sub_qvel"][k], dtype=torch.float32)
            st.qfrc_bias[:] = torch.tensor(np.array(od.qfrc_bias), dtype=torch.float32)
            st.qM[:] = torch.tensor(od.full_M(), dtype=torch.float32)
            if s == 0:                                           # set_goal at the policy step, from the state of its first substep
                c.set_goal(torch.tensor(np.repeat(a[None, :7], 2, 0), dtype=torch.float32))
            assert np.abs(c.goal_qpos[0].numpy() - g["sub_goal_right"][k]).max() < 2e-6, (t, s)
            tau = c.run_controller()
            e = np.abs(tau[0].numpy() - g["sub_tau_right"][k]).max() / max(1.0, np.abs(g["sub_tau_right"][k]).max())
            worst = max(worst, float(e))
    assert worst < 1e-4, worst                                   # fp32 state against the reference float64 (kp x rounding of q through the mass matrix)
