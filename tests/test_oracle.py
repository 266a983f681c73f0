"""CPU oracle (oracle/rsim_oracle.c) against the committed golden fixtures and analytic identities.

Golden fixtures (tests/golden/*.npz) were produced by tools/gen_golden.py: the UNMODIFIED reference robosuite
(env loop base.py:467-521, controllers osc.py / simple_grip.py, reset code) running in the build container
on top of the mujoco-shaped shim.  Everything under ep/eR/.../tau is output of the reference's own controller
Python, so those tests PIN the controller restatement.  The physics half is pinned only by identities
(no MuJoCo binary exists here: SURVEY.md section 8c, "parity unpinned").
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from robosuite_amd import mjcf
from tests.util import TAGS, load_golden, make_oracle


@pytest.mark.parametrize("tag", TAGS)
def test_osc_torque_law_matches_reference_python(tag):
    """oracle rso_osc_torques == reference OperationalSpaceController.run_controller (osc.py:403-495) on its own inputs."""
    g, cfg, _ = load_golden(tag)
    kp = np.array(cfg["kp"])
    kd = 2 * np.sqrt(kp) * cfg["damping_ratio"]
    worst = 0.0
    for i in range(0, len(g["tau"]), 7):
        tau = O.osc_torques(kp, kd, g["ep"][i], g["eR"][i], g["ev"][i], g["op"][i], g["oR"][i], g["bv"][i], g["goal_pos"][i], g["goal_ori"][i],
                            g["J"][i], g["M"][i], g["bias"][i], g["q"][i], g["qd"][i], g["q0"][i], uncouple=bool(cfg["uncouple"]))
        worst = max(worst, np.abs(tau - g["tau"][i]).max() / max(1.0, np.abs(g["tau"][i]).max()))
    assert worst < 1e-9


@pytest.mark.parametrize("tag", TAGS)
def test_env_step_replay_matches_reference_loop(tag):
    """oracle native substep loop + C controllers == the reference env.step loop driving the same physics (states after every env.step)."""
    g, cfg, flat = load_golden(tag)
    om, od, oc = make_oracle(flat, cfg)
    nq = flat.nq
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0
    od.forward(); oc.reset(od)
    for t in range(len(g["actions"])):
        oc.env_step(od, g["actions"][t], 25)
        # controller clipped outputs written to ctrl (fixed_base_robot.py:149-153)
        assert np.abs(od.ctrl - g["ctrl"][t]).max() < 5e-3 * max(1.0, np.abs(g["ctrl"][t]).max())
        assert np.abs(od.qpos - g["states"][t + 1][1:1 + nq]).max() < 5e-5
        assert np.abs(od.qvel - g["states"][t + 1][1 + nq:]).max() < 5e-4


CTL_TAGS = ("ctl_joint_position", "ctl_joint_torque", "ctl_osc_position",
            # variable-impedance action layouts (osc.py:243-253, joint_pos.py:204-214): [damping_ratio, kp, goal update] / [kp, goal update]
            "ctl_osc_pose_variable", "ctl_osc_pose_variable_kp", "ctl_joint_position_variable",
            # LinearInterpolator (utils/traj_utils.py:25-155) incl. the OSC quirk of using the ramped base-frame goal as a world position
            "ctl_joint_position_linear", "ctl_joint_torque_linear", "ctl_osc_position_linear",
            # + the orientation interpolator of OSC_POSE: error vectors ramped as Euler angles through quaternion slerp (traj_utils.py:129-146)
            "ctl_osc_pose_linear")


@pytest.mark.parametrize("tag", CTL_TAGS)
def test_other_part_controllers_match_reference_loop(tag):
    """JOINT_POSITION (generic/joint_pos.py:200-266), JOINT_TORQUE (generic/joint_tor.py:111-167) and OSC_POSITION (osc.py:255-263, use_ori=False):
    the C restatement replays env.step fixtures recorded with the reference's own controller classes (tools/gen_golden.py --controllers-only)."""
    g, cfg, flat = load_golden(tag)
    om, od, oc = make_oracle(flat, cfg)
    nq = flat.nq
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0
    od.forward(); oc.reset(od)
    from robosuite_amd.backend import gain_dim
    assert g["actions"].shape[1] == len(cfg["input_min"]) + gain_dim(cfg) + 1
    for t in range(len(g["actions"])):
        oc.env_step(od, g["actions"][t], 25)
        # the reference's mat2quat runs a float32 eigh (transform_utils.py:327-350): 1e-7 of quaternion noise per substep on the slerp path
        vtol = 5e-5 if tag == "ctl_osc_pose_linear" else 1e-5
        assert np.abs(od.ctrl - g["ctrl"][t]).max() < 1e-4 * max(1.0, np.abs(g["ctrl"][t]).max())
        assert np.abs(od.qpos - g["states"][t + 1][1:1 + nq]).max() < 1e-6
        assert np.abs(od.qvel - g["states"][t + 1][1 + nq:]).max() < vtol


@pytest.mark.parametrize("tag", ("ctl_joint_position", "ctl_joint_torque", "ctl_joint_velocity",
                                 # Baxter's default: one OSC_POSE object per arm, each around its own "<arm>_center" site (the kernel runs both arm
                                 # parts too: tests/test_hip_parity.py::test_baxter_two_osc_arm_parts_track_the_reference_loop)
                                 "ctl_osc_pose"))
def test_two_arm_joint_space_controllers_match_reference_loop(tag):
    """TwoArmPegInHole / Baxter (BASELINE configs[3] model), one part controller per arm (composite_controller.py:70-121): the oracle loop
    with two controller objects replays the env.step fixture recorded with the reference's own classes.  JOINT_VELOCITY (the type BASELINE
    configs[3] names) was recorded with the reference's own set_goal / run_controller after the constructor defect was patched as SURVEY.md
    prescribes (tools/gen_golden.py patch_joint_velocity_defect)."""
    from oracle.oracle import env_step_parts
    from tests.util import make_oracle_parts
    g, cfg, flat = load_golden(tag, "peg_baxter")
    om, od, parts = make_oracle_parts(flat, cfg)
    nq = flat.nq
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0
    od.forward()
    for c, _ in parts:
        c.reset(od)
    for t in range(len(g["actions"])):
        env_step_parts(od, parts, g["actions"][t], 25)
        assert np.abs(od.ctrl - g["ctrl"][t]).max() < 1e-4 * max(1.0, np.abs(g["ctrl"][t]).max())
        assert np.abs(od.qpos - g["states"][t + 1][1:1 + nq]).max() < 1e-6
        assert np.abs(od.qvel - g["states"][t + 1][1 + nq:]).max() < 1e-5


def test_pickplace_iiwa_robotiq_fixture_replays_on_the_oracle():
    """BASELINE configs[4] model (nv 37; Robotiq140 = 6 finger joints tied by 4 fixed tendons under equality/tendon rows): the C controllers
    (OSC_POSE on the IIWA arm, GRIP with the Robotiq sign table) + oracle loop replay the fixture recorded from the reference env loop, and
    the tendon couplings hold along the way.  CPU only: the fused kernel has no configuration for this model yet."""
    g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
    assert flat.nv == 37 and int(flat.ntendon) == 4 and int(flat.neq) == 4
    om, od, oc = make_oracle(flat, cfg)
    nq = flat.nq
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0
    od.forward(); oc.reset(od)
    assert od.efc_types()[:4] == [4, 4, 4, 4]
    for t in range(len(g["actions"])):
        oc.env_step(od, g["actions"][t], 25)
        assert np.abs(od.ctrl - g["ctrl"][t]).max() < 1e-4 * max(1.0, np.abs(g["ctrl"][t]).max())
        dq = np.abs(od.qpos - g["states"][t + 1][1:1 + nq])
        fingers = np.zeros(nq, dtype=bool); fingers[7:13] = True
        # C vs Python controllers agree to ~1e-6 in ctrl; the undamped 5e-5 kg m^2 finger links under kp = 20 position actuators amplify that
        assert dq[~fingers].max() < 5e-5 and dq[fingers].max() < 5e-3, t
    # the reset pose of the gripper (robotiq_140_gripper.py:26-27) violates its own couplings; the soft equality rows pull the four tendon
    # lengths back towards their reference while the position actuators hold still (zero action = no change of the gripper command)
    def lengths():
        return np.array([sum(flat.wrap_prm[k] * od.qpos[flat.jnt_qposadr[flat.wrap_objid[k]]]
                             for k in range(int(flat.tendon_adr[t]), int(flat.tendon_adr[t]) + int(flat.tendon_num[t]))) for t in range(4)])
    for _ in range(30):
        oc.env_step(od, np.zeros(7), 25)
    assert np.isfinite(od.qpos).all() and np.abs(lengths() - flat.tendon_length0).max() < 0.3


def test_lift_ur5e_robotiq85_fixture_replays_on_the_oracle():
    """Lift / UR5e + Robotiq85 (two spring-loaded fixed tendons with length limits, no equality rows; robotiq_gripper_85.xml:15-29): the oracle
    loop with the C controllers replays the fixture recorded from the reference env loop; the tendon springs act (non-zero passive force)."""
    g, cfg, flat = load_golden("seed0", "lift_ur5e")
    assert int(flat.ntendon) == 2 and int(flat.neq) == 0 and flat.tendon_stiffness.tolist() == [0.4, 0.4]
    om, od, oc = make_oracle(flat, cfg)
    nq = flat.nq
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0
    od.forward(); oc.reset(od)
    fingers = np.zeros(nq, dtype=bool); fingers[cfg["grip_qpos_idx"]] = True
    assert np.abs(od.qfrc_passive[cfg["grip_dof_idx"]]).max() > 1e-3
    for t in range(len(g["actions"])):
        oc.env_step(od, g["actions"][t], 25)
        dq = np.abs(od.qpos - g["states"][t + 1][1:1 + nq])
        assert dq[~fingers].max() < 5e-5 and dq[fingers].max() < 5e-2, t     # undamped finger links under kp = 20 actuators amplify 1e-6 of ctrl


def test_lift_jaco_three_finger_fixture_replays_on_the_oracle():
    """Lift / Jaco + three-finger gripper (jaco_three_finger_gripper.xml:15-42): three fixed tendons, each with an equality/tendon row, a spring,
    a length limit and frictionloss = 0.4 -> friction-loss rows on tendon coefficient rows (after the dof friction rows).  The oracle loop
    replays the fixture recorded from the reference env loop and builds the rows in MuJoCo's order."""
    g, cfg, flat = load_golden("seed0", "lift_jaco")
    assert int(flat.ntendon) == 3 and int(flat.neq) == 3 and flat.tendon_frictionloss.tolist() == [0.4, 0.4, 0.4]
    om, od, oc = make_oracle(flat, cfg)
    nq = flat.nq
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0
    od.forward(); oc.reset(od)
    types = od.efc_types()
    nfd = int((np.asarray(flat.dof_frictionloss) > 0).sum())
    assert types[:3] == [4, 4, 4] and types[3:3 + nfd] == [0] * nfd and types[3 + nfd:6 + nfd] == [6, 6, 6]
    fingers = np.zeros(nq, dtype=bool); fingers[cfg["grip_qpos_idx"]] = True
    for t in range(len(g["actions"])):
        oc.env_step(od, g["actions"][t], 25)
        dq = np.abs(od.qpos - g["states"][t + 1][1:1 + nq])
        assert dq[~fingers].max() < 5e-5 and dq[fingers].max() < 5e-3, t


def test_reset_path_known_answers():
    """values produced by the reference's own reset code (placement_samplers.py:221-309,
    robots/robot.py:247-259, lift.py:311-318) for seed 0, re-derived from the documented draw order."""
    rng = np.random.default_rng(0)
    size = rng.uniform(0.020, 0.022, 3)
    assert size == pytest.approx([0.021273923374642907, 0.02053957342752774, 0.02008194704787239], abs=0)
    init = np.array([0, np.pi / 16.0, 0.00, -np.pi / 2.0 - np.pi / 3.0, 0.00, np.pi - 0.2, np.pi / 4])
    q = init + rng.standard_normal(7) * 0.02
    assert q == pytest.approx([0.002098002343060794, 0.18563615338613984, 0.007231901098189695, -2.591913877088891, 0.018941619262584843,
                               2.927517948873653, 0.7600897339765272], abs=1e-15)
    g, _, flat = load_golden("seed0_gentle")
    assert g["make_qpos"][:7] == pytest.approx(q, abs=1e-12)
    # the fixture's post-reset() state is the SECOND draw block (seed 0)
    assert g["reset_qpos"] == pytest.approx([-0.010885179657146199, 0.19002353772197897, 0.008232610727482657, -2.5971436106026404,
                                             -0.0025706932588806853, 2.968921923000787, 0.772094269927716, 0.020833, -0.020833, 0.008831370694455005,
                                             0.006923106688875233, 0.8303513112412052, 0.3573581645142664, 0, 0, 0.933967420339165], abs=1e-12)
    assert g["cube_size"] == pytest.approx([0.020067171150610928, 0.021459310892859886, 0.02035131124120512], abs=1e-15)


def _random_state(flat, rng, vel=1.0):
    q = flat.qpos0.copy()
    for j in range(flat.njnt):
        a, t = flat.jnt_qposadr[j], flat.jnt_type[j]
        if t in (2, 3):
            lo, hi = flat.jnt_range[j] if flat.jnt_limited[j] else (-1.0, 1.0)
            q[a] = rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo))
        elif t == 0:
            q[a:a + 3] += rng.uniform(-0.02, 0.02, 3) + np.array([0, 0, 0.2])
            quat = rng.standard_normal(4)
            q[a + 3:a + 7] = quat / np.linalg.norm(quat)
    return q, vel * rng.standard_normal(flat.nv)


def test_mass_matrix_symmetric_pd_and_matches_numpy_crba():
    g, cfg, flat = load_golden("seed1_full")
    om, od, _ = make_oracle(flat)
    rng = np.random.default_rng(3)
    for _ in range(5):
        q, v = _random_state(flat, rng)
        od.qpos[:] = q; od.qvel[:] = v; od.forward()
        M = od.full_M()
        assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0
        Mnp, _ = mjcf.mass_matrix_np(flat, q)
        assert np.allclose(M, Mnp, rtol=1e-9, atol=1e-10)


def test_site_jacobian_is_derivative_of_site_position():
    """J(q) qd == d/dt site_xpos along the integrated motion (central difference in q)."""
    g, cfg, flat = load_golden("seed1_full")
    om, od, _ = make_oracle(flat)
    rng = np.random.default_rng(5)
    site = cfg["eef_site"]
    q, _ = _random_state(flat, rng)
    od.qpos[:] = q; od.qvel[:] = 0; od.forward()
    jp, jr = od.jac("site", site)
    eps = 1e-6
    for d in range(7):
        qp, qm = q.copy(), q.copy()
        qp[d] += eps; qm[d] -= eps
        od.qpos[:] = qp; od.forward(); pp = od.site_xpos[3 * site:3 * site + 3].copy()
        od.qpos[:] = qm; od.forward(); pm = od.site_xpos[3 * site:3 * site + 3].copy()
        assert np.allclose((pp - pm) / (2 * eps), jp[:, d], atol=1e-6)


def test_equation_of_motion_residual():
    """M qacc + qfrc_bias == qfrc_passive + qfrc_actuator + qfrc_constraint after forward()."""
    g, cfg, flat = load_golden("seed1_full")
    om, od, _ = make_oracle(flat)
    rng = np.random.default_rng(9)
    for i in (0, 300, 900):
        od.qpos[:] = g["sub_qpos"][i]; od.qvel[:] = g["sub_qvel"][i]; od.ctrl[:] = rng.uniform(-1, 1, flat.nu); od.qacc_warmstart[:] = 0
        od.forward()
        res = od.full_M() @ od.qacc + od.qfrc_bias - od.qfrc_passive - od.qfrc_actuator - od.qfrc_constraint
        assert np.abs(res).max() < 1e-6 * max(1.0, np.abs(od.qfrc_bias).max())


@pytest.mark.parametrize("tag,model", (("seed1_full", "lift_panda"), ("seed0_full", "stack_panda"), ("seed0_full", "pickplace_iiwa")))
def test_primal_newton_and_dual_pgs_solve_the_same_constraint_problem(tag, model):
    """Independent check of the constraint solver: the oracle's primal Newton method (the path's solver, rsim_oracle.c solve_newton) and its
    projected Gauss-Seidel on the dual (solve_pgs: A = J M^-1 J' + R, elliptic cones projected block-wise) are different algorithms for the same
    convex problem; at recorded contact states (cube on the table under the gripper, two stacked cubes, the PickPlace bins with the Robotiq's
    self-contacts and tendon equality rows) their accelerations must coincide."""
    g, cfg, flat = load_golden(tag, model)
    dual = flat.copy()
    dual.arrays["solver"][:] = 0; dual.arrays["iterations"][:] = 50000; dual.arrays["tolerance"][:] = 0
    om, od, _ = make_oracle(flat)
    om2, od2, _ = make_oracle(dual)
    nq = flat.nq
    newton_iters = 0
    for i in (1, 5, 10, 15, len(g["states"]) - 1):
        s = g["states"][i]
        for d in (od, od2):
            d.qpos[:] = s[1:1 + nq]; d.qvel[:] = s[1 + nq:]; d.ctrl[:] = g["ctrl"][min(i, len(g["ctrl"]) - 1)]; d.qacc_warmstart[:] = 0
            d.forward()
        assert od.nefc == od2.nefc and od.nefc > 0
        assert np.abs(od.qacc - od2.qacc).max() < 1e-6 * max(1.0, np.abs(od.qacc).max()), (i, od.solver_iter, od2.solver_iter)
        newton_iters += od.solver_iter
    assert newton_iters >= 5


def test_free_fall_energy_and_momentum():
    """Cube in free flight (no contact, fluid off): linear acceleration = g exactly, angular momentum conserved by Euler to O(h)."""
    g, cfg, flat = load_golden("seed1_full")
    flat = flat.copy()
    flat.arrays["density"][:] = 0; flat.arrays["viscosity"][:] = 0
    om, od, _ = make_oracle(flat)
    q = g["states"][0][1:1 + flat.nq].copy()
    q[11] += 0.5  # lift the cube well above the table
    od.qpos[:] = q; od.qvel[:] = 0; od.qvel[9:12] = [0.1, -0.2, 0.3]; od.qvel[12:15] = [1.0, 2.0, -1.5]
    od.forward()
    assert od.ncon == 0
    assert od.qacc[9:12] == pytest.approx([0, 0, -9.81], abs=1e-9)
    w0 = od.qvel[12:15].copy()
    for _ in range(50):
        od.step()
    assert od.qvel[9:12] == pytest.approx([0.1, -0.2, 0.3 - 9.81 * 0.002 * 50], abs=1e-9)
    # near-isotropic cube inertia: body angular velocity magnitude stays put to first order
    assert np.linalg.norm(od.qvel[12:15]) == pytest.approx(np.linalg.norm(w0), rel=1e-3)


def test_resting_contact_normal_forces_balance_weight():
    """Cube resting on the table: sum of contact normal forces == m g (soft-constraint steady state)."""
    g, cfg, flat = load_golden("seed0_gentle")
    om, od, _ = make_oracle(flat)
    q = g["states"][0][1:1 + flat.nq].copy()
    od.qpos[:] = q; od.qvel[:] = 0
    # hold the arm still with gravity compensation, let the cube settle
    for _ in range(400):
        od.step1()
        od.ctrl[:7] = od.qfrc_bias[:7]
        od.ctrl[7:9] = [0.04, -0.04]
        od.step2()
    od.forward()
    cube_body = flat.name2id("body", "cube_main")
    mass = flat.body_mass[cube_body]
    cube_geoms = {i for i in range(flat.ngeom) if flat.geom_bodyid[i] == cube_body}
    fn = sum(c["normal_force"] for c in od.contacts() if c["geom1"] in cube_geoms or c["geom2"] in cube_geoms)
    assert fn == pytest.approx(mass * 9.81, rel=2e-3)
    assert np.abs(od.qvel[9:15]).max() < 1e-4


def test_gravity_bias_is_the_gradient_of_the_potential_energy():
    """Known answer from mechanics: at rest, qfrc_bias = dV/dq with V = sum_b m_b g z_com,b.  The right-hand side uses only the body frames of the
    forward kinematics and the model's masses / COM offsets; the left-hand side is the recursive Newton-Euler pass."""
    g, cfg, flat = load_golden("seed1_full")
    f2 = flat.copy()
    f2.arrays["density"][:] = 0; f2.arrays["viscosity"][:] = 0
    om, od, _ = make_oracle(f2)
    nq = flat.nq

    def rot(q, v):
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return R @ v

    def potential(q):
        od.qpos[:] = q; od.qvel[:] = 0; od.forward()
        xp, xq = np.array(od.xpos).reshape(-1, 3), np.array(od.xquat).reshape(-1, 4)
        return sum(flat.body_mass[b] * 9.81 * (xp[b] + rot(xq[b], flat.body_ipos[b]))[2] for b in range(flat.nbody))

    for i in (0, 5, 30):
        q0 = g["states"][i][1:1 + nq].copy()
        od.qpos[:] = q0; od.qvel[:] = 0; od.forward()
        bias = np.array(od.qfrc_bias).copy()
        for j in range(9):                      # arm hinges and finger slides (qpos index = dof index for these)
            qp, qm = q0.copy(), q0.copy()
            qp[j] += 1e-6; qm[j] -= 1e-6
            assert bias[j] == pytest.approx((potential(qp) - potential(qm)) / 2e-6, abs=2e-6 * max(1.0, np.abs(bias).max())), (i, j)


def test_energy_is_conserved_to_first_order_in_the_timestep():
    """Known answer from mechanics: without damping, friction loss, fluid or actuation the arm falling under gravity trades potential for kinetic
    energy, E = 1/2 qd' M qd + sum m g z.  Semi-implicit Euler keeps E to O(h): the drift is a percent of the kinetic energy at h = 2 ms and halves
    with the step.  Ties the Coriolis / centrifugal terms of the bias force to the mass matrix and the kinematics."""
    g, cfg, flat = load_golden("seed1_full")
    nq = flat.nq

    def rot(q, v):
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return R @ v

    drift = {}
    for h, n in ((0.002, 150), (0.001, 300)):
        f2 = flat.copy()
        for k in ("density", "viscosity", "dof_damping", "dof_frictionloss"):
            f2.arrays[k][:] = 0
        f2.arrays["timestep"][:] = h
        om, od, _ = make_oracle(f2)

        def energy():
            od.forward()
            xp, xq, v = np.array(od.xpos).reshape(-1, 3), np.array(od.xquat).reshape(-1, 4), np.array(od.qvel)
            V = sum(flat.body_mass[b] * 9.81 * (xp[b] + rot(xq[b], flat.body_ipos[b]))[2] for b in range(flat.nbody))
            return 0.5 * v @ od.full_M() @ v, V

        od.qpos[:] = g["states"][0][1:1 + nq]; od.qvel[:] = 0; od.ctrl[:] = 0
        T0, V0 = energy()
        for _ in range(n):
            od.step1()
            od.ctrl[7:9] = od.qpos[7:9]          # the fingers' position actuators stay unloaded
            od.step2()
        T, V = energy()
        assert T > 5.0 and V0 - V == pytest.approx(T, rel=0.02)          # 6 J moved from potential to kinetic
        drift[h] = (T + V) - (T0 + V0)
    assert abs(drift[0.002]) < 0.015 * 6.0
    assert drift[0.002] / drift[0.001] == pytest.approx(2.0, rel=0.2)


def test_friction_cone_sticks_below_and_slides_above_the_coulomb_limit():
    """Known answer from mechanics, not from any simulator: with gravity tilted by theta about the table, the cube (friction mu = 0.3 against the
    table, elliptic cone, impratio 20) stays put while tan(theta) < mu and slides with a = g (sin(theta) - mu cos(theta)) above it."""
    g, cfg, flat = load_golden("seed0_gentle")
    cube, table = flat.name2id("geom", "cube_g0"), flat.name2id("geom", "table_collision")
    mu = 0.3
    for fac, slides in ((0.5, False), (0.9, False), (1.2, True), (2.0, True)):
        th = np.arctan(fac * mu)
        f2 = flat.copy()
        f2.arrays["gravity"][:] = 9.81 * np.array([np.sin(th), 0.0, -np.cos(th)]); f2.arrays["density"][:] = 0; f2.arrays["viscosity"][:] = 0
        f2.arrays["geom_friction"][cube][0] = mu; f2.arrays["geom_friction"][table][0] = mu
        om, od, _ = make_oracle(f2)
        od.qpos[:] = g["states"][0][1:1 + flat.nq]; od.qvel[:] = 0
        vx = []
        for _ in range(150):
            od.step1()
            od.ctrl[:7] = od.qfrc_bias[:7]; od.ctrl[7:9] = [0.04, -0.04]      # arm held by gravity compensation, fingers open
            od.step2()
            vx.append(od.qvel[9])
        if not slides:
            assert abs(vx[-1]) < 1e-3 and od.ncon == 4, (fac, vx[-1])          # soft contacts creep at ~1e-5 m/s, they do not slide
        else:
            a = (vx[-1] - vx[-51]) / (50 * 0.002)
            assert a == pytest.approx(9.81 * (np.sin(th) - mu * np.cos(th)), rel=0.05), (fac, a)


def test_scripted_grasp_lifts_the_cube():
    """Behavioural anchor of the contact / friction model (reference shape: GripperTester raises unless the cube is lifted,
    models/grippers/gripper_tester.py:204-226 via tests/test_grippers/test_panda_gripper.py): hover, descend, close, lift."""
    from tests.util import scripted_grasp_and_lift
    g, cfg, flat = load_golden("seed1_full")
    acts, qs, cube_z, od = scripted_grasp_and_lift(flat, cfg, g["states"][0][1:1 + flat.nq])
    assert cube_z > 0.8 + 0.04 + 0.1                      # Lift._check_success threshold cleared with margin
    pads = {flat.names["geom"].index(n) for n in ("gripper0_right_finger1_pad_collision", "gripper0_right_finger2_pad_collision")}
    cube = flat.names["geom"].index("cube_g0")
    touching = {c["geom1"] if c["geom2"] == cube else c["geom2"] for c in od.contacts() if cube in (c["geom1"], c["geom2"])}
    assert pads <= touching                               # held by both finger pads (ManipulationEnv._check_grasp)


def _documented_impedance(solimp, r):
    """d(r) of MuJoCo's documentation (Computation > Soft constraints > solimp = d0, dwidth, width, midpoint, power), written from the text:
    x = |r| / width; y(x) = a x^p for x <= midpoint, 1 - b (1 - x)^p above, a = 1 / midpoint^(p-1), b = 1 / (1 - midpoint)^(p-1); d = d0 + y (dwidth - d0)."""
    d0, dw, width, mid, p = solimp
    x = min(1.0, abs(r) / width)
    if x >= 1.0:
        return dw
    y = x if p == 1 else (x ** p / mid ** (p - 1) if x <= mid else 1.0 - (1.0 - x) ** p / (1.0 - mid) ** (p - 1))
    return d0 + y * (dw - d0)


@pytest.mark.parametrize("solimp,depth", (((0.9, 0.95, 0.001, 0.5, 2.0), 0.0005), ((0.9, 0.95, 0.001, 0.5, 2.0), 0.0002), ((0.9, 0.95, 0.001, 0.5, 2.0), 0.0008),
                                          ((0.8, 0.99, 0.002, 0.3, 3.0), 0.0011), ((0.9, 0.95, 0.001, 0.5, 1.0), 0.0004), ((0.9, 0.95, 0.001, 0.5, 2.0), 0.003)))
def test_documented_constraint_model_known_answers(solimp, depth):
    """The soft-constraint quantities the solver consumes, against MuJoCo's DOCUMENTED formulas evaluated by hand here (no simulator involved; these
    were only exercised end to end before).  A cube pushed `depth` into the table while moving down at 0.1 m/s:
      impedance d(r) from solimp; solref = (timeconst, dampratio) -> b = 2 / (dmax timeconst), k = d(r) / (dmax^2 timeconst^2 dampratio^2);
      reference acceleration aref = -b v - k r;  regulariser R = (1 - d) / d x diagApprox, D = 1 / R, diagApprox of a contact's normal row =
      translational inverse weights of the two bodies (1 / m for a free body, 0 for the world); the friction rows of an elliptic cone carry
      R / impratio (impratio = 20, models/assets/base.xml:4).
    Also a joint friction-loss row (r = 0: d = d0, aref = -b qd) and a violated joint limit (finger slide 1 mm below its range)."""
    g, cfg, flat = load_golden("seed0_gentle")
    f2 = flat.copy()
    cube, table = flat.name2id("geom", "cube_g0"), flat.name2id("geom", "table_collision")
    solref = (0.02, 1.0)
    for gg in (cube, table):
        f2.arrays["geom_solref"][gg] = solref; f2.arrays["geom_solimp"][gg] = solimp
    om, od, _ = make_oracle(f2)
    q = g["states"][0][1:1 + flat.nq].copy()
    half = flat.geom_size[cube]
    q[12:16] = (1, 0, 0, 0)
    q[11] = 0.8 + half[2] - depth                    # table top at z = 0.8 (lift.py:153): the four bottom corners are `depth` inside
    q[7] = -0.001                                    # finger_joint1 (range [0, 0.04]) one millimetre past its lower limit
    v = np.zeros(flat.nv); v[11] = -0.1; v[3] = 0.7; v[7] = -0.02
    od.qpos[:] = q; od.qvel[:] = v; od.forward()
    types = od.efc_types()
    R, D, aref, pos = np.array(od.efc_R), np.array(od.efc_D), np.array(od.efc_aref), np.array(od.efc_pos)
    # ---- contact rows: four corner contacts, condim 3, elliptic: rows (normal, tangent, tangent) per contact
    con = [c for c in od.contacts() if {c["geom1"], c["geom2"]} == {cube, table}]
    assert len(con) == 4 and all(c["dim"] == 3 for c in con)
    mass = 1000.0 * 8 * half[0] * half[1] * half[2]
    d = _documented_impedance(solimp, depth)
    dmax = solimp[1]
    b, k = 2.0 / (dmax * solref[0]), d / (dmax ** 2 * solref[0] ** 2 * solref[1] ** 2)
    for c in con:
        r0 = c["efc_address"]
        assert types[r0] == 3 and pos[r0] == pytest.approx(-depth, abs=1e-12)
        Rn = (1.0 - d) / d * (1.0 / mass)
        assert R[r0] == pytest.approx(Rn, rel=1e-9) and D[r0] == pytest.approx(1.0 / Rn, rel=1e-9)
        assert aref[r0] == pytest.approx(-b * (-0.1) - k * (-depth), rel=1e-9)
        assert R[r0 + 1] == pytest.approx(Rn / 20.0, rel=1e-9) and R[r0 + 2] == pytest.approx(Rn / 20.0, rel=1e-9)
    # ---- friction-loss row of arm joint 4 (dof 3): r = 0 -> d = d0 of the dof's solimp, aref = -b qd
    fr = [i for i, t in enumerate(types) if t == 0]
    assert len(fr) == 9
    ds, dr = flat.dof_solimp[3], flat.dof_solref[3]
    d0 = _documented_impedance(ds, 0.0)
    assert R[fr[3]] == pytest.approx((1 - d0) / d0 * flat.dof_invweight0[3], rel=1e-9)
    assert aref[fr[3]] == pytest.approx(-2.0 / (ds[1] * max(dr[0], 2 * 0.002)) * 0.7, rel=1e-9)
    # ---- the violated lower limit of the finger slide: r = q - lower = -1 mm, J = +1
    lim = [i for i, t in enumerate(types) if t == 1]
    assert len(lim) == 1 and pos[lim[0]] == pytest.approx(-0.001, abs=1e-12)
    js, jr = flat.jnt_solimp[7], flat.jnt_solref[7]
    dl = _documented_impedance(js, 0.001)
    tc = max(jr[0], 2 * 0.002)
    assert R[lim[0]] == pytest.approx((1 - dl) / dl * flat.dof_invweight0[7], rel=1e-9)
    assert aref[lim[0]] == pytest.approx(-2.0 / (js[1] * tc) * (-0.02) - dl / (js[1] ** 2 * tc ** 2 * jr[1] ** 2) * (-0.001), rel=1e-9)


# ---- force / torque sensors (mj_sensorAcc -> mj_rnePostConstraint; robots/robot.py:739-751, 795-815) --------------------------------------------
def _subtree(flat, b):
    out = []
    for i in range(flat.nbody):
        k = i
        while k > 0 and k != b:
            k = int(flat.body_parentid[k])
        if k == b and b > 0:
            out.append(i)
    return out


def _sensor_site(flat):
    st, so = flat.arrays["sensor_type"], flat.arrays["sensor_objid"]
    assert list(st[:2]) == [0, 1] and so[0] == so[1]          # <force site="ft_frame"/> <torque site="ft_frame"/> of the gripper XMLs
    return int(so[0])


def test_force_torque_sensor_reads_the_weight_it_carries_when_nothing_accelerates():
    """Known answer from statics: with every generalised force cancelled (qacc = 0, qvel = 0) the gripper's wrist sensor must read the weight of
    everything outboard of it, as the support force +m g z and its moment about the sensor site, both in the site frame."""
    g, cfg, flat = load_golden("seed0_gentle")
    om, od, _ = make_oracle(flat)
    od.qpos[:] = g["states"][0][1:1 + flat.nq]; od.qvel[:] = 0; od.ctrl[:] = 0; od.qacc_warmstart[:] = 0
    od.forward()
    od.qfrc_applied[:] = -(np.array(od.qfrc_passive) - np.array(od.qfrc_bias) + np.array(od.qfrc_actuator))
    od.forward()
    robot = np.concatenate([cfg["dof_idx"], cfg["grip_dof_idx"]])
    assert np.abs(np.array(od.qacc)[robot]).max() < 1e-8
    site = _sensor_site(flat)
    sub = _subtree(flat, int(flat.site_bodyid[site]))
    mass = np.asarray(flat.body_mass)[sub]
    assert len(sub) >= 3 and mass.sum() == pytest.approx(float(flat.body_subtreemass[flat.site_bodyid[site]]), rel=1e-12)
    com = (np.array(od.xipos).reshape(-1, 3)[sub] * mass[:, None]).sum(0) / mass.sum()
    R, P = np.array(od.site_xmat).reshape(-1, 3, 3)[site], np.array(od.site_xpos).reshape(-1, 3)[site]
    W = mass.sum() * np.array([0, 0, 9.81])
    s = np.array(od.sensordata)
    assert s[:3] == pytest.approx(R.T @ W, abs=1e-9) and s[3:6] == pytest.approx(R.T @ np.cross(com - P, W), abs=1e-9)
    assert np.linalg.norm(s[:3]) == pytest.approx(mass.sum() * 9.81, rel=1e-12)


def test_force_torque_sensor_equals_momentum_rate_minus_gravity_and_contacts_while_holding_the_cube():
    """Newton-Euler for the bodies outboard of the sensor, from positions alone: the wrench the wrist transmits = d/dt (linear, angular momentum about
    the sensor site) of those bodies - their weight - the contact wrenches acting on them.  Momenta come from central differences of the
    oracle's KINEMATICS along q(t) = q + v t + a t^2 / 2 (no velocity or force quantity of the oracle enters), contact forces from the solver's
    rows.  State: the scripted grasp, cube between the pads and moving up -- pad contacts (external to the subtree) carry the cube."""
    from tests.util import scripted_grasp_and_lift
    g, cfg, flat = load_golden("seed1_full")
    acts, qs, cube_z, od = scripted_grasp_and_lift(flat, cfg, g["states"][0][1:1 + flat.nq])
    od.forward()
    site = _sensor_site(flat)
    sub = _subtree(flat, int(flat.site_bodyid[site]))
    P = np.array(od.site_xpos).reshape(-1, 3)[site].copy()
    R = np.array(od.site_xmat).reshape(-1, 3, 3)[site].copy()
    q0, v0, a0 = np.array(od.qpos).copy(), np.array(od.qvel).copy(), np.array(od.qacc).copy()
    scalar = [(int(flat.jnt_qposadr[j]), int(flat.jnt_dofadr[j])) for j in range(flat.njnt) if flat.jnt_type[j] in (2, 3)]
    _, ok, _ = make_oracle(flat)
    mass, inertia = np.asarray(flat.body_mass), np.asarray(flat.body_inertia).reshape(-1, 3)

    def frames(t):
        q = q0.copy()
        for qa, da in scalar:                      # the free cube is not outboard of the wrist: its pose may stay put
            q[qa] = q0[qa] + v0[da] * t + 0.5 * a0[da] * t * t
        ok.qpos[:] = q; ok.qvel[:] = 0; ok.forward()
        return np.array(ok.xipos).reshape(-1, 3)[sub].copy(), np.array(ok.ximat).reshape(-1, 3, 3)[sub].copy()

    def momenta(t, d=2e-4):
        (xp, Rp), (xm, Rm), (x, Rm0) = frames(t + d), frames(t - d), frames(t)
        lin, ang = np.zeros(3), np.zeros(3)
        for k, b in enumerate(sub):
            v = (xp[k] - xm[k]) / (2 * d)
            S = Rp[k] @ Rm[k].T                   # rotation over 2 d: axis-angle
            w = np.array([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]]) / 2.0
            n = np.linalg.norm(w)
            w = w * (np.arcsin(min(1.0, n)) / n if n > 1e-300 else 1.0) / (2 * d)
            lin += mass[b] * v
            ang += np.cross(x[k] - P, mass[b] * v) + Rm0[k] @ (inertia[b] * (Rm0[k].T @ w))
        return lin, ang

    e = 2e-4
    (lp, ap), (lm, am) = momenta(e), momenta(-e)
    F, T = (lp - lm) / (2 * e), (ap - am) / (2 * e)
    x0, _ = frames(0.0)
    for k, b in enumerate(sub):                    # minus the weight
        F -= mass[b] * np.array([0, 0, -9.81]); T -= np.cross(x0[k] - P, mass[b] * np.array([0, 0, -9.81]))
    n_ext = 0
    for c in od.contacts():                        # minus the contact wrenches on outboard bodies (contacts between two of them cancel)
        if c["efc_address"] < 0:
            continue
        f = np.array(od.efc_force)[c["efc_address"]:c["efc_address"] + c["dim"]]
        fr = np.array(c["frame"]).reshape(3, 3)
        fw = fr[:min(3, c["dim"])].T @ f[:3]
        tw = fr[:c["dim"] - 3].T @ f[3:] if c["dim"] > 3 else np.zeros(3)
        for gid, sgn in ((c["geom1"], -1.0), (c["geom2"], 1.0)):
            if int(flat.geom_bodyid[gid]) in sub:
                F -= sgn * fw; T -= sgn * (tw + np.cross(np.array(c["pos"]) - P, fw)); n_ext += 1
    s = np.array(od.sensordata)
    assert n_ext >= 2 and np.abs(a0).max() > 0.1                         # pads on the cube, and the state is not static
    assert R @ s[:3] == pytest.approx(F, abs=2e-5 * max(1.0, np.abs(F).max())), (R @ s[:3], F)
    assert R @ s[3:6] == pytest.approx(T, abs=2e-5 * max(1.0, np.abs(T).max())), (R @ s[3:6], T)
    # and it carries more than the gripper alone: the cube's weight comes in through the pads
    assert (R @ s[:3])[2] > (mass[sub].sum() + 0.5 * float(flat.body_mass[flat.names["body"].index("cube_main")])) * 9.81


def test_pickplace_single_object_mode_1_reset_path_reproduces_the_reference_episodes():
    """PickPlaceSingle draws the episode's object with rng.choice over a SET of the object names (pick_place.py:716-722): one rng.integers(0, 4) after
    the placement draws, mapped through the set order of the recording process (cfg["task"]["mode1_order"]).  Block 0 of the env generator is
    make()'s own reset, block e + 1 the e-th user reset: objects and qpos of four recorded episodes."""
    from robosuite_amd import pick_place
    g, cfg, flat = load_golden("seed3", "pickplace_single_iiwa")
    t = cfg["task"]
    rng = np.random.default_rng(3)
    for block in range(1 + len(g["ep_object"])):
        d = pick_place.reset_draws(rng, t["placement"], True)
        obj = pick_place.active_object(t, d)
        q = pick_place.initial_qpos(d, t["placement"], flat.nq, obj)
        ref_obj, ref_q = (int(g["make_object"]), g["make_qpos"]) if block == 0 else (int(g["ep_object"][block - 1]), g["ep_reset_qpos"][block - 1])
        assert obj == ref_obj and q == pytest.approx(ref_q, abs=1e-12), block
        assert np.abs(pick_place.episode_setup(cfg, flat.nq, 3, [0], block)[0] - ref_q).max() < 1e-12
    away = [o["qposadr"] for i, o in enumerate(t["placement"]["objects"]) if i != obj]
    assert all(q[a] == 10.0 for a in away)


def _documented_spring(solref, dmax, dt):
    tc = max(solref[0], 2 * dt)
    return 1.0 / (dmax * dmax * tc * tc * solref[1] * solref[1])


@pytest.mark.parametrize("solref,solimp", (((0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2)), ((0.01, 0.7), (0.8, 0.8, 0.002, 0.5, 2)), ((0.003, 1.0), (0.95, 0.99, 0.0005, 0.3, 3))))
def test_resting_depths_follow_from_the_documented_constraint_model(solref, solimp):
    """End-to-end known answer of the soft-constraint model (MuJoCo documentation, "Computation: soft constraint model"), no simulator involved: a single
    constraint at rest carries the load a0 it opposes, and with R = (1 - d) / d x A its reference acceleration a_ref = -k d r must equal
    -a0 (1 - d) / d, i.e. the penetration solves r = a0 (1 - d(r)) / (k d(r)^2) with k from solref (refsafe: the time constant is at least 2 dt) and d(r)
    the impedance curve.  (a) a sphere resting on a plane under gravity (one frictional contact: its normal row), (b) a pendulum lying against its joint
    limit under a constant torque."""
    dt, g0 = 0.002, 9.81
    k = _documented_spring(solref, solimp[1], dt)

    def depth(a0):
        r = a0 / k
        for _ in range(200):
            d = _documented_impedance(solimp, r)
            r = a0 * (1 - d) / (k * d * d)
        return r

    sr, si = " ".join(map(str, solref)), " ".join(map(str, solimp))
    xml = f"""<mujoco><option timestep="{dt}" cone="elliptic"/><worldbody><geom name="floor" type="plane" size="1 1 0.1" solref="{sr}" solimp="{si}"/>
              <body name="ball" pos="0 0 0.05"><freejoint/><geom name="ball" type="sphere" size="0.05" density="1000" solref="{sr}" solimp="{si}"/></body></worldbody></mujoco>"""
    om, od, _ = make_oracle(mjcf.compile_mjcf(xml))
    od.qpos[:] = om.field("qpos0"); od.qvel[:] = 0
    for _ in range(3000):
        od.step()
    od.forward()
    assert np.abs(od.qvel).max() < 1e-7 and od.ncon == 1
    assert 0.05 - od.qpos[2] == pytest.approx(depth(g0), rel=1e-4)
    assert od.contacts()[0]["normal_force"] == pytest.approx(1000 * 4 / 3 * np.pi * 0.05**3 * g0, rel=1e-6)
    # (b) hinge about y with its limit at 0.3 rad, pushed into it by gravity on an off-axis mass
    xml = f"""<mujoco><compiler angle="radian"/><option timestep="{dt}" gravity="0 0 0"/><worldbody><body name="p"><joint name="h" type="hinge" axis="0 1 0" range="-0.3 0.3" solreflimit="{sr}" solimplimit="{si}"/>
              <geom type="sphere" size="0.02" pos="0.2 0 0" mass="0.5" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    flat = mjcf.compile_mjcf(xml)
    om, od, _ = make_oracle(flat)
    tau = 0.4
    od.qpos[0] = 0.25; od.qvel[:] = 0
    for _ in range(6000):
        od.qfrc_applied[0] = tau
        od.step()
    od.forward()
    inertia = float(od.full_M()[0, 0])
    assert abs(od.qvel[0]) < 1e-7 and od.nefc == 1
    assert od.qpos[0] - 0.3 == pytest.approx(depth(tau / inertia), rel=1e-4)


def test_joint_friction_loss_damping_and_torsional_friction_known_answers():
    """Three more closed forms of the documented model, each on a one-body system:
    (a) frictionloss F on a hinge: under a torque below F the soft row lets the joint creep at tau R / b (closed form below), a torque above accelerates it
        with (tau - F) / I (a bounded-force constraint);
    (b) joint damping b integrated implicitly in velocity: one step takes v to v I / (I + h b) exactly (Euler with implicit damping);
    (c) torsional friction (condim 4, second friction coefficient, a length): a ball spinning about the contact normal on a plane slips in torsion, so
        the constraint force lies ON the elliptic cone (|f_torsion| = mu_t f_normal), and the ball's accelerations are Newton's and Euler's for exactly
        those forces."""
    h = 0.002
    pend = """<mujoco><compiler angle="radian"/><option timestep="%g" gravity="0 0 0"/><worldbody><body><joint type="hinge" axis="0 0 1" %s/>
              <geom type="box" size="0.1 0.02 0.02" mass="1.5" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    # (a)
    om, od, _ = make_oracle(mjcf.compile_mjcf(pend % (h, 'frictionloss="0.3"')))
    od.forward(); inertia = float(od.full_M()[0, 0])
    od.qfrc_applied[0] = 0.25
    for _ in range(400):
        od.step()
    # "rest" of a SOFT constraint: the row is in its quadratic zone, force = -(b v + a) / R with R = (1 - d0) / d0 x (1 / I) and b = 2 / (dmax tc) (no
    # position term: friction has no reference position), so the joint creeps at the velocity where that force balances the torque: v = tau R / b
    creep = 0.25 * (0.1 / 0.9) * (1.0 / inertia) / (2.0 / (0.95 * 0.02))
    assert od.qvel[0] == pytest.approx(creep, rel=1e-6) and abs(od.qacc[0]) < 1e-9
    od.qfrc_applied[0] = 0.8; od.forward()
    assert od.qacc[0] == pytest.approx((0.8 - 0.3) / inertia, rel=2e-3)        # the soft bound is reached up to the constraint's regularisation
    # (b)
    b = 0.7
    om, od, _ = make_oracle(mjcf.compile_mjcf(pend % (h, 'damping="%g"' % b)))
    od.qvel[0] = 2.0
    v = 2.0
    for _ in range(50):
        od.step(); v *= inertia / (inertia + h * b)
        assert od.qvel[0] == pytest.approx(v, rel=1e-12)
    # (c)
    mu_t, r = 0.02, 0.05
    xml = f"""<mujoco><option timestep="{h}" cone="elliptic" impratio="1"/><worldbody><geom type="plane" size="1 1 0.1" condim="4" friction="1 {mu_t} 0.0001"/>
              <body name="ball" pos="0 0 {r}"><freejoint/><geom type="sphere" size="{r}" density="1000" condim="4" friction="1 {mu_t} 0.0001"/></body></worldbody></mujoco>"""
    om, od, _ = make_oracle(mjcf.compile_mjcf(xml))
    od.qpos[:] = om.field("qpos0")
    for _ in range(1500):                                                     # settle into the resting depth first
        od.step()
    m_ball = 1000 * 4 / 3 * np.pi * r**3
    od.qvel[:] = 0; od.qvel[5] = 30.0                                          # spin about z (world = body frame at rest)
    od.forward()
    c = od.contacts()[0]
    f = np.array(od.efc_force)[c["efc_address"]:c["efc_address"] + c["dim"]]
    assert c["dim"] == 4 and abs(f[1]) < 1e-9 and abs(f[2]) < 1e-9            # no translational slip
    assert abs(f[3]) == pytest.approx(mu_t * f[0], rel=1e-9)                   # slipping in torsion: the force sits ON the elliptic cone, |f_torsion| = mu_t f_normal
    a = np.array(od.qacc)
    assert a[5] == pytest.approx(f[3] / (0.4 * m_ball * r * r), rel=1e-6)      # Euler's equation about the contact normal
    assert m_ball * a[2] == pytest.approx(f[0] - m_ball * 9.81, rel=1e-6)      # Newton's, along it (the cone couples them: slipping raises the normal force)
    assert np.abs(a[[0, 1, 3, 4]]).max() < 1e-6


def planar_2r_xml(l1=0.4, lc1=0.15, lc2=0.12, m1=1.2, m2=0.8, I1=0.02, I2=0.01):
    return f"""<mujoco><compiler angle="radian"/><option timestep="0.002" gravity="0 -9.81 0"/><worldbody>
      <body name="l1"><joint name="q1" type="hinge" axis="0 0 1"/><inertial pos="{lc1} 0 0" mass="{m1}" diaginertia="0.001 {I1} {I1}"/>
        <body name="l2" pos="{l1} 0 0"><joint name="q2" type="hinge" axis="0 0 1"/><inertial pos="{lc2} 0 0" mass="{m2}" diaginertia="0.001 {I2} {I2}"/></body></body></worldbody></mujoco>"""


def planar_2r_closed_form(q, qd, l1=0.4, lc1=0.15, lc2=0.12, m1=1.2, m2=0.8, I1=0.02, I2=0.01, g=9.81):
    """Mass matrix and bias (Coriolis / centrifugal + gravity) of the planar two-link arm, the textbook example (e.g. Spong, Hutchinson, Vidyasagar,
    "Robot Modeling and Control", planar elbow manipulator): angles from the x axis, gravity along -y."""
    c2, s2 = np.cos(q[1]), np.sin(q[1])
    m12 = m2 * (lc2**2 + l1 * lc2 * c2) + I2
    M = np.array([[m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2 * c2) + I1 + I2, m12], [m12, m2 * lc2**2 + I2]])
    hh = -m2 * l1 * lc2 * s2
    cor = np.array([hh * (2 * qd[0] * qd[1] + qd[1]**2), -hh * qd[0]**2])
    grav = np.array([(m1 * lc1 + m2 * l1) * g * np.cos(q[0]) + m2 * lc2 * g * np.cos(q[0] + q[1]), m2 * lc2 * g * np.cos(q[0] + q[1])])
    return M, cor + grav


def test_planar_two_link_arm_has_the_textbook_mass_matrix_and_bias():
    """CRBA and RNE against the closed-form dynamics of the planar elbow manipulator at random states (machine precision)."""
    om, od, _ = make_oracle(mjcf.compile_mjcf(planar_2r_xml()))
    rng = np.random.default_rng(0)
    for _ in range(8):
        q, qd = rng.uniform(-2, 2, 2), rng.uniform(-3, 3, 2)
        od.qpos[:] = q; od.qvel[:] = qd; od.forward()
        M, bias = planar_2r_closed_form(q, qd)
        assert od.full_M() == pytest.approx(M, abs=1e-14) and np.array(od.qfrc_bias) == pytest.approx(bias, abs=1e-13)
        assert np.array(od.qacc) == pytest.approx(np.linalg.solve(M, -bias), abs=1e-10)      # unforced: M qacc + bias = 0


# ---- narrow phase against geometry (the contact contract of the documentation: dist < 0 in penetration, the normal points from geom1 to geom2 -- pairs are
# ordered by geom type, lower type first --, the position is the midpoint between the two surfaces) ------------------------------------------------------------
def narrow_phase_scene(bodies):
    b = "".join('<body name="b%d" pos="%s" %s><freejoint/><geom name="g%d" %s/></body>' % (i, p, ('quat="%s"' % q) if q else "", i, g) for i, (p, q, g) in enumerate(bodies))
    return f'<mujoco><compiler angle="radian"/><option timestep="0.002"/><worldbody><geom name="floor" type="plane" size="2 2 0.1"/>{b}</worldbody></mujoco>'


def narrow_phase_cases():
    """(name, scene, expected contacts as (dist, pos, normal) in any order) -- every number below is elementary geometry."""
    z = np.array([0.0, 0.0, 1.0])
    c30, s30 = np.cos(np.pi / 6), np.sin(np.pi / 6)
    qz = f"{np.cos(np.pi / 12)} 0 0 {np.sin(np.pi / 12)}"                     # 30 degrees about z
    qx, qy = f"{np.cos(np.pi / 4)} {np.sin(np.pi / 4)} 0 0", f"{np.cos(np.pi / 4)} 0 {np.sin(np.pi / 4)} 0"
    d = np.array([0.06, 0.03, 0.04]); dn = np.linalg.norm(d)
    corners = [np.array([sx * 0.03 * c30 - sy * 0.03 * s30, sx * 0.03 * s30 + sy * 0.03 * c30, 1.0495]) for sx in (-1, 1) for sy in (-1, 1)]
    return [
        ("sphere on the plane, 1 mm deep", [("0.1 0.2 0.049", None, 'type="sphere" size="0.05"')], [(-0.001, [0.1, 0.2, -0.0005], z)]),
        ("two spheres", [("0 0 1", None, 'type="sphere" size="0.05"'), ("0.06 0.03 1.04", None, 'type="sphere" size="0.04"')],
         [(dn - 0.09, np.array([0, 0, 1.0]) + d / dn * (0.05 + 0.5 * (dn - 0.09)), d / dn)]),
        ("sphere on a box face (sphere is geom1: lower type)", [("0 0 1", None, 'type="box" size="0.1 0.1 0.05"'), ("0.02 -0.03 1.089", None, 'type="sphere" size="0.04"')],
         [(-0.001, [0.02, -0.03, 1.0495], -z)]),
        ("box on the plane: its four lower corners", [("0 0 0.0495", None, 'type="box" size="0.05 0.03 0.05"')],
         [(-0.0005, [sx * 0.05, sy * 0.03, -0.00025], z) for sx in (-1, 1) for sy in (-1, 1)]),
        ("small box turned by 30 degrees on a big one: its four lower corners", [("0 0 1", None, 'type="box" size="0.1 0.1 0.05"'), ("0 0 1.079", qz, 'type="box" size="0.03 0.03 0.03"')],
         [(-0.001, c, z) for c in corners]),
        ("two crossed capsules", [("0 0 1", qx, 'type="capsule" size="0.02 0.1"'), ("0 0 1.035", qy, 'type="capsule" size="0.02 0.1"')], [(-0.005, [0, 0, 1.0175], z)]),
    ]


def check_narrow_phase(contacts, expected, tol_d, tol_p, tol_n, name):
    assert len(contacts) == len(expected), (name, len(contacts))
    left = list(expected)
    for c in contacts:
        k = int(np.argmin([np.linalg.norm(np.asarray(c["pos"]) - np.asarray(e[1])) for e in left]))
        dist, pos, nrm = left.pop(k)
        assert abs(c["dist"] - dist) < tol_d, (name, c["dist"], dist)
        assert np.abs(np.asarray(c["pos"]) - np.asarray(pos)).max() < tol_p, (name, c["pos"], pos)
        assert np.abs(np.asarray(c["frame"]).reshape(3, 3)[0] - np.asarray(nrm)).max() < tol_n, (name, c["frame"], nrm)


def test_narrow_phase_against_elementary_geometry():
    """plane-convex, plane-box, box-box and the MPR path (sphere-sphere, sphere-box, capsule-capsule) of the oracle: depths, midpoints and normals."""
    for name, bodies, expected in narrow_phase_cases():
        flat = mjcf.compile_mjcf(narrow_phase_scene(bodies))
        om, od, _ = make_oracle(flat)
        od.qpos[:] = om.field("qpos0"); od.forward()
        mpr = "sphere" in name and "plane" not in name or "capsule" in name        # iterative (portal refinement to 1e-6); the other three are direct
        check_narrow_phase(od.contacts(), expected, 1e-6 if mpr else 1e-9, 1e-4 if mpr else 1e-8, 1e-5 if mpr else 1e-8, name)


# ---- mesh hulls in the narrow phase (every Panda / IIWA / Robotiq link, every PickPlace object; the slowest and most-used path): tessellated primitives whose
# contacts with planes, boxes and each other are elementary geometry.  Convention stated here as a test: ONE contact per convex pair -- the deepest point along
# the normal for plane-hull, MPR's penetration along the origin ray otherwise (MuJoCo's libccd / MPR default without multiccd [3P]); a flat-on-flat contact has
# its depth and normal determined, its POINT only up to the contact patch.
def _mesh_assets(tmp):
    """STL files of a box (8 vertices), a 64-gon prism and a geodesic sphere (642 vertices), centred and axis-aligned; (half extents, radius / half height, radius)."""
    from scipy.spatial import ConvexHull
    from tests.test_mjcf import _write_stl

    def hull_faces(v):
        fs = ConvexHull(v).simplices.copy()
        for k, fc in enumerate(fs):
            if np.dot(np.cross(v[fc[1]] - v[fc[0]], v[fc[2]] - v[fc[0]]), v[fc[0]]) < 0:
                fs[k] = fc[::-1]
        return fs
    h = np.array([0.04, 0.025, 0.03])
    box = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=float) * h
    a = 2 * np.pi * np.arange(64) / 64
    ring = np.stack([0.03 * np.cos(a), 0.03 * np.sin(a)], 1)
    prism = np.vstack([np.c_[ring, np.full(64, -0.05)], np.c_[ring, np.full(64, 0.05)]])
    # geodesic sphere: icosahedron subdivided three times (642 vertices on the sphere of radius 0.045)
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = np.array(v, dtype=float); v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(3):
        fs = ConvexHull(v).simplices
        mid = {tuple(sorted(e)) for f in fs for e in ((f[0], f[1]), (f[1], f[2]), (f[0], f[2]))}
        m = np.array([(v[i] + v[j]) / 2 for i, j in mid]); m /= np.linalg.norm(m, axis=1, keepdims=True)
        v = np.vstack([v, m])
    ball = 0.045 * v
    for name, vv in (("mbox", box), ("mprism", prism), ("mball", ball)):
        vv32 = vv.astype(np.float32).astype(np.float64)
        _write_stl(os.path.join(tmp, name + ".stl"), vv32, hull_faces(vv32))
    return h, (0.03, 0.05), 0.045, len(ball)


def mesh_scene(tmp, bodies):
    b = "".join('<body name="b%d" pos="%s" %s><freejoint/><geom name="g%d" %s/></body>' % (i, p, ('quat="%s"' % q) if q else "", i, g) for i, (p, q, g) in enumerate(bodies))
    return (f'<mujoco><compiler angle="radian" meshdir="{tmp}"/><option timestep="0.002"/><asset><mesh name="mbox" file="mbox.stl"/><mesh name="mprism" file="mprism.stl"/>'
            f'<mesh name="mball" file="mball.stl"/></asset><worldbody><geom name="floor" type="plane" size="2 2 0.1"/>{b}</worldbody></mujoco>')


def mesh_narrow_phase_cases(tmp):
    """(name, scene, expected (dist, pos or None, normal), (tol dist, tol pos, tol normal) for an exact-arithmetic implementation of the contract)."""
    h, (pr, ph), br, nball = _mesh_assets(str(tmp))
    z = np.array([0.0, 0.0, 1.0])
    qx45 = f"{np.cos(np.pi / 8)} {np.sin(np.pi / 8)} 0 0"       # 45 degrees about x: an edge of the box points down / up
    qy45 = f"{np.cos(np.pi / 8)} 0 {np.sin(np.pi / 8)} 0"
    # box mesh tilted about x and y so that ONE vertex is lowest: R = Rx(0.3) Ry(-0.4); the lowest vertex is the one minimising (R v).z
    cx, sx, cy, sy = np.cos(0.3), np.sin(0.3), np.cos(-0.4), np.sin(-0.4)
    R = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    from robosuite_amd.mjcf import mat2quat
    qt = " ".join(f"{x:.12f}" for x in mat2quat(R))
    verts = np.array([[a, b, c] for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)], dtype=float) * h
    low = (R @ verts.T).T
    low = low[np.argmin(low[:, 2])]
    sag = br * (1 - np.cos(0.5 * np.arccos(1 - 2.0 / nball) ))   # a facet of a geodesic sphere lies at most ~r (1 - cos(half the vertex spacing)) inside the sphere
    return [
        ("mesh box flat on the plane, 1 mm deep: one contact at a lowest vertex", [("0.1 0.2 %.6f" % (h[2] - 0.001), None, 'type="mesh" mesh="mbox"')],
         [(-0.001, None, z)], (1e-7, None, 1e-9)),
        ("tilted mesh box, one vertex 2 mm into the plane", [("0 0 %.9f" % (-low[2] - 0.002), qt, 'type="mesh" mesh="mbox"')],
         [(-0.002, [low[0], low[1], -0.001], z)], (1e-7, 1e-7, 1e-9)),
        ("the same vertex 2 mm into the top face of a box", [("0 0 1", None, 'type="box" size="0.2 0.2 0.05"'), ("0 0 %.9f" % (1.05 - low[2] - 0.002), qt, 'type="mesh" mesh="mbox"')],
         [(-0.002, [low[0], low[1], 1.049], z)], (2e-6, 2e-4, 1e-4)),
        ("mesh box flat on a box, 1 mm deep: depth and normal of the primitive pair, the point anywhere in the patch", [("0 0 1", None, 'type="box" size="0.2 0.2 0.05"'), ("0.01 -0.02 %.6f" % (1.05 + h[2] - 0.001), None, 'type="mesh" mesh="mbox"')],
         [(-0.001, None, z)], (2e-6, None, 1e-4)),
        ("two mesh boxes, edge on edge (crossed at 90 degrees), 1 mm deep", [("0 0 1", qx45, 'type="mesh" mesh="mbox"'), ("0 0 %.9f" % (1 + (h[1] + h[2]) / 2 ** 0.5 + (h[0] + h[2]) / 2 ** 0.5 - 0.001), qy45, 'type="mesh" mesh="mbox"')],
         [(-0.001, [(h[0] - h[2]) / 2 ** 0.5, (h[1] - h[2]) / 2 ** 0.5, 1 + (h[1] + h[2]) / 2 ** 0.5 - 0.0005], z)], (2e-6, 2e-4, 1e-4)),   # where the upper box's lowest edge (along y, at x = (hx - hz) / sqrt 2) crosses the lower box's highest (along x)
        ("64-gon prism mesh standing on a box, 1 mm deep", [("0 0 1", None, 'type="box" size="0.2 0.2 0.05"'), ("0.03 0.01 %.6f" % (1.05 + ph - 0.001), None, 'type="mesh" mesh="mprism"')],
         [(-0.001, None, z)], (2e-6, None, 1e-4)),
        ("geodesic sphere mesh on a box: the sphere's answer to the accuracy of its tessellation", [("0 0 1", None, 'type="box" size="0.2 0.2 0.05"'), ("0.02 -0.01 %.6f" % (1.05 + br - 0.002), None, 'type="mesh" mesh="mball"')],
         [(-0.002, [0.02, -0.01, 1.049], z)], (sag + 2e-6, 0.006, 0.08)),
        ("sphere primitive on the top face of a mesh box (sphere is geom1)", [("0 0 1", None, 'type="mesh" mesh="mbox"'), ("0.02 -0.01 %.6f" % (1 + h[2] + 0.04 - 0.001), None, 'type="sphere" size="0.04"')],
         [(-0.001, [0.02, -0.01, 1 + h[2] - 0.0005], -z)], (2e-6, 2e-4, 1e-3)),
    ]


def check_mesh_contacts(contacts, expected, tol, name, slack=1.0):
    assert len(contacts) == len(expected) == 1, (name, len(contacts))          # one contact per convex pair
    c, (dist, pos, nrm), (td, tp, tn) = contacts[0], expected[0], tol
    assert abs(c["dist"] - dist) < td * slack, (name, c["dist"], dist)
    assert np.abs(np.asarray(c["frame"]).reshape(3, 3)[0] - np.asarray(nrm)).max() < max(tn * slack, 1e-7), (name, c["frame"], nrm)
    if pos is not None:
        assert np.abs(np.asarray(c["pos"]) - np.asarray(pos)).max() < tp * slack, (name, c["pos"], pos)


def test_mesh_narrow_phase_against_elementary_geometry(tmp_path):
    """Hull vertex on the plane / on a box face, hull flat on a box, hull edge on hull edge, a 64-gon prism and a geodesic sphere as meshes, a sphere primitive on a
    hull: depth, normal and (where the geometry determines it) contact point of the oracle's plane-hull and MPR paths."""
    for name, bodies, expected, tol in mesh_narrow_phase_cases(tmp_path):
        flat = mjcf.compile_mjcf(mesh_scene(str(tmp_path), bodies))
        om, od, _ = make_oracle(flat)
        od.qpos[:] = om.field("qpos0"); od.forward()
        cs = od.contacts()
        check_mesh_contacts(cs, expected, tol, name)
        check_mesh_patch(cs, bodies, name)


def check_mesh_patch(cs, bodies, name, tol=1e-5):
    """The point of a flat-on-flat contact: halfway between the two surfaces, somewhere inside the footprint of the upper body."""
    if "flat on" in name or "standing" in name:
        p = np.asarray(cs[0]["pos"])
        centre = np.array([float(x) for x in bodies[-1][0].split()])
        assert abs(p[2] - (-0.0005 if len(bodies) == 1 else 1.0495)) < tol and np.abs(p[:2] - centre[:2]).max() < 0.0401, (name, p)


def test_line_search_does_not_creep_on_a_stacked_cube_state():
    """A state of the Stack bench workload (env 871 of 4096, control step 300, substep 18: two cubes stacked under a grasp) on which the oracle's line search used
    to creep in from both ends of its bracket for all 50 evaluations, end above its start and make the Newton iteration give up at iteration 0 -- found in round 4
    because the KERNEL's acceleration had the lower objective (profiles/r04_x10_line_search.txt).  The oracle must converge here, and at least as low as that point."""
    import json

    from robosuite_amd import mjcf

    adir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "assets")
    flat = mjcf.load_model(os.path.join(adir, "stack_panda.rsim"))
    cfg = json.load(open(os.path.join(adir, "stack_panda.cfg.json")))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stack_line_search_case.npz"))
    om, od, _ = make_oracle(flat, cfg)
    od.qpos[:] = z["qpos"]; od.qvel[:] = z["qvel"]; od.qacc_warmstart[:] = z["qacc_warmstart"]; od.ctrl[:] = z["ctrl"]
    od.forward()
    own, at_kernel = od.cost(od.qacc.copy()), od.cost(z["qacc_kernel"])
    assert od.solver_iter >= 2, od.solver_iter
    assert own <= at_kernel + 1e-6 * abs(at_kernel), (own, at_kernel)          # 730.08 against 695.86 before the fix
    c, g = od.cost(od.qacc.copy(), with_gradient=True)
    assert np.abs(g).max() < 1e-6 * max(1.0, abs(c)), np.abs(g).max()           # a stationary point of the convex objective
    # ... and the one an independent algorithm finds: projected Gauss-Seidel on the dual of the same problem (solve_pgs).  The kernel's fp32 acceleration of
    # that substep, which exposed the defect by having the lower objective, sits at the same point to fp32 accuracy.
    dual = flat.copy()
    dual.arrays["solver"][:] = 0; dual.arrays["iterations"][:] = 50000; dual.arrays["tolerance"][:] = 0
    om2, od2, _ = make_oracle(dual)
    od2.qpos[:] = z["qpos"]; od2.qvel[:] = z["qvel"]; od2.qacc_warmstart[:] = z["qacc_warmstart"]; od2.ctrl[:] = z["ctrl"]
    od2.forward()
    scale = max(1.0, np.abs(od.qacc).max())
    assert np.abs(od.qacc - od2.qacc).max() < 1e-6 * scale, np.abs(od.qacc - od2.qacc).max()
    assert np.abs(z["qacc_kernel"] - od2.qacc).max() < 1e-4 * scale, np.abs(z["qacc_kernel"] - od2.qacc).max()


def test_conditioning_probe_rounding_the_solver_inputs_to_float32_moves_pickplace_accelerations_by_parts_per_million():
    """What is the floor under an fp32 kernel's deviation from this oracle on the widest BASELINE model?  rso_set_round_rows makes the fp64 constraint solve see its
    inputs -- Jacobian rows, reference accelerations, regularisers, friction coefficients, mass matrix, smooth forces -- rounded to float32.  On PickPlace states
    reached under full-range random actions the solution moves by 1e-6 .. 1e-5 of each dof group's largest acceleration (objects, gripper, arm), forces by 1e-7 of
    the largest: storing the problem in single precision is NOT what bounds the kernel's per-env agreement (round 5: the tail of tests/test_full_size_parity.py
    was traced to solves that the fp32 exit rules ended before any fp64 look, not to conditioning)."""
    import json, os
    from robosuite_amd import lift, mjcf, pick_place
    adir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "assets")
    flat = mjcf.load_model(os.path.join(adir, "pickplace_iiwa.rsim")); cfg = json.load(open(os.path.join(adir, "pickplace_iiwa.cfg.json")))
    arm, fing = np.asarray(cfg["dof_idx"]), np.asarray(cfg["grip_dof_idx"])
    groups = {"arm": arm, "gripper": fing, "objects": np.setdiff1d(np.arange(flat.nv), np.concatenate([arm, fing]))}
    worst = {k: 0.0 for k in groups} | {"force": 0.0}
    ncon = 0
    for i in (0, 700, 1500, 2900, 4100, 6000):
        om, od, oc = make_oracle(flat, cfg)
        od.qpos[:] = pick_place.episode_setup(cfg, flat.nq, 0, [i], block=0)[0]; od.qvel[:] = 0; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward(); oc.reset(od)
        acts = lift.env_actions(np.array([i]), 40, action_dim=7)[:, 0]
        for t in range(40):
            oc.env_step(od, acts[t], 25)
        q, v, ws, ctrl = (np.array(x) for x in (od.qpos, od.qvel, od.qacc_warmstart, od.ctrl))
        od.forward(); a0 = np.array(od.qacc); f0 = np.array(od.efc_force[:od.nefc]); n0 = od.nefc
        od.qpos[:] = q; od.qvel[:] = v; od.qacc_warmstart[:] = ws; od.ctrl[:] = ctrl
        od.set_round_rows(True); od.forward(); a1 = np.array(od.qacc); f1 = np.array(od.efc_force[:od.nefc])
        assert od.nefc == n0 > 0
        ncon += od.ncon
        for k, idx in groups.items():
            worst[k] = max(worst[k], float(np.abs(a1[idx] - a0[idx]).max() / max(1.0, np.abs(a0[idx]).max())))
        worst["force"] = max(worst["force"], float(np.abs(f1 - f0).max() / max(1.0, np.abs(f0).max())))
    print("rounding the solver's inputs to float32, worst relative change over 6 PickPlace states:", {k: f"{x:.1e}" for k, x in worst.items()}, "contacts", ncon)
    assert ncon >= 30 and 0 < worst["objects"] < 1e-4 and worst["gripper"] < 1e-3 and worst["arm"] < 1e-4 and worst["force"] < 1e-5
