"""Fixed tendons, equality/tendon constraints and tendon limits in the MJCF compiler and the CPU oracle (the Robotiq coupling pattern of
BASELINE configs[4], reference models/assets/grippers/robotiq_gripper_140.xml:15-44; spring and friction-loss attributes as on the Robotiq85
and Jaco fingers).  Device-side counterparts: tests/test_hip_parity.py (PickPlace / UR5e / Jaco fixtures)."""
import os

import numpy as np
import pytest

from robosuite_amd import mjcf
from tests.util import GOLD, make_oracle

XML = open(os.path.join(GOLD, "coupled_fingers.xml")).read()


def test_compiler_tables():
    m = mjcf.compile_mjcf(XML)
    assert int(m.ntendon) == 2 and int(m.neq) == 1
    assert list(m.tendon_num) == [2, 1] and list(m.tendon_adr) == [0, 2]
    assert list(m.wrap_objid) == [0, 1, 2] and list(m.wrap_prm) == [1.0, 1.5, 2.0]
    assert list(m.tendon_limited) == [1, 1] and m.tendon_range.tolist() == [[-0.5, 0.5], [-0.3, 0.3]]
    assert m.tendon_length0 == pytest.approx([0.0, 0.0])
    # inverse weight = J M^-1 J^T at qpos0
    M, _ = mjcf.mass_matrix_np(m, m.qpos0)
    J = np.array([1.0, 1.5, 0.0])
    assert m.tendon_invweight0[0] == pytest.approx(J @ np.linalg.inv(M) @ J)
    assert m.names["tendon"] == ["cpl", "lim_only"] and m.names["equality"] == ["cpl_eq"]
    m2 = mjcf.from_blob(mjcf.to_blob(m))
    assert int(m2.ntendon) == 2 and m2.wrap_prm.tolist() == [1.0, 1.5, 2.0]


def test_equality_keeps_the_coupled_length_while_the_actuator_drives_the_finger():
    m = mjcf.compile_mjcf(XML)
    om, od, _ = make_oracle(m)
    od.forward()
    assert od.nefc == 1 and od.efc_types()[0] == 4          # the equality row is always present and comes first
    worst = 0.0
    for t in range(600):
        od.ctrl[:] = [0.8, 0.0]
        od.step()
        worst = max(worst, abs(od.qpos[0] + 1.5 * od.qpos[1]))
    assert od.qpos[0] > 0.5                               # the position actuator moved the first phalanx ...
    assert od.qpos[1] == pytest.approx(-od.qpos[0] / 1.5, abs=2e-3)   # ... and the second followed the coupling q1 + 1.5 q2 = 0
    assert worst < 5e-2                                   # soft constraint (solref 0.02 s): bounded transient while the actuator steps
    # equation of motion with the equality force included
    od.forward()
    res = od.full_M() @ od.qacc + od.qfrc_bias - od.qfrc_passive - od.qfrc_actuator - od.qfrc_constraint
    assert np.abs(res).max() < 1e-8


def test_tendon_limit_stops_the_uncoupled_finger():
    m = mjcf.compile_mjcf(XML)
    om, od, _ = make_oracle(m)
    for t in range(1500):
        od.ctrl[:] = [0.0, 0.5]                           # constant torque on the free finger
        od.step()
    assert od.qpos[2] < 0.5                               # the joint range alone (+-1 rad) would let it go on
    assert 2 * od.qpos[2] == pytest.approx(0.3, abs=0.03)   # it rests on the tendon's upper length limit (2 q <= 0.3), soft
    od.forward()
    types = list(od.efc_types())
    assert types == [4, 5]                                # equality, then the tendon limit row


def test_tendon_frictionloss_row_sticks_and_saturates():
    """Dry friction along a fixed tendon (jaco_three_finger_gripper.xml:17: frictionloss on the finger tendons): a friction-loss row on the
    tendon's coefficient row, placed after the dof friction rows.  Below the limit the tendon sticks; above it the row force saturates at
    -frictionloss * sign(velocity), i.e. the joint feels coef * frictionloss."""
    def model(fl):
        return mjcf.compile_mjcf(XML.replace('<fixed name="lim_only" range="-0.3 0.3" limited="true">',
                                             f'<fixed name="lim_only" range="-0.3 0.3" limited="true" frictionloss="{fl}">'))
    m = model(0.6)
    assert m.tendon_frictionloss.tolist() == [0.0, 0.6] and m.tendon_solref_fri.tolist()[1] == [0.02, 1.0]
    om, od, _ = make_oracle(m)
    for t in range(300):
        od.ctrl[:] = [0.0, 0.5]                            # 0.5 N m on the joint < coef * frictionloss = 1.2 N m: the row does not saturate
        od.step()
    od.forward()
    assert list(od.efc_types()) == [4, 6]                  # equality, then the tendon friction row
    # soft constraint: in the unsaturated regime the row is a damper, force = -D B v with D = 1 / R, R = (1 - d0) / d0 * invweight0,
    # B = 2 / (dmax * timeconst) [3P: mj_makeImpedance, K = 0 for friction rows]; torque balance on the joint gives the creep velocity
    R, Bv = (1 - 0.9) / 0.9 * m.tendon_invweight0[1], 2 / (0.95 * 0.02)
    v_pred = 0.5 / (2 * 2 * Bv / R + 0.05)                 # 0.5 = coef * (B / R) * (coef * qvel) + damping * qvel
    assert od.qvel[2] == pytest.approx(v_pred, rel=2e-2)
    assert od.qfrc_constraint[2] == pytest.approx(-(0.5 - 0.05 * od.qvel[2]), abs=2e-3)
    m = model(0.2)
    om, od, _ = make_oracle(m)
    od.ctrl[:] = [0.0, 0.5]
    for t in range(20):
        od.step()
    od.forward()
    assert od.qvel[2] > 1e-2                                # 0.5 > 2 * 0.2: it slides ...
    assert od.qfrc_constraint[2] == pytest.approx(-0.4, abs=1e-6)   # ... against the saturated row: coef * frictionloss
    assert od.efc_force[1] == pytest.approx(-0.2, abs=1e-6)
