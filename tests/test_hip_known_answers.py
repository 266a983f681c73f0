"""The fused kernel against closed forms of MuJoCo's DOCUMENTED model -- not against the oracle: the same one-body systems tests/test_oracle.py holds the
fp64 restatement to (soft-constraint resting depths of a contact and a joint limit, implicit joint damping, soft friction loss, torsional slip on the
elliptic cone), run through the C-ABI's B = 1 entries (rsim_step / rsim_forward) on the MI355X.  What agrees here agrees with the documentation itself,
whatever the oracle does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from robosuite_amd import mjcf  # noqa: E402
from tests.test_oracle import _documented_impedance, _documented_spring  # noqa: E402
from tests.util import make_hip  # noqa: E402

H, G0 = 0.002, 9.81


def _batch(xml):
    flat = mjcf.compile_mjcf(xml)
    hm, hb = make_hip(flat, None, B=1)
    hb.set("qpos", np.asarray(flat.qpos0, dtype=np.float64)[None]); hb.set("qvel", 0.0); hb.set("ctrl", 0.0); hb.set("qacc_warmstart", 0.0)
    return flat, hb


def _depth(a0, solref, solimp):
    k = _documented_spring(solref, solimp[1], H)
    r = a0 / k
    for _ in range(200):
        d = _documented_impedance(solimp, r)
        r = a0 * (1 - d) / (k * d * d)
    return r


@pytest.mark.parametrize("solref,solimp", (((0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2)), ((0.01, 0.7), (0.8, 0.8, 0.002, 0.5, 2)), ((0.003, 1.0), (0.95, 0.99, 0.0005, 0.3, 3))))
def test_resting_depths_of_a_contact_and_a_joint_limit(solref, solimp):
    sr, si = " ".join(map(str, solref)), " ".join(map(str, solimp))
    flat, hb = _batch(f"""<mujoco><option timestep="{H}" cone="elliptic"/><worldbody><geom name="floor" type="plane" size="1 1 0.1" solref="{sr}" solimp="{si}"/>
        <body name="ball" pos="0 0 0.05"><freejoint/><geom name="ball" type="sphere" size="0.05" density="1000" solref="{sr}" solimp="{si}"/></body></worldbody></mujoco>""")
    for _ in range(3000):
        hb.step()
    hb.forward()
    q, v = hb.get("qpos")[0], hb.get("qvel")[0]
    want = _depth(G0, solref, solimp)
    assert np.abs(v).max() < 1e-4 and hb.get("ncon")[0] == 1
    assert 0.05 - q[2] == pytest.approx(want, rel=2e-3, abs=2e-7)               # fp32 position of a 5 cm ball: 4e-9 m of rounding on depths of 1e-5 .. 4e-4 m
    assert hb.contacts(0)[0]["normal_force"] == pytest.approx(1000 * 4 / 3 * np.pi * 0.05**3 * G0, rel=1e-3)   # measured 2e-4 (fp32)
    # a pendulum pushed against its joint limit by a motor
    flat, hb = _batch(f"""<mujoco><compiler angle="radian"/><option timestep="{H}" gravity="0 0 0"/><worldbody><body name="p"><joint name="h" type="hinge" axis="0 1 0" range="-0.3 0.3" solreflimit="{sr}" solimplimit="{si}"/>
        <geom type="sphere" size="0.02" pos="0.2 0 0" mass="0.5" contype="0" conaffinity="0"/></body></worldbody><actuator><motor joint="h" gear="1"/></actuator></mujoco>""")
    hb.set("qpos", np.array([[0.25]])); hb.set("ctrl", np.array([[0.4]]))
    for _ in range(6000):
        hb.step()
    hb.forward()
    inertia = float(hb.get("qM")[0].ravel()[0])
    assert abs(hb.get("qvel")[0][0]) < 1e-4 and hb.get("nefc")[0] == 1
    assert hb.get("qpos")[0][0] - 0.3 == pytest.approx(_depth(0.4 / inertia, solref, solimp), rel=5e-3, abs=3e-7)


def test_damping_friction_loss_and_torsional_slip():
    pend = """<mujoco><compiler angle="radian"/><option timestep="%g" gravity="0 0 0"/><worldbody><body><joint name="h" type="hinge" axis="0 0 1" %s/>
              <geom type="box" size="0.1 0.02 0.02" mass="1.5" contype="0" conaffinity="0"/></body></worldbody><actuator><motor joint="h" gear="1"/></actuator></mujoco>"""
    # implicit joint damping: one step takes v to v I / (I + h b)
    b = 0.7
    flat, hb = _batch(pend % (H, 'damping="%g"' % b))
    hb.forward(); inertia = float(hb.get("qM")[0].ravel()[0])
    hb.set("qvel", np.array([[2.0]]))
    v = 2.0
    for _ in range(50):
        hb.step(); v *= inertia / (inertia + H * b)
        assert hb.get("qvel")[0][0] == pytest.approx(v, rel=2e-5)            # fp32 over 50 steps: measured 2e-6
    # soft friction loss: creep at tau R / b below the bound, (tau - F) / I above it
    flat, hb = _batch(pend % (H, 'frictionloss="0.3"'))
    hb.set("ctrl", np.array([[0.25]]))
    for _ in range(400):
        hb.step()
    creep = 0.25 * (0.1 / 0.9) * (1.0 / inertia) / (2.0 / (0.95 * 0.02))
    assert hb.get("qvel")[0][0] == pytest.approx(creep, rel=1e-4)
    hb.set("ctrl", np.array([[0.8]])); hb.forward()
    assert hb.get("qacc")[0][0] == pytest.approx((0.8 - 0.3) / inertia, rel=2e-3)
    # torsional slip: the force on the elliptic cone, Newton's and Euler's equations for it
    mu_t, r = 0.02, 0.05
    flat, hb = _batch(f"""<mujoco><option timestep="{H}" cone="elliptic" impratio="1"/><worldbody><geom type="plane" size="1 1 0.1" condim="4" friction="1 {mu_t} 0.0001"/>
        <body name="ball" pos="0 0 {r}"><freejoint/><geom type="sphere" size="{r}" density="1000" condim="4" friction="1 {mu_t} 0.0001"/></body></worldbody></mujoco>""")
    for _ in range(1500):
        hb.step()
    m_ball = 1000 * 4 / 3 * np.pi * r**3
    v = np.zeros((1, 6)); v[0, 5] = 30.0
    hb.set("qvel", v); hb.forward()
    c = hb.contacts(0)[0]
    f = hb.get("efc_force")[0][c["efc_address"]:c["efc_address"] + c["dim"]]
    a = hb.get("qacc")[0]
    assert c["dim"] == 4 and abs(f[1]) < 1e-4 and abs(f[2]) < 1e-4
    assert abs(f[3]) == pytest.approx(mu_t * f[0], rel=1e-5)
    assert a[5] == pytest.approx(f[3] / (0.4 * m_ball * r * r), rel=1e-4)
    assert m_ball * a[2] == pytest.approx(f[0] - m_ball * G0, rel=1e-3)


def test_free_flight_and_coulomb_threshold():
    """Mechanics, no simulator: (a) a box in free flight follows the semi-implicit Euler recursion of constant gravity exactly (v_n = v_0 - g h n,
    z_n = z_0 + v_0 h n - g h^2 n (n + 1) / 2) and keeps its angular velocity (isotropic inertia); (b) a flat box on a plane under gravity tilted by theta stays put
    while tan(theta) < mu (elliptic cone, impratio 20, four corner contacts); above it every slipping contact's force sits on the cone, |f_t| = mu f_n against
    the motion, and the box obeys Newton's equation for those forces."""
    flat, hb = _batch(f"""<mujoco><option timestep="{H}"/><worldbody><body pos="0 0 1"><freejoint/><geom type="box" size="0.03 0.03 0.03" density="400"/></body></worldbody></mujoco>""")
    v0 = np.array([[0.1, -0.2, 0.3, 1.0, 2.0, -1.5]])
    hb.set("qvel", v0)
    n = 200
    for _ in range(n):
        hb.step()
    q, v = hb.get("qpos")[0], hb.get("qvel")[0]
    assert v[:3] == pytest.approx([0.1, -0.2, 0.3 - G0 * H * n], rel=1e-5, abs=1e-6)
    assert q[:3] == pytest.approx([0.1 * H * n, -0.2 * H * n, 1.0 + 0.3 * H * n - G0 * H * H * n * (n + 1) / 2], abs=2e-6)
    assert v[3:] == pytest.approx(v0[0, 3:], rel=1e-5)
    mu = 0.3
    m_box = 400 * 0.2 * 0.2 * 0.02
    for fac in (0.5, 0.9, 1.5):
        th = np.arctan(fac * mu)
        gx, gz = G0 * np.sin(th), -G0 * np.cos(th)
        flat, hb = _batch(f"""<mujoco><option timestep="{H}" cone="elliptic" impratio="20" gravity="{gx} 0 {gz}"/><worldbody><geom type="plane" size="5 5 0.1" friction="{mu} 0.005 0.0001"/>
            <body pos="0 0 0.01"><freejoint/><geom type="box" size="0.1 0.1 0.01" density="400" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>""")
        if fac < 1.0:                                           # below the Coulomb threshold: stays put (soft contacts creep at ~3e-5 m/s)
            for _ in range(300):
                hb.step()
            assert abs(hb.get("qvel")[0][0]) < 1e-3 and hb.get("ncon")[0] == 4, fac
            continue
        # above it: settle under the normal component alone, then give it the downhill velocity and evaluate one forward(): every corner contact slips, its
        # force lies ON the cone (|f_t| = mu f_n) against the motion, and the box obeys Newton's equation along the plane for exactly those forces.
        # (A sustained slide is no clean known answer in this model: the primal cone couples friction into the normal direction and the box hops.)
        flat0, hb0 = _batch(f"""<mujoco><option timestep="{H}" cone="elliptic" impratio="20" gravity="0 0 {gz}"/><worldbody><geom type="plane" size="5 5 0.1" friction="{mu} 0.005 0.0001"/>
            <body pos="0 0 0.01"><freejoint/><geom type="box" size="0.1 0.1 0.01" density="400" friction="{mu} 0.005 0.0001"/></body></worldbody></mujoco>""")
        for _ in range(1500):
            hb0.step()
        hb.set("qpos", hb0.get("qpos").astype(np.float64))
        v = np.zeros((1, 6)); v[0, 0] = 0.5
        hb.set("qvel", v); hb.forward()
        ft_sum, fn_sum = 0.0, 0.0
        cons = hb.contacts(0)
        assert len(cons) == 4
        for c in cons:
            f = hb.get("efc_force")[0][c["efc_address"]:c["efc_address"] + c["dim"]]
            t_world = c["frame"][1] * f[1] + c["frame"][2] * f[2]
            assert np.linalg.norm(f[1:3]) == pytest.approx(mu * f[0], rel=1e-4)
            assert t_world[0] < 0 and abs(t_world[1]) < 1e-3 * abs(t_world[0])       # against the motion
            ft_sum += t_world[0]; fn_sum += f[0]
        a = hb.get("qacc")[0]
        assert m_box * a[0] == pytest.approx(m_box * gx + ft_sum, rel=1e-3)
        assert m_box * a[2] == pytest.approx(m_box * gz + fn_sum, rel=1e-3, abs=1e-3)


def test_actuator_models_and_clamps():
    """fwd_actuation as documented (XML reference: actuator/motor, actuator/position, ctrlrange, forcerange, gear): force = gain x clamp(ctrl) + bias . [1, length,
    velocity], clamped to forcerange, times the gear on the joint -- one hinge per actuator, forward() accelerations against I^-1 x the expected torque."""
    xml = f"""<mujoco><compiler angle="radian"/><option timestep="{H}" gravity="0 0 0"/><worldbody>
      <body pos="0 0 0"><joint name="a" type="hinge" axis="0 0 1"/><geom type="box" size="0.1 0.02 0.02" mass="1.5" contype="0" conaffinity="0"/></body>
      <body pos="0 1 0"><joint name="b" type="hinge" axis="0 0 1"/><geom type="box" size="0.1 0.02 0.02" mass="1.5" contype="0" conaffinity="0"/></body>
      <body pos="0 2 0"><joint name="c" type="hinge" axis="0 0 1"/><geom type="box" size="0.1 0.02 0.02" mass="1.5" contype="0" conaffinity="0"/></body>
      <body pos="0 3 0"><joint name="d" type="slide" axis="1 0 0"/><geom type="box" size="0.1 0.02 0.02" mass="1.5" contype="0" conaffinity="0"/></body></worldbody>
      <actuator><motor joint="a" gear="3" ctrllimited="true" ctrlrange="-1 1"/><position joint="b" kp="20"/>
                <position joint="c" kp="200" forcelimited="true" forcerange="-1 1"/><position joint="d" kp="50" ctrllimited="true" ctrlrange="0 0.04"/></actuator></mujoco>"""
    flat, hb = _batch(xml)
    hb.set("qpos", np.array([[0.2, 0.1, 0.1, 0.01]])); hb.set("ctrl", np.array([[5.0, 0.3, 0.3, 0.5]]))
    hb.forward()
    M = hb.get("qM")[0].reshape(4, 4)
    inertia, mass = float(M[0, 0]), float(M[3, 3])
    assert inertia == pytest.approx(1.5 / 3 * (0.1**2 + 0.02**2), rel=1e-5) and mass == pytest.approx(1.5, rel=1e-6)
    want = np.array([3.0 * 1.0 / inertia,                      # motor: ctrl clamped to 1, gear 3
                     20.0 * (0.3 - 0.1) / inertia,              # position servo: kp (ctrl - q)
                     1.0 / inertia,                             # 200 x 0.2 = 40 N m clamped to the force range
                     50.0 * (0.04 - 0.01) / mass])              # ctrl clamped to 0.04
    assert hb.get("qacc")[0] == pytest.approx(want, rel=2e-5)
    assert hb.get("qfrc_actuator")[0] == pytest.approx(want * np.array([inertia, inertia, inertia, mass]), rel=2e-5)


def test_planar_two_link_arm_has_the_textbook_mass_matrix_and_bias():
    """The kernel's CRBA (incidence-matrix MFMAs) and RNE against the closed-form dynamics of the planar elbow manipulator, eight random states in one batch."""
    from tests.test_oracle import planar_2r_closed_form, planar_2r_xml
    flat = mjcf.compile_mjcf(planar_2r_xml())
    hm, hb = make_hip(flat, None, B=8)
    rng = np.random.default_rng(0)
    q, qd = rng.uniform(-2, 2, (8, 2)), rng.uniform(-3, 3, (8, 2))
    hb.set("qpos", q); hb.set("qvel", qd); hb.set("ctrl", 0.0); hb.set("qacc_warmstart", 0.0)
    hb.forward()
    for e in range(8):
        M, bias = planar_2r_closed_form(q[e], qd[e])
        assert hb.get("qM")[e].reshape(2, 2) == pytest.approx(M, rel=2e-6, abs=1e-7)
        assert hb.get("qfrc_bias")[e] == pytest.approx(bias, rel=1e-5, abs=2e-6)
        assert hb.get("qacc")[e] == pytest.approx(np.linalg.solve(M, -bias), rel=2e-5, abs=1e-4)


def test_narrow_phase_against_elementary_geometry():
    """The kernel's plane-box, plane-convex, box-box and MPR paths on the geometry cases of tests/test_oracle.py: depths, midpoints and normals of the contacts
    in fp32 (support points relative to the first geom keep MPR's rounding at ~1e-8 m)."""
    from tests.test_oracle import check_narrow_phase, narrow_phase_cases, narrow_phase_scene
    for name, bodies, expected in narrow_phase_cases():
        flat, hb = _batch(narrow_phase_scene(bodies))
        hb.forward()
        mpr = "sphere" in name and "plane" not in name or "capsule" in name
        # MPR stops when the portal is within 1e-6 m of the surface of the Minkowski difference: on curved shapes that leaves the normal free by about
        # sqrt(2 tol / r) ~ 6e-3 rad (measured 4e-3 on the two spheres; the fp64 oracle happens to start on the centre line and is exact) and the
        # point by r times that; depths are good to the tolerance itself
        check_narrow_phase(hb.contacts(0), expected, 3e-6, 1e-3 if mpr else 5e-6, 2e-2 if mpr else 1e-5, name)


def test_contact_conventions_on_the_kernel():
    """Contact COUNT and PLACEMENT per pair type on the canonical poses of tests/test_contact_conventions.py (what MuJoCo documents per pair and where this project's
    narrow phase deviates is written there): plane-box corners, one contact for plane-convex, the clipped-face manifold of box-box up to its eight vertices, one
    contact inside the patch for the MPR pairs -- the kernel's answers in fp32."""
    from tests.test_contact_conventions import check_conventions, convention_cases
    from tests.test_oracle import narrow_phase_scene
    for name, bodies, expected in convention_cases():
        flat, hb = _batch(narrow_phase_scene(bodies))
        hb.forward()
        mpr = "MPR" in name
        check_conventions(hb.contacts(0), expected, 3e-6, 5e-6, 2e-3 if mpr else 1e-5, name)


def test_mesh_narrow_phase_against_elementary_geometry(tmp_path):
    """The kernel's plane-hull and MPR-on-hull paths (hull vertices in registers, support scans as wave arg-max, coordinates relative to the first geom) on the
    mesh cases of tests/test_oracle.py: a hull vertex in the plane and in a box face, a hull flat on a box, two hulls edge on edge, a 64-gon prism, a geodesic
    sphere of 642 vertices (scanned from global memory: more than 256), a sphere primitive on a hull.  One contact per convex pair; depth, normal and -- where
    the geometry determines it -- the point, in fp32."""
    from tests.test_oracle import check_mesh_contacts, check_mesh_patch, mesh_narrow_phase_cases, mesh_scene
    for name, bodies, expected, tol in mesh_narrow_phase_cases(tmp_path):
        flat, hb = _batch(mesh_scene(str(tmp_path), bodies))
        hb.forward()
        cs = hb.contacts(0)
        # fp32: depths to 3e-6 m (MPR's own tolerance is 1e-6), points of vertex / edge contacts to 1e-4 m, normals to 2e-3
        check_mesh_contacts(cs, expected, (max(tol[0], 3e-6), None if tol[1] is None else max(tol[1], 2e-4), max(tol[2], 2e-3)), name)
        check_mesh_patch(cs, bodies, name, tol=2e-5)
