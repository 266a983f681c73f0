"""Contact COUNT and PLACEMENT conventions, per pair type the four BASELINE models use -- the one parity lever on the physics half that needs no `mujoco` wheel
(round-5 review, item 6).  The narrow phase is this project's own design in BOTH the fp64 oracle and the kernel (oracle/rsim_oracle.c header), so a difference of
convention between it and MuJoCo is invisible to every HIP-vs-oracle comparison.  This file writes the conventions down next to what MuJoCo documents
(mujoco.readthedocs.io: "Computation > Collision detection"; XML reference `option/flag/multiccd`, `geom/condim|margin|gap`; [3P]: third-party knowledge, MuJoCo's
source is not under /root/reference) and asserts them on canonical poses whose contacts are elementary geometry: the oracle here (CPU), the kernel through the C-ABI
in tests/test_hip_known_answers.py::test_contact_conventions_on_the_kernel (same cases).

Where the reference relies on them: `sim.data.contact[:ncon]` is read by `utils/sim_utils.py` (check_contact / get_contacts) and by every env's grasp test
(environments/manipulation/manipulation_env.py `_check_grasp`), and contact count / placement decides the constraint rows of utils/binding_utils.py:1089-1107's
`mj_step`.  Rows marked DIFFERS (same = False) are known deviations of convention, stated in DESIGN.md section 3.1.  Among the candidate pairs of the four BASELINE models they
occur in one place only -- the arena floor against the cylinders of Baxter's arm links and the peg, far below the working volume of TwoArmPegInHole -- which the
first test asserts; Lift, Stack and PickPlace pair their planes with boxes and mesh hulls only."""
import numpy as np
import pytest

from robosuite_amd import mjcf
from tests.test_oracle import narrow_phase_scene
from tests.util import make_oracle

# pair type -> (what MuJoCo documents [3P], what oracle and kernel do, same convention?)
CONVENTIONS = {
    "plane-box": ("up to 4 contacts at the penetrating corners (mjc_PlaneBox), position half-way between corner and plane, normal = plane normal",
                  "the same: lanes 0-7 test the eight corners, the first four penetrating ones in corner order", True),
    "plane-sphere / ellipsoid / mesh hull": ("one contact at the deepest point (mjc_PlaneSphere / mjc_PlaneConvex), midpoint, plane normal", "the same: support point along -normal", True),
    "plane-capsule": ("up to 2 contacts, one per end sphere (mjc_PlaneCapsule)", "ONE contact at the deepest support point (for a lying capsule: one end of the contact line)", False),
    "plane-cylinder": ("up to 4 contacts on the rim / along the line (mjc_PlaneCylinder)", "ONE contact at the support point (upright: the cap's centre; lying: one end of the contact line)", False),
    "box-box": ("up to 8 contacts: the incident face clipped against the reference face (mjc_BoxBox), reference-face normal", "the same construction: SAT axis, lane-parallel Sutherland-Hodgman, up to 8 polygon vertices", True),
    "convex-convex (cylinder, capsule, ellipsoid, mesh hull, sphere-box ...)": ("ONE contact from the general convex routine -- libccd MPR before 3.2, native GJK / EPA since; multiccd is opt-in and "
                                                                               "not set by robosuite's XML (models/assets/base.xml) -- penetration depth, position between the witness points",
                                                                               "ONE contact from MPR (cold start every substep): depth along the origin ray, position = midpoint of the witness points", True),
    "margin / gap": ("contact detected when dist < margin, constraint rows when dist < margin - gap", "the same (rows only when dist < margin - 1e-7 on the fp32 kernel: guard against exactly-touching pairs)", True),
}

QY = f"{np.cos(np.pi / 4)} 0 {np.sin(np.pi / 4)} 0"          # 90 degrees about y: the local z axis lies along world x
QZ45 = f"{np.cos(np.pi / 8)} 0 0 {np.sin(np.pi / 8)}"
Z = np.array([0.0, 0.0, 1.0])
K = 0.05 * (np.sqrt(2.0) - 1.0)                              # where the edges of a square turned by 45 degrees cut the edges of an equal square


def convention_cases():
    """(name, bodies, expected) with expected = list of (dist, position or None, normal, patch) -- `position` where the geometry determines the point, otherwise `patch`,
    a predicate the point must satisfy (MPR leaves the point of a flat-on-flat contact free inside the contact patch)."""
    oct8 = [(sx * 0.05, sy * K) for sx in (-1, 1) for sy in (-1, 1)] + [(sx * K, sy * 0.05) for sx in (-1, 1) for sy in (-1, 1)]
    return [
        ("plane-box: four lower corners", [("0 0 0.0495", None, 'type="box" size="0.05 0.03 0.05"')],
         [(-0.0005, [sx * 0.05, sy * 0.03, -0.00025], Z, None) for sx in (-1, 1) for sy in (-1, 1)]),
        ("plane-ellipsoid: one contact under the centre", [("0 0 0.0295", None, 'type="ellipsoid" size="0.05 0.04 0.03"')], [(-0.0005, [0, 0, -0.00025], Z, None)]),
        ("plane-capsule upright: one contact", [("0 0 0.1195", None, 'type="capsule" size="0.02 0.1"')], [(-0.0005, [0, 0, -0.00025], Z, None)]),
        ("plane-capsule lying: ONE contact at an end of the line (MuJoCo: two)", [("0 0 0.0195", QY, 'type="capsule" size="0.02 0.1"')],
         [(-0.0005, None, Z, lambda p: abs(abs(p[0]) - 0.1) < 1e-6 and abs(p[1]) < 1e-6 and abs(p[2] + 0.00025) < 1e-6)]),
        ("plane-cylinder upright: ONE contact at the cap's centre (MuJoCo: rim contacts)", [("0 0 0.0495", None, 'type="cylinder" size="0.03 0.05"')], [(-0.0005, [0, 0, -0.00025], Z, None)]),
        ("plane-cylinder lying: ONE contact at an end of the line (MuJoCo: several)", [("0 0 0.0295", QY, 'type="cylinder" size="0.03 0.05"')],
         [(-0.0005, None, Z, lambda p: abs(abs(p[0]) - 0.05) < 1e-6 and abs(p[1]) < 1e-6 and abs(p[2] + 0.00025) < 1e-6)]),
        ("box-box aligned, equal: four corners of the common face", [("0 0 1", None, 'type="box" size="0.05 0.05 0.05"'), ("0 0 1.099", None, 'type="box" size="0.05 0.05 0.05"')],
         [(-0.001, [sx * 0.05, sy * 0.05, 1.0495], Z, None) for sx in (-1, 1) for sy in (-1, 1)]),
        ("box-box turned by 45 degrees, equal: the eight vertices of the octagon", [("0 0 1", None, 'type="box" size="0.05 0.05 0.05"'), ("0 0 1.099", QZ45, 'type="box" size="0.05 0.05 0.05"')],
         [(-0.001, [x, y, 1.0495], Z, None) for x, y in oct8]),
        ("cylinder upright on a box (MPR): one contact inside the disc", [("0 0 1", None, 'type="box" size="0.1 0.1 0.05"'), ("0.01 0.02 1.099", None, 'type="cylinder" size="0.03 0.05"')],
         [(-0.001, None, -Z, lambda p: np.hypot(p[0] - 0.01, p[1] - 0.02) <= 0.03 + 1e-4 and abs(p[2] - 1.0495) < 2e-5)]),
        ("cylinder lying on a box (MPR): one contact on the contact line", [("0 0 1", None, 'type="box" size="0.1 0.1 0.05"'), ("0.01 0.02 1.079", QY, 'type="cylinder" size="0.03 0.05"')],
         [(-0.001, None, -Z, lambda p: abs(p[0] - 0.01) <= 0.05 + 1e-4 and abs(p[1] - 0.02) < 2e-3 and abs(p[2] - 1.0495) < 2e-5)]),
    ]


def check_conventions(contacts, expected, tol_d, tol_p, tol_n, name):
    assert len(contacts) == len(expected), (name, "contact COUNT", len(contacts), len(expected))
    left = list(expected)
    for c in contacts:
        p = np.asarray(c["pos"], dtype=np.float64)
        k = int(np.argmin([np.linalg.norm(p - np.asarray(e[1])) if e[1] is not None else 0.0 for e in left]))
        dist, pos, nrm, patch = left.pop(k)
        assert abs(c["dist"] - dist) < tol_d, (name, c["dist"], dist)
        if pos is not None:
            assert np.abs(p - np.asarray(pos)).max() < tol_p, (name, "contact PLACEMENT", p, pos)
        else:
            assert patch(p), (name, "contact PLACEMENT outside the patch", p)
        assert np.abs(np.asarray(c["frame"]).reshape(3, 3)[0] - np.asarray(nrm)).max() < tol_n, (name, c["frame"], nrm)


def test_conventions_table_names_every_pair_type_of_the_baseline_models():
    """The geom types of the four shipped BASELINE models pair up only in ways the table above covers.  The DIFFERS rows (plane against cylinder / capsule /
    ellipsoid) occur as CANDIDATE pairs in exactly one place: the arena floor against the cylinders of Baxter's arm links and the peg (TwoArmPegInHole) -- the floor
    the robot's pedestal stands on, far below the arms' working volume; Lift, Stack and PickPlace have none (their plane meets boxes and mesh hulls only)."""
    import os
    adir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosuite_amd", "assets")
    PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH = 0, 2, 3, 4, 5, 6, 7
    differs = {}
    for stem in ("lift_panda", "stack_panda", "peg_baxter_joint_velocity", "pickplace_iiwa"):
        flat = mjcf.load_model(os.path.join(adir, stem + ".rsim"))
        gt = np.asarray(flat.geom_type).ravel()
        g1, g2 = np.asarray(flat.arrays["pair_geom1"]).ravel(), np.asarray(flat.arrays["pair_geom2"]).ravel()       # candidate pairs, model geom ids
        assert len(g1) > 0
        names = flat.names["geom"]
        for a, b in zip(g1, g2):
            t = tuple(sorted((int(gt[a]), int(gt[b]))))
            assert all(x in (PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH) for x in t), (stem, t)
            if t in ((PLANE, CAPSULE), (PLANE, ELLIPSOID), (PLANE, CYLINDER)):
                differs.setdefault(stem, []).append((names[a], names[b]))
    assert set(differs) == {"peg_baxter_joint_velocity"}, differs
    for a, b in differs["peg_baxter_joint_velocity"]:
        assert a == "floor" and (b.startswith("robot0_") or b.startswith("peg")), (a, b)
    assert sum(not same for _, _, same in CONVENTIONS.values()) == 2


def test_oracle_contact_count_and_placement_per_pair_type():
    for name, bodies, expected in convention_cases():
        flat = mjcf.compile_mjcf(narrow_phase_scene(bodies))
        om, od, _ = make_oracle(flat)
        od.qpos[:] = om.field("qpos0"); od.forward()
        mpr = "MPR" in name
        check_conventions(od.contacts(), expected, 2e-6 if mpr else 1e-9, 1e-8, 1e-5 if mpr else 1e-8, name)
