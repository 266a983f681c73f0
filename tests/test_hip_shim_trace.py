"""The `mujoco`-shaped surface robosuite drives (utils/binding_utils.py:1059-1192 mj_forward / mj_step1 / mj_step2, :681-851 mj_jacSite,
controllers/parts/controller.py:226-227 mj_fullM) on the HIP backend, without a reference checkout on the GPU box.

tools/gen_shim_trace.py recorded, in the build container, every backend call the UNMODIFIED reference makes for
`suite.make("Lift") -> reset() -> 10 x step(action)` over the shim (2262 calls: 250 x step1 / step2, 1504 site Jacobians, 251 mass matrices, the
forwards of make / reset), with the state robosuite had written before each call and everything it can read back after it.  Here the trace is
replayed call by call through HipShimBackend -- inputs from the record, so nothing drifts -- and every returned array is compared.
"""
import os

import numpy as np
import pytest

from robosuite_amd import mjcf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("trace,n_step1", (("shim_trace_lift.npz", 250), ("shim_trace_lift_playback.npz", 100)))
def test_binding_utils_call_trace_of_the_reference_replays_on_the_hip_shim_backend(trace, n_step1):
    """shim_trace_lift: make -> reset -> 10 x step.  shim_trace_lift_playback: the second half of the reference's action-playback determinism test
    (tests/test_environments/test_action_playback.py:46-68) -- env.reset(), reset_from_xml_string(sim.model.get_xml()), sim.reset(),
    set_state_from_flattened(recorded state), forward(), replay of the recorded actions: the model compiled again from the XML the shim hands back,
    the restored flattened state, and (recorder side, tools/gen_shim_trace.py --playback) a replay that equals the first run bitwise."""
    from robosuite_amd.hip_shim_backend import HipShimBackend

    g = np.load(os.path.join(GOLD, trace))
    if "playback_bitwise" in g.files:
        assert int(g["playback_bitwise"]) == 1
    ops, PRE, POST = [str(x) for x in g["ops"]], [str(x) for x in g["pre"]], [str(x) for x in g["post"]]
    backends, cursor, worst, knife, qacc_at = {}, {}, {}, 0, None
    rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    for opc, mi, arg in g["events"]:
        op, mi = ops[opc], int(mi)
        if mi not in backends:
            backends[mi] = HipShimBackend(mjcf.from_blob(g[f"model{mi}"].tobytes()))
        hb = backends[mi]
        f = hb.flat
        k = cursor.get((op, mi), 0); cursor[(op, mi)] = k + 1
        row = np.asarray(g[f"rows_{op}_{mi}"][k], dtype=np.float64)
        sizes = dict(qpos=f.nq, qvel=f.nv, ctrl=f.nu, qacc_warmstart=f.nv, time=1, xpos=3 * f.nbody, xquat=4 * f.nbody, xmat=9 * f.nbody, site_xpos=3 * f.nsite,
                     site_xmat=9 * f.nsite, geom_xpos=3 * f.ngeom, qfrc_bias=f.nv, qacc=f.nv)
        o = 0
        for name in PRE:                      # what robosuite wrote through its views before the call
            hb.d[name][:] = row[o:o + sizes[name]]; o += sizes[name]
        post = {}
        for name in POST:
            post[name] = row[o:o + sizes[name]]; o += sizes[name]
        ncon = int(row[o]); o += 1
        extra = row[o:]
        if op.startswith("jac"):
            hb.forward()                      # the Jacobian is evaluated at the recorded state (the oracle's derived arrays were current)
            jp, jr = hb.jac(op[4:], int(arg))
            got = np.concatenate([jp.ravel(), jr.ravel()])
            worst["jac"] = max(worst.get("jac", 0.0), float(np.abs(got - extra).max()))
            continue
        if op == "full_M":
            hb.forward()
            worst["full_M"] = max(worst.get("full_M", 0.0), rel(hb.full_M().ravel(), extra))
            continue
        getattr(hb, op)()
        stepped = op in ("step2", "step")
        for name in ("xpos", "xquat", "xmat", "site_xpos", "site_xmat", "geom_xpos"):
            if op != "step2":                 # step2 does not recompute positions (they stay those of step1)
                worst[name] = max(worst.get(name, 0.0), float(np.abs(hb.d[name] - post[name]).max()))
        if op != "step2":
            worst["qfrc_bias"] = max(worst.get("qfrc_bias", 0.0), rel(hb.d["qfrc_bias"], post["qfrc_bias"]))
        if op in ("forward", "step2", "step"):
            if hb.ncon == ncon:               # (with one contact more or less the accelerations are those of another constraint set)
                e = rel(hb.d["qacc"], post["qacc"])
                if e > worst.get("qacc", 0.0):
                    worst["qacc"], qacc_at = e, (k, op, int(ncon))
            else:
                # construction-time states of the playback trace (arm at its initial pose, fingers at qpos0 = 0 with the two pad boxes overlapping by
                # exactly 1 mm, face to face and edge to edge): the box-box clip of two perfectly aligned faces keeps or drops a polygon vertex that
                # lies ON a clipping edge depending on the last bit, 4 contacts in fp64, 5 in fp32.  No state the simulation steps from.
                assert op == "forward" and abs(hb.ncon - ncon) == 1, (op, k, hb.ncon, ncon, [(c["geom1"], c["geom2"], float(c["dist"])) for c in hb.contacts()])
                knife += 1
        if stepped:
            worst["qpos"] = max(worst.get("qpos", 0.0), float(np.abs(hb.d["qpos"] - post["qpos"]).max()))
            worst["qvel"] = max(worst.get("qvel", 0.0), rel(hb.d["qvel"], post["qvel"]))
            assert abs(hb.d["time"][0] - post["time"][0]) < 1e-6
    print("worst deviations over the trace:", {k: f"{v:.2e}" for k, v in worst.items()}, "; worst qacc at (event, op, ncon):", qacc_at)
    assert sum(cursor.values()) == len(g["events"]) and cursor[("step1", max(backends))] == n_step1 and knife <= 4
    for name in ("xpos", "xquat", "xmat", "site_xpos", "site_xmat", "geom_xpos"):
        assert worst[name] < 1.5e-6, (name, worst[name])          # measured 2e-7 .. 4e-7 (fp32 kernel, fp64 record)
    assert worst["jac"] < 2e-6 and worst["full_M"] < 3e-6 and worst["qfrc_bias"] < 3e-6   # measured 4e-7, 7e-7, 6e-7
    assert worst["qacc"] < 5e-4 and worst["qpos"] < 1e-6 and worst["qvel"] < 5e-6         # measured 2e-5 (2e-4 on the playback trace: step2 with the cube's four contacts), 2e-7, 1.4e-6


@pytest.mark.parametrize("name", ("panda", "robotiq140", "rethink"))
def test_gripper_tester_of_the_reference_on_the_hip_shim_backend(name):
    """The reference's gripper behaviour tests (tests/test_grippers/test_panda_gripper.py:8-24, test_robotiq_140.py, test_rethink_gripper.py ->
    models/grippers/gripper_tester.py:204-226) on the HIP backend.  tools/gen_shim_trace.py --gripper ran the UNMODIFIED GripperTester over the shim (oracle
    arithmetic) and recorded its 4 x 400 sim.step() calls: a gripper on a position-actuated vertical slide above a cube on a table -- lower, grip, raise --
    with ctrl and the gravity-compensating qfrc_applied (= qfrc_bias of the slide) written before every step.
    (1) call by call: every 8th recorded step replayed from its recorded inputs through HipShimBackend.step(), state after it compared;
    (2) the behaviour test itself, closed loop: the same 1600 steps on the HIP backend from the recorded initial state, the gravity compensation taken from the
        backend's OWN qfrc_bias as GripperTester._apply_gravity_compensation does, and the reference test's verdict asserted: the cube ends above y_baseline."""
    from robosuite_amd.hip_shim_backend import HipShimBackend

    g = np.load(os.path.join(GOLD, f"shim_trace_gripper_{name}.npz"))
    hb = HipShimBackend(mjcf.from_blob(g["model"].tobytes()))
    f = hb.flat
    PRE, POST = [str(x) for x in g["pre"]], [str(x) for x in g["post"]]
    sizes = dict(qpos=f.nq, qvel=f.nv, ctrl=f.nu, qacc_warmstart=f.nv, time=1, qfrc_applied=f.nv, xpos=3 * f.nbody, xquat=4 * f.nbody, xmat=9 * f.nbody, site_xpos=3 * f.nsite,
                 site_xmat=9 * f.nsite, geom_xpos=3 * f.ngeom, qfrc_bias=f.nv, qacc=f.nv)
    worst = {"qpos": 0.0, "qvel": 0.0, "xpos": 0.0}
    per_step = []
    for row in np.asarray(g["rows_step"], dtype=np.float64):
        o = 0
        for k in PRE:
            hb.d[k][:] = row[o:o + sizes[k]]; o += sizes[k]
        post = {}
        for k in POST:
            post[k] = row[o:o + sizes[k]]; o += sizes[k]
        hb.step()
        worst["qpos"] = max(worst["qpos"], float(np.abs(hb.d["qpos"] - post["qpos"]).max()))
        worst["qvel"] = max(worst["qvel"], float(np.abs(hb.d["qvel"] - post["qvel"]).max() / max(1.0, np.abs(post["qvel"]).max())))
        worst["xpos"] = max(worst["xpos"], float(np.abs(hb.d["xpos"] - post["xpos"]).max()))
        per_step.append((float(np.abs(hb.d["qpos"] - post["qpos"]).max()), float(np.abs(hb.d["qvel"] - post["qvel"]).max() / max(1.0, np.abs(post["qvel"]).max()))))
    med = np.median(np.array(per_step), axis=0)
    print(f"{name}: one-step deviations over {len(g['rows_step'])} recorded steps: worst", {k: f"{v:.1e}" for k, v in worst.items()}, f"median qpos {med[0]:.1e} qvel {med[1]:.1e}")
    assert med[0] < 2e-5 and med[1] < 2e-3 and worst["xpos"] < 2e-5, (worst, med)
    if name == "robotiq140":
        # The Robotiq140's finger / knuckle collision meshes interpenetrate by ~1 cm in every pose (adjacent links of its four-bar linkages): MPR between two fine
        # polytopes in deep penetration ends on one of several neighbouring facets, fp32 and fp64 break the ties differently (tests/test_full_size_parity.py says the
        # same of the PickPlace gripper), and in the steps where that happens one finger joint (5e-5 kg m^2) gets another push: up to 1 rad/s in that step.
        assert worst["qpos"] < 5e-3 and worst["qvel"] < 0.5, worst
    else:
        assert worst["qpos"] < 2e-5 and worst["qvel"] < 2e-3, worst
    # ---- the behaviour test, closed loop
    nq, zd, ob = int(g["nq"]), int(g["z_dof"]), int(g["object_body"])
    s0 = np.asarray(g["state0"], dtype=np.float64)
    hb.reset()
    hb.d["time"][:] = s0[0]; hb.d["qpos"][:] = s0[1:1 + nq]; hb.d["qvel"][:] = s0[1 + nq:1 + nq + f.nv]        # MjSim.set_state (binding_utils.py:1150-1170)
    heights = []
    for t in range(len(g["ctrl"])):
        hb.d["ctrl"][:] = g["ctrl"][t]                              # the lower / grip / raise schedule (gripper_tester.py:171-180, 204-214)
        hb.d["qfrc_applied"][:] = 0.0
        hb.d["qfrc_applied"][zd] = hb.d["qfrc_bias"][zd]            # _apply_gravity_compensation (gripper_tester.py:197-202)
        hb.step()
        heights.append(float(hb.d["xpos"].reshape(-1, 3)[ob][2] - float(g["object_default_z"])))
    ref = np.asarray(g["height"])
    print(f"{name}: cube height after lower / grip / raise: HIP {heights[-1]:.4f} m, recorded (oracle) {ref[-1]:.4f} m; largest difference along the way {np.abs(np.array(heights) - ref).max():.2e}")
    assert heights[-1] > float(g["y_baseline"])                     # the reference test's own assertion (gripper_tester.py:216-220)
    assert abs(heights[-1] - ref[-1]) < 5e-3
