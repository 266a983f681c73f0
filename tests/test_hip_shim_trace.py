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
