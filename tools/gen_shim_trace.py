"""Record the exact `mujoco`-surface call trace robosuite makes for make -> reset -> N x env.step on Lift/Panda (build container only).

The UNMODIFIED reference runs over robosuite_amd.shim with the CPU oracle as arithmetic backend, wrapped in a tracer: every backend call
robosuite issues through utils/binding_utils.py:1059-1192 (mj_forward / mj_step1 / mj_step2 / mj_resetData), controllers/parts/controller.py:
226-227 (mj_fullM) and binding_utils.py:681-851 (mj_jacSite / mj_jacBody) is logged with
  * the state robosuite had written through its numpy views right before the call (qpos, qvel, ctrl, qacc_warmstart, time), and
  * everything it can read back afterwards (state, body / site / geom frames, qfrc_bias, qacc, ncon, the Jacobian or mass matrix returned).
tests/test_hip_shim_trace.py replays the trace on the GPU box (no reference checkout there) through HipShimBackend, call by call from the
recorded inputs, and compares every returned array.

Writes tests/golden/shim_trace_lift.npz.   Usage: python tools/gen_shim_trace.py [n_steps] [--playback] [--gripper panda|robotiq140|rethink]
--playback: the trace of the reference's action-playback determinism test instead (get_xml -> reset_from_xml_string -> set_state_from_flattened ->
replay; tests/golden/shim_trace_lift_playback.npz).
--gripper NAME: the trace of the reference's gripper behaviour test (tests/test_grippers/test_panda_gripper.py:8-24, test_robotiq_140.py, test_rethink_gripper.py ->
models/grippers/gripper_tester.py:204-226): a gripper on a vertical slide above a cube on a table, lower / grip / raise, 4 x 400 calls of sim.step() with
ctrl and the gravity-compensating qfrc_applied written before each; the test's own verdict (cube lifted above y_baseline) is recorded with it
(tests/golden/shim_trace_gripper_<name>.npz; every 8th step keeps its full record, all keep ctrl / qfrc_applied / cube height).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from robosuite_amd import mjcf, shim  # noqa: E402
from oracle.shim_backend import OracleBackend  # noqa: E402

PRE = ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")
POST = ("qpos", "qvel", "qacc_warmstart", "time", "xpos", "xquat", "xmat", "site_xpos", "site_xmat", "geom_xpos", "qfrc_bias", "qacc")
OPS = ("reset", "forward", "step1", "step2", "step", "jac_site", "jac_body", "full_M")

EVENTS = []     # (op code, model index, int argument)
ROWS = {}       # per op: list of flat float32 rows  [pre..., post..., extra...]
MODELS = []     # blobs, one per backend instance


class TracingBackend:
    def __init__(self, flat):
        self.inner = OracleBackend(flat)
        self.mi = len(MODELS)
        MODELS.append(np.frombuffer(mjcf.to_blob(flat), dtype=np.uint8).copy())

    def __getattr__(self, k):          # model_array / data_array / sync_model / ncon / contacts pass through
        return getattr(self.inner, k)

    def _log(self, op, arg, pre, extra=()):
        post = [np.asarray(self.inner.data_array(k), dtype=np.float64).ravel() for k in POST]
        row = np.concatenate(pre + post + [np.array([float(self.inner.ncon)])] + [np.asarray(e, dtype=np.float64).ravel() for e in extra])
        ROWS.setdefault((op, self.mi), []).append(row.astype(np.float32) if op.startswith("jac") or op == "full_M" else row)
        EVENTS.append((OPS.index(op), self.mi, int(arg)))

    def _pre(self):
        return [np.asarray(self.inner.data_array(k), dtype=np.float64).ravel().copy() for k in PRE]

    def _run(self, op):
        pre = self._pre()
        getattr(self.inner, op)()
        self._log(op, 0, pre)

    def forward(self): self._run("forward")
    def step1(self): self._run("step1")
    def step2(self): self._run("step2")
    def step(self): self._run("step")
    def reset(self): self._run("reset")

    def jac(self, kind, idx):
        pre = self._pre()
        jp, jr = self.inner.jac(kind, idx)
        self._log("jac_" + kind, idx, pre, (jp, jr))
        return jp, jr

    def full_M(self):
        pre = self._pre()
        M = self.inner.full_M()
        self._log("full_M", 0, pre, (M,))
        return M


def gripper_trace(name):
    """GripperTester of the reference over the shim (oracle arithmetic), traced call by call."""
    global PRE
    PRE = PRE + ("qfrc_applied",)
    shim.install(TracingBackend)
    from robosuite.models.grippers import GripperTester, PandaGripper, RethinkGripper, Robotiq140Gripper

    cases = {"panda": (PandaGripper, dict(gripper_low_pos=-0.10, gripper_high_pos=0.01)),                                   # test_panda_gripper.py:13-20
             "robotiq140": (Robotiq140Gripper, dict(gripper_low_pos=0.02, gripper_high_pos=0.1, box_size=[0.025] * 3)),     # test_robotiq_140.py:10-18
             "rethink": (RethinkGripper, dict(gripper_low_pos=-0.07, gripper_high_pos=0.02))}                               # test_rethink_gripper.py:14-21
    cls, kw = cases[name]
    tester = GripperTester(gripper=cls(), pos="0 0 0.3", quat="0 0 1 0", render=False, **kw)
    tester.start_simulation()
    del EVENTS[:]; ROWS.clear()
    heights, ctrls, applied = [], [], []
    inner_step = tester.sim.step

    def step_and_note():
        ctrls.append(np.array(tester.sim.data.ctrl)); applied.append(np.array(tester.sim.data.qfrc_applied))
        inner_step()
        heights.append(tester.object_height)

    tester.sim.step = step_and_note
    tester.loop(total_iters=1, test_y=True)          # raises ValueError if the cube is not lifted: the reference test's verdict
    mi = len(MODELS) - 1
    rows = ROWS[("step", mi)]
    keep = [k for k in range(len(rows)) if k % 8 == 0 or k == len(rows) - 1]
    f = tester.sim.model._model._flat
    out = dict(ops=np.array(OPS), pre=np.array(PRE), post=np.array(POST), model=MODELS[mi], kept=np.array(keep, dtype=np.int32), rows_step=np.stack([rows[k] for k in keep]),
               ctrl=np.array(ctrls), qfrc_applied=np.array(applied), height=np.array(heights), y_baseline=np.array(0.01),
               object_body=np.array(tester.object_id), z_dof=np.array(tester._gravity_corrected_qvels[0]), object_default_z=np.array(tester.object_default_pos[2]),
               state0=np.array(tester.sim_state.flatten()), nq=np.array(int(f.nq)))
    path = os.path.join(ROOT, "tests", "golden", f"shim_trace_gripper_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "steps", len(rows), "kept", len(keep), "final cube height", heights[-1], "bytes", os.path.getsize(path))


if __name__ == "__main__":
    if "--gripper" in sys.argv:
        gripper_trace(sys.argv[sys.argv.index("--gripper") + 1])
        sys.exit(0)
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    shim.install(TracingBackend)
    import robosuite as suite

    env = suite.make("Lift", robots="Panda", has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, use_object_obs=True,
                     reward_shaping=True, control_freq=20, horizon=500, ignore_done=True, seed=0)
    env.reset()
    rng = np.random.default_rng(10**6)
    playback = "--playback" in sys.argv
    extra_out = {}
    if not playback:
        for t in range(n_steps):
            env.step(rng.uniform(-1, 1, env.action_dim))
    else:
        # the reference's own determinism test (tests/test_environments/test_action_playback.py:17-71): record states, then rebuild the env from
        # sim.model.get_xml() (binding_utils.py:504-509 -> mj_saveLastXML), restore the first flattened state (binding_utils.py:1172-1184) and replay the
        # actions; the replayed states must equal the recorded ones BITWISE.  Only the playback half is traced (new model instance included).
        xml0 = env.sim.model.get_xml()
        state0 = np.array(env.sim.get_state().flatten())

        def restore():
            env.reset_from_xml_string(xml0)
            env.sim.reset()
            env.sim.set_state_from_flattened(state0)
            env.sim.forward()

        restore()                                        # "trick for ensuring that we can play MuJoCo demonstrations back deterministically" (:40-45)
        acts = [0.1 * rng.uniform(-1, 1, env.action_dim) for _ in range(n_steps)]
        states = []
        for a in acts:
            env.step(a); states.append(np.array(env.sim.get_state().flatten()))
        env.reset()
        del EVENTS[:]; ROWS.clear()                      # keep the model blobs (indices stay valid), trace from here on
        restore()
        replay = []
        for a in acts:
            env.step(a); replay.append(np.array(env.sim.get_state().flatten()))
        same = all(np.array_equal(a, b) for a, b in zip(states, replay))
        assert same, "playback over the shim (oracle backend) is not bitwise"
        extra_out = dict(playback_states=np.array(replay), playback_bitwise=np.array(int(same)), state0=state0)
    out = dict(events=np.array(EVENTS, dtype=np.int32), ops=np.array(OPS), pre=np.array(PRE), post=np.array(POST), n_models=len(MODELS), **extra_out)
    for i, b in enumerate(MODELS):
        out[f"model{i}"] = b
    for (op, mi), rows in ROWS.items():
        out[f"rows_{op}_{mi}"] = np.stack(rows)
    path = os.path.join(ROOT, "tests", "golden", "shim_trace_lift_playback.npz" if playback else "shim_trace_lift.npz")
    np.savez_compressed(path, **out)
    names = {i: MODELS[i].size for i in range(len(MODELS))}
    print("events", len(EVENTS), {OPS[k]: int((out["events"][:, 0] == k).sum()) for k in range(len(OPS))}, "models", names, "bytes", os.path.getsize(path))
