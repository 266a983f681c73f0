"""Record MuJoCo-produced known answers for the physics half of the oracle -- to be run where the `mujoco` wheel exists.

Status: this container (and the GPU box) has no `mujoco` distribution (`pip download mujoco` finds none, there is no network), so the
physics half of oracle/rsim_oracle.c is still "parity unpinned" (DESIGN.md section 3).  This script is the other half of the remedy the
round-1 review asked for: on any machine with `pip install mujoco` (>= 3.3, reference setup.py:18) and the reference checkout on
PYTHONPATH it drives the UNMODIFIED reference (real MjSim, no shim) with the seeds and action tapes of tests/golden/ and records, from
MuJoCo itself, the quantities the oracle restates (reference call sites: utils/binding_utils.py:1095-1107 mj_step1/mj_step2/mj_forward,
controllers/parts/controller.py:226-227 mj_fullM):

    per control step   time, qpos, qvel (sim.get_state), ctrl, reward
    per sampled state  qacc_warmstart, qM (mj_fullM), qfrc_bias, qfrc_passive, qfrc_actuator, qacc, qfrc_constraint,
                       ncon + contact records (dist, pos, frame, geom1, geom2, dim), nefc + efc_force / efc_aref / efc_R
    once               the MJCF string MuJoCo compiled (sim.model.get_xml()) so that mjcf.py compiles the very same model

into tests/golden/mujoco_<task>_<robot>_seed<k>.npz (+ .xml).  tests/test_mujoco_pin.py then holds the fp64 oracle to those files
(skipped while they are absent).  Nothing here imports the oracle or the HIP library.

Contact-count convention: next to every snapshot's contact records the script stores, per geom pair in contact, the pair's two geom types and the
NUMBER of contacts MuJoCo generated for it (`snapK_pair_counts`, rows geom1 geom2 type1 type2 count).  This project's narrow phase returns one contact
per convex (MPR) pair and up to eight per box-box pair; MuJoCo's libccd path returns one per convex pair unless `multiccd` is enabled, its box-box
routine up to eight -- the first thing tests/test_mujoco_pin.py diffs once a fixture exists.

Usage:  python tools/gen_golden_with_mujoco.py [--out DIR] [--cases Lift:0,Stack:0] [--steps N]
        exits 0 with a message when `mujoco` is not importable.  The recorder is exercised end to end today by tests/test_mujoco_pin.py, which runs it
        against robosuite_amd.shim posing as `mujoco` (fp64 oracle backend): file layout, keys and the comparisons of that test stay alive.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
if "--out" in sys.argv:
    OUT = sys.argv[sys.argv.index("--out") + 1]
ONLY = sys.argv[sys.argv.index("--cases") + 1].split(",") if "--cases" in sys.argv else None      # "Env:seed" items
MAX_STEPS = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None

CASES = (  # (env, robots, controller type or None = robot default, seed, control steps, action scale): the tapes of tools/gen_golden.py
    ("Lift", "Panda", None, 0, 40, 0.1),
    ("Lift", "Panda", None, 1, 40, 1.0),
    ("Stack", "Panda", None, 0, 30, 1.0),
    ("PickPlace", "IIWA", None, 0, 20, 1.0),
)


def snapshot(sim):
    """Forward quantities of the CURRENT state, straight from mjData (after the env's last mj_step: one mj_forward re-evaluates them there)."""
    import mujoco

    m, d = sim.model._model, sim.data._data
    ws = np.array(d.qacc_warmstart)
    mujoco.mj_forward(m, d)
    qM = np.zeros((m.nv, m.nv))
    mujoco.mj_fullM(m, qM, d.qM)
    con = np.zeros((d.ncon, 16))
    for i in range(d.ncon):
        c = d.contact[i]
        con[i, 0] = c.dist; con[i, 1:4] = c.pos; con[i, 4:13] = np.asarray(c.frame); con[i, 13] = c.geom1; con[i, 14] = c.geom2; con[i, 15] = c.dim
    pairs = {}
    for r in con:
        pairs[(int(r[13]), int(r[14]))] = pairs.get((int(r[13]), int(r[14])), 0) + 1
    gt = np.asarray(m.geom_type)
    pair_counts = np.array([[g1, g2, int(gt[g1]), int(gt[g2]), n] for (g1, g2), n in sorted(pairs.items())], dtype=np.int64).reshape(-1, 5)
    return dict(ws=ws, qpos=np.array(d.qpos), qvel=np.array(d.qvel), ctrl=np.array(d.ctrl), qM=qM, qfrc_bias=np.array(d.qfrc_bias),
                qfrc_passive=np.array(d.qfrc_passive), qfrc_actuator=np.array(d.qfrc_actuator), qacc=np.array(d.qacc),
                qfrc_constraint=np.array(d.qfrc_constraint), ncon=int(d.ncon), contact=con, nefc=int(d.nefc),
                efc_force=np.array(d.efc_force[:d.nefc]), efc_aref=np.array(d.efc_aref[:d.nefc]), efc_R=np.array(d.efc_R[:d.nefc]),
                efc_type=np.array(d.efc_type[:d.nefc]), pair_counts=pair_counts)


def record(env_name, robot, ctype, seed, n_steps, scale):
    import robosuite as suite

    kw = dict(robots=robot, has_renderer=False, has_offscreen_renderer=False, use_camera_obs=False, use_object_obs=True, reward_shaping=True,
              control_freq=20, horizon=500, ignore_done=True, seed=seed)
    if ctype is not None:
        from robosuite.controllers import load_part_controller_config
        from robosuite.controllers.composite.composite_controller_factory import refactor_composite_controller_config

        kw["controller_configs"] = refactor_composite_controller_config(load_part_controller_config(default_controller=ctype), robot, ["right"])
    env = suite.make(env_name, **kw)
    env.reset()
    sim = env.sim
    rng = np.random.default_rng(10**6 + seed)                       # the action stream of tools/gen_golden.py
    states, actions, ctrls, rewards, snaps = [sim.get_state().flatten()], [], [], [], [snapshot(sim)]
    for t in range(n_steps):
        a = scale * rng.uniform(-1, 1, env.action_dim)
        _, r, _, _ = env.step(a)
        actions.append(a); states.append(sim.get_state().flatten()); ctrls.append(np.array(sim.data.ctrl)); rewards.append(r)
        if t % 5 == 4:
            snaps.append(snapshot(sim))
    out = dict(states=np.array(states), actions=np.array(actions), ctrl=np.array(ctrls), rewards=np.array(rewards), n_snap=len(snaps))
    for i, s in enumerate(snaps):
        for k, v in s.items():
            out[f"snap{i}_{k}"] = np.asarray(v)
    tag = f"mujoco_{env_name.lower()}_{robot.lower()}_seed{seed}"
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    with open(os.path.join(OUT, tag + ".xml"), "w") as f:
        f.write(sim.model.get_xml())
    print(tag, "control steps", n_steps, "snapshots", len(snaps), "max ncon", max(s["ncon"] for s in snaps))


if __name__ == "__main__":
    try:
        import mujoco  # noqa: F401  (a real wheel, or -- in the self-test -- robosuite_amd.shim registered as `mujoco` by the caller)
    except ImportError:
        print("gen_golden_with_mujoco: the `mujoco` wheel is not importable here -- nothing recorded (physics half of the oracle stays unpinned)")
        sys.exit(0)
    sys.path.insert(0, os.environ.get("ROBOSUITE_ROOT", "/root/reference"))
    os.makedirs(OUT, exist_ok=True)
    for case in CASES:
        if ONLY is not None and f"{case[0]}:{case[3]}" not in ONLY:
            continue
        if MAX_STEPS is not None:
            case = case[:4] + (min(case[4], MAX_STEPS),) + case[5:]
        record(*case)
