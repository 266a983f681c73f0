"""The fp32 step rule of the Newton iteration (RSIM_NEWTON_NS / NA, rsim_step.hip solve_newton) on the WIDE configurations (RSIM_NEWTON_WIDE=1): what it buys and
what it costs in how closely the fused control step tracks the fp64 oracle loop on the PickPlace / IIWA + Robotiq140 and Stack fixtures (20 control
steps, worst |dq| of the arm, the gripper's finger joints, the objects) -- round 3 found the Lift setting (1e-5, 1e-5) to cost a factor of ten on the
Robotiq's undamped 5e-5 kg m^2 finger links.  Throughput: lockstep control steps of 2048 PickPlace / 4096 Stack envs under full-range random actions.
Usage (GPU box): python tools/newton_wide_sweep.py "wide,ns,na" ...   (wide = 0: rule off)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, pick_place, stack
from tests.util import load_golden, make_hip, make_oracle
settings = [tuple(s.split(",")) for s in sys.argv[1:]] or [("0", "1e-5", "1e-5")]


def tracking(tag, model, groups):
    g, cfg, flat = load_golden(tag, model)
    nq = flat.nq
    om, od, oc = make_oracle(flat, cfg)
    hm, hb = make_hip(flat, cfg, B=2)
    s0 = g["states"][0]
    od.qpos[:] = s0[1:1 + nq]; od.qvel[:] = s0[1 + nq:]; od.qacc_warmstart[:] = 0; od.ctrl[:] = 0; od.forward(); oc.reset(od)
    hb.set("qpos", s0[1:1 + nq][None].repeat(2, 0)); hb.set("qvel", s0[1 + nq:][None].repeat(2, 0)); hb.set("qacc_warmstart", 0); hb.set("ctrl", 0)
    hb.forward(); hb.ctrl_reset()
    worst = {k: 0.0 for k in groups}
    for t in range(len(g["actions"])):
        hb.control_step(torch.tensor(np.repeat(g["actions"][t][None], 2, 0), dtype=torch.float32, device="cuda"), 25)
        oc.env_step(od, g["actions"][t], 25)
        dq = np.abs(hb.get("qpos")[0] - od.qpos)
        for k, idx in groups.items():
            worst[k] = max(worst[k], float(dq[idx].max()))
    return worst


def throughput(which, B, T0, T):
    if which == "pickplace":
        g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
        env = pick_place.PickPlaceBatch(flat, cfg, np.arange(B), seed0=0)
    else:
        g, cfg, flat = load_golden("seed0_full", "stack_panda")
        env = stack.StackBatch(flat, cfg, np.arange(B), seed0=0)
    tape = torch.tensor(lift.env_actions(np.arange(B), T0 + T, action_dim=env.model.action_dim), device="cuda")
    for t in range(T0): env.step(tape[t])
    env.batch.sync(); t0 = time.perf_counter()
    for t in range(T0, T0 + T): env.step(tape[t])
    env.batch.sync(); dt = time.perf_counter() - t0
    env.batch.profile(True); env.batch.profile_env(-1); env.step(tape[T0 + T - 1]); env.batch.sync(); p = env.batch.profile(False)
    return B * T / dt, p["n_newton"] / max(1, p["n_sub"]), p["n_ls"] / max(1, p["n_sub"]), int(env.batch.get("diverged").sum())


for wide, ns, na in settings:
    os.environ["RSIM_NEWTON_WIDE"], os.environ["RSIM_NEWTON_NS"], os.environ["RSIM_NEWTON_NA"] = wide, ns, na
    gpp, cpp, fpp = load_golden("seed0_full", "pickplace_iiwa")
    nq = fpp.nq
    arm = np.arange(7); fingers = np.arange(7, 13); objs = np.arange(13, nq)
    w1 = tracking("seed0_full", "pickplace_iiwa", dict(arm=arm, fingers=fingers, objects=objs))
    gs, cs, fs = load_golden("seed0_full", "stack_panda")
    w2 = tracking("seed0_full", "stack_panda", dict(arm=np.arange(7), fingers=np.arange(7, 9), objects=np.arange(9, fs.nq)))
    r1 = throughput("pickplace", 2048, 15, 12)
    r2 = throughput("stack", 4096, 30, 30)
    print(f"wide {wide} NS {ns} NA {na}: PickPlace {r1[0]:.0f} env-steps/s, newton {r1[1]:.2f} ls {r1[2]:.2f} per substep, diverged {r1[3]}; tracking |dq| arm {w1['arm']:.1e} fingers {w1['fingers']:.1e} objects {w1['objects']:.1e}"
          f" || Stack {r2[0]:.0f}, newton {r2[1]:.2f} ls {r2[2]:.2f}; tracking arm {w2['arm']:.1e} fingers {w2['fingers']:.1e} objects {w2['objects']:.1e}", flush=True)
