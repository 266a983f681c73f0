"""Newton iterations of the fused kernel against the fp64 oracle on the SAME control step of the same envs.
GPU part (this file, on the box): Stack / Lift / Peg @B envs to control step `nskip`, one more step with the wave log on; the states before that step of the
K Newton-heaviest envs and of K envs spread over the batch, their actions and the kernel's iteration / line-search counts go to gpurun_out/newton_hard_<task>.npz.
CPU part (`--oracle`, build container): the oracle steps the same control step from those states and counts its own iterations per substep.
Usage: python tools/newton_hard_envs.py stack 300 4096 48        |        python tools/newton_hard_envs.py --oracle stack"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import mjcf
STEMS = {"lift": "lift_panda", "stack": "stack_panda", "peg": "peg_baxter_joint_velocity"}


def model(task):
    adir = os.path.join(ROOT, "robosuite_amd", "assets")
    return mjcf.load_model(os.path.join(adir, STEMS[task] + ".rsim")), json.load(open(os.path.join(adir, STEMS[task] + ".cfg.json")))


if sys.argv[1] == "--oracle":
    from tests.util import make_oracle
    task = sys.argv[2]
    flat, cfg = model(task)
    z = np.load(next(p for p in (os.path.join(ROOT, d, f"newton_hard_{task}.npz") for d in ("gpurun_out", os.path.join("profiles", "data"))) if os.path.exists(p)))
    n_sub = int(z["n_sub"])
    print(f"{task}: env, kernel newton / line-search evaluations per substep | oracle newton per substep (max over substeps), contacts / rows at the end | |dq| after the step")
    tot_k, tot_o = 0.0, 0.0
    for i, e in enumerate(z["envs"]):
        om, od, oc = make_oracle(flat, cfg)
        od.qpos[:] = z["qpos"][i]; od.qvel[:] = z["qvel"][i]; od.qacc_warmstart[:] = z["qacc_warmstart"][i]; od.ctrl[:] = z["ctrl"][i]
        od.forward(); oc.reset(od)
        st = oc.state
        cs = z["cstate"][i]
        st[:20] = cs[:20]; st[20:24] = cs[20:24]; st[24:28] = cs[20:24]
        its = []
        a = z["actions"][i].astype(np.float64)
        for s in range(n_sub):     # rso_env_step, one substep at a time: step1, set_goal on the first substep, controller, step2
            od.step1()
            if s == 0:
                oc.set_goal(od, a)
            oc.run(od)
            od.step2()
            its.append(int(od.solver_iter))
        kn, kl = z["newton"][i] / n_sub, z["ls"][i] / n_sub
        tot_k += kn; tot_o += np.mean(its)
        print(f"  env {int(e):5d}: kernel {kn:5.2f} / {kl:5.2f} | oracle {np.mean(its):5.2f} (max {max(its)}), {od.ncon} / {od.nefc} | {np.abs(z['qpos1'][i] - od.qpos).max():.1e}")
    print(f"mean newton per substep: kernel {tot_k / len(z['envs']):.2f}  oracle {tot_o / len(z['envs']):.2f}")
    sys.exit(0)

import torch
from robosuite_amd import lift, peg_in_hole, stack
task = sys.argv[1]; nskip = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096; K = int(sys.argv[4]) if len(sys.argv) > 4 else 48
flat, cfg = model(task)
CLS = {"lift": lift.LiftBatch, "stack": stack.StackBatch, "peg": peg_in_hole.PegBatch}[task]
env = CLS(flat, cfg, np.arange(B), seed0=0)
tape = torch.tensor(lift.env_actions(np.arange(B), nskip + 1, action_dim=env.model.action_dim), device="cuda")
env.batch.set("ep_step", ((197 * np.arange(B)) % 500).astype(np.int32))
for t in range(nskip): env.step(tape[t])
b = env.batch
b.sync()
pre = {k: b.get(k) for k in ("qpos", "qvel", "qacc_warmstart", "ctrl", "cstate")}
b.profile(True); b.profile_env(-2)
env.step(tape[nskip]); b.sync()
w = b.wavelog()
newton = w[:, 6].astype(np.int64)
p = b.profile(False)
pick = np.unique(np.concatenate([np.argsort(-newton)[:K], np.linspace(0, B - 1, K).astype(int)]))
# line-search evaluations are only in the shared accumulators: per-env values need one run per env; keep the batch mean instead
out = dict(envs=pick, n_sub=env.n_sub, newton=newton[pick], ls=np.full(len(pick), p.get("n_ls", 0) / B), actions=tape[nskip][pick].cpu().numpy(), qpos1=b.get("qpos")[pick],
           **{k: v[pick] for k, v in pre.items()})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(next(p for p in (os.path.join(ROOT, d, f"newton_hard_{task}.npz") for d in ("gpurun_out", os.path.join("profiles", "data"))) if os.path.exists(p)), **out)
print(f"{task}: step {nskip}, newton iterations per launch mean {newton.mean():.1f} p99 {np.percentile(newton, 99):.0f} max {newton.max()}; saved {len(pick)} envs")
