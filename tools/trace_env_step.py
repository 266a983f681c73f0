"""One control step of single envs, one substep at a time, kernel against oracle with TEACHER FORCING: the kernel's state after k substeps (control_step(a, k)
from the saved state: true controller semantics) -> ONE oracle step from that very state with the ctrl the kernel applied in substep k + 1 -> compared with the
kernel's state after k + 1 substeps.  Separates "the two sides compute a different substep" from "equal substeps, sensitive trajectory".
Input: gpurun_out/newton_hard_<task>.npz (tools/newton_hard_envs.py).  Usage (GPU box): python tools/trace_env_step.py stack 871 746 166"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import mjcf
from tests.util import make_hip, make_oracle
task = sys.argv[1]; want = [int(x) for x in sys.argv[2:]]
STEMS = {"lift": "lift_panda", "stack": "stack_panda", "peg": "peg_baxter_joint_velocity"}
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, STEMS[task] + ".rsim")); cfg = json.load(open(os.path.join(adir, STEMS[task] + ".cfg.json")))
z = np.load(next(p for p in (os.path.join(ROOT, d, f"newton_hard_{task}.npz") for d in ("gpurun_out", os.path.join("profiles", "data"))) if os.path.exists(p)))
n_sub = int(z["n_sub"])
hm, hb = make_hip(flat, cfg, B=1)
for e in want:
    i = int(np.nonzero(z["envs"] == e)[0][0])
    a1 = torch.tensor(z["actions"][i][None], device="cuda")

    def after(k):
        for f in ("qpos", "qvel", "qacc_warmstart", "ctrl", "cstate"): hb.set(f, z[f][i][None])
        if k: hb.control_step(a1, k)
        out = {f: hb.get(f)[0].astype(np.float64).copy() for f in ("qpos", "qvel", "qacc_warmstart", "ctrl")}
        return out
    om, od, _ = make_oracle(flat, cfg)
    prev = after(0)
    dump = []
    print(f"{task} env {e}: substep | one-substep deviation |dq| |dv| (kernel vs oracle from the kernel's own state) | kernel ncon nefc niter | oracle ncon nefc niter")
    for k in range(n_sub):
        cur = after(k + 1)
        kn = (int(hb.get("ncon")[0]), int(hb.get("nefc")[0]), int(hb.get("niter")[0]))     # of the forward pass behind the read: state after k + 1 substeps
        od.qpos[:] = prev["qpos"]; od.qvel[:] = prev["qvel"]; od.qacc_warmstart[:] = prev["qacc_warmstart"]; od.ctrl[:] = cur["ctrl"]
        # the kernel's acceleration of this very substep: forward() of the debug entry on the same inputs (same solver, same warm start)
        for f in ("qpos", "qvel", "qacc_warmstart"): hb.set(f, prev[f][None])
        hb.set("ctrl", cur["ctrl"][None]); hb.forward()
        ka = hb.get("qacc")[0].astype(np.float64).copy()
        od.forward()
        ck, co = od.cost(ka), od.cost(od.qacc.copy())
        dump.append(dict(qpos=prev["qpos"], qvel=prev["qvel"], qacc_warmstart=prev["qacc_warmstart"], ctrl=cur["ctrl"], qacc_kernel=ka, qacc_oracle=od.qacc.copy(), cost_kernel=ck, cost_oracle=co))
        od.step()
        dq, dv = np.abs(cur["qpos"] - od.qpos), np.abs(cur["qvel"] - od.qvel)
        flag = "   <--" if dq.max() > 1e-4 or dv.max() > 1e-2 else ""
        print(f"   {k + 1:2d} | {dq.max():.1e} (dof {int(dq.argmax())}) {dv.max():.1e} (dof {int(dv.argmax())}) | {kn} | {od.ncon} {od.nefc} {od.solver_iter} | oracle objective at the kernel's qacc - at its own: {ck - co:+.2e} (of {co:.3e}){flag}")
        prev = cur
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"trace_env_step_{task}_{e}.npz"), **{k: np.array([d[k] for d in dump]) for k in dump[0]})
    print(f"   whole step |dq| kernel vs saved kernel end state: {np.abs(prev['qpos'] - z['qpos1'][i]).max():.1e}")
