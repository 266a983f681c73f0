"""Replay single substeps saved by tools/trace_env_step.py through the kernel's forward() under the solver settings of the environment (RSIM_NEWTON_EXACT,
RSIM_NEWTON_WIDE, RSIM_NEWTON_NS / NA, RSIM_NEWTON_REFINE) and report the fp64 oracle's objective at the kernel's acceleration against its own optimum.
Usage (GPU box): [RSIM_NEWTON_EXACT=0] python tools/replay_substeps.py stack 871 1 2 3 12 13"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import mjcf
from tests.util import make_hip, make_oracle
task, e = sys.argv[1], int(sys.argv[2]); subs = [int(x) for x in sys.argv[3:]]
STEMS = {"lift": "lift_panda", "stack": "stack_panda", "peg": "peg_baxter_joint_velocity"}
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, STEMS[task] + ".rsim")); cfg = json.load(open(os.path.join(adir, STEMS[task] + ".cfg.json")))
z = np.load(next(p for p in (os.path.join(ROOT, d, f"trace_env_step_{task}_{e}.npz") for d in ("gpurun_out", os.path.join("profiles", "data"))) if os.path.exists(p)))
hm, hb = make_hip(flat, cfg, B=1)
om, od, _ = make_oracle(flat, cfg)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("RSIM_NEWTON_EXACT", "RSIM_NEWTON_WIDE", "RSIM_NEWTON_NS", "RSIM_NEWTON_NA", "RSIM_NEWTON_REFINE") if k in os.environ) or "defaults"
for s in subs:
    i = s - 1
    for f in ("qpos", "qvel", "qacc_warmstart", "ctrl"): hb.set(f, z[f][i][None])
    hb.forward()
    ka = hb.get("qacc")[0].astype(np.float64)
    od.qpos[:] = z["qpos"][i]; od.qvel[:] = z["qvel"][i]; od.qacc_warmstart[:] = z["qacc_warmstart"][i]; od.ctrl[:] = z["ctrl"][i]; od.forward()
    co = od.cost(od.qacc.copy()); qa0 = od.qacc.copy(); ck0 = od.cost(ka)
    # the same with the KERNEL's contact geometry handed to the oracle (distances, points, frames of its fp32 narrow phase): solver against solver on one problem
    kc = hb.contacts(0)
    geo = ""
    if len(kc) == od.ncon and od.forward_with_contact_geometry(kc):
        cg = od.cost(od.qacc.copy())
        geo = f" | with the kernel's contact geometry: {od.cost(ka) - cg:+.3e} (of {cg:.3e}), max |dqacc| {np.abs(ka - od.qacc).max():.2e}"
        oc_ = od.contacts()
    else:
        geo = f" | contact lists differ: kernel {len(kc)} oracle {od.ncon}"
    od.forward()
    oc0 = od.contacts()
    if len(kc) == len(oc0):
        dn = max(float(np.abs(np.asarray(a["frame"])[0] - np.asarray(b["frame"])[0]).max()) for a, b in zip(kc, oc0)) if kc else 0.0
        dp_ = max(float(np.abs(np.asarray(a["pos"]) - np.asarray(b["pos"])).max()) for a, b in zip(kc, oc0)) if kc else 0.0
        dd = max(abs(a["dist"] - b["dist"]) for a, b in zip(kc, oc0)) if kc else 0.0
        geo += f" | geometry kernel vs oracle: normal {dn:.1e} point {dp_:.1e} m depth {dd:.1e} m"
    print(f"[{tag}] env {e} substep {s}: kernel niter {int(hb.get('niter')[0])} oracle niter {od.solver_iter}; objective at the kernel's qacc - oracle's optimum {ck0 - co:+.3e} (of {co:.3e}); max |dqacc| {np.abs(ka - qa0).max():.2e}{geo}")
