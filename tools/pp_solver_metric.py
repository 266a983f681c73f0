"""CPU analysis of gpurun_out/pp_dump.npz (tools/pp_dump.py): the PickPlace envs whose forces differ most from the oracle, looked at in the solver's
own metric.  For each env the oracle is fed the kernel's contact geometry; printed: the objective at the oracle's minimiser and at the kernel's
acceleration, the gradient left at the kernel's point (which dofs, their inertia), the largest acceleration difference.
Usage: python tools/pp_solver_metric.py [dump.npz]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import mjcf
from tests.util import load_golden
from tests.test_full_size_parity import PARAM_FIELDS
from oracle.oracle import OracleData, OracleModel
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pp_dump.npz"))
nq, nv = flat.nq, flat.nv
names = [None] * nv
for j in range(flat.njnt):
    for k in range(6 if flat.jnt_type[j] == 0 else 1):
        names[flat.jnt_dofadr[j] + k] = flat.names["joint"][j] + (f"[{k}]" if flat.jnt_type[j] == 0 else "")
for e in d["envs"]:
    f = flat.copy()
    for k in PARAM_FIELDS:
        if k in f.arrays: f.arrays[k] = d[f"e{e}_p_{k}"].reshape(f.arrays[k].shape).astype(np.float64)
    opt = d[f"e{e}_p_opt"]
    f.arrays["timestep"] = np.array([opt[0]]); f.arrays["gravity"] = opt[1:4].copy(); f.arrays["density"] = np.array([opt[4]])
    f.arrays["viscosity"] = np.array([opt[5]]); f.arrays["impratio"] = np.array([opt[6]]); f.arrays["wind"] = opt[7:10].copy()
    om = OracleModel(mjcf.to_blob(f)); od = OracleData(om)
    s = d[f"e{e}_state"]; od.qpos[:] = s[:nq]; od.qvel[:] = s[nq:nq + nv]; od.qacc_warmstart[:] = s[nq + nv:nq + 2 * nv]; od.ctrl[:] = s[nq + 2 * nv:]
    fed = od.forward_with_contact_geometry([dict(dist=c[0], pos=c[1:4], frame=c[4:13]) for c in d[f"e{e}_con"]])
    a_o, a_h = np.array(od.qacc).copy(), d[f"e{e}_qacc"]
    c_o, g_o = od.cost(a_o, True); c_h, g_h = od.cost(a_h, True)
    dq = a_h - a_o; i = int(np.argmax(np.abs(dq))); M = od.full_M()
    print(f"env {e}: geometry fed {fed}; oracle iterations {od.solver_iter}; objective at the oracle's minimiser {c_o:.6f}, at the kernel's acceleration {c_h:.6f}: "
          f"+{c_h - c_o:.3e} ({(c_h - c_o) / abs(c_o):.1e} relative); gradient left: oracle {np.abs(g_o).max():.1e}, kernel {np.abs(g_h).max():.2e} N m; "
          f"largest |dqacc| {abs(dq[i]):.3f} at {names[i]} (|a| {abs(a_o[i]):.1f}); largest smooth force {np.abs(od.qfrc_smooth).max():.1f}, largest constraint force {np.abs(od.efc_force).max():.1f}")
    print("    kernel gradient, largest components:", [(names[t], f"{g_h[t]:+.2e}", f"M_ii {M[t, t]:.1e}", f"dqacc {dq[t]:+.2f}") for t in np.argsort(-np.abs(g_h))[:4]])
