"""Dump the PickPlace envs whose constraint forces / accelerations differ most from the oracle (fed the kernel's contact geometry) for analysis
without a GPU: state, the env's live model parameters, the kernel's contact list, forces and accelerations.  Same rollout as
tests/test_full_size_parity.py::test_pickplace_8192...  Usage (GPU box): python tools/pp_dump.py [n_worst=4] [B=8192] [dr=1]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, pick_place
from tests.util import load_golden
from tests.test_full_size_parity import PARAM_FIELDS, compare_reached_states, spread
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g, cfg, flat = load_golden("seed0_full", "pickplace_iiwa")
env = pick_place.PickPlaceBatch(flat, cfg, np.arange(B), seed0=0, horizon=500, bank_episodes=2, per_env_params=True)
b = env.batch
b.dr_save_defaults()
tape = torch.tensor(lift.env_actions(np.arange(B), 50), device="cuda")
for t in range(50):
    if dr: b.randomize_dynamics(seed=11, step=t)
    env.step(tape[t])
res = compare_reached_states(flat, b, spread(B, 32))
ok = [r for r in res if r["same"] and "g_force" in r]
ok.sort(key=lambda r: -r["g_force"] / max(1.0, r["g_fscale"]))
q, v, ws, ctrl = b.get("qpos"), b.get("qvel"), b.get("qacc_warmstart"), b.get("ctrl")
b.forward()
qacc, efc, nefc, niter = b.get("qacc"), b.get("efc_force"), b.get("nefc"), b.get("niter")
out = {}
for r in ok[:nw] + ok[-1:]:
    e = r["env"]
    print(f"env {e}: rel dforce {r['g_force'] / max(1.0, r['g_fscale']):.2e} (scale {r['g_fscale']:.3g}) rel dqacc {r['g_qacc'] / max(1.0, r['g_ascale']):.2e} ncon {r['ncon']} nefc {r['nefc']} newton iterations {niter[e]}")
    out[f"e{e}_state"] = np.concatenate([q[e], v[e], ws[e], ctrl[e]]).astype(np.float64)
    out[f"e{e}_qacc"], out[f"e{e}_efc"] = qacc[e].astype(np.float64), efc[e][:nefc[e]].astype(np.float64)
    hc = b.contacts(int(e))
    out[f"e{e}_con"] = np.array([[c["dist"], *c["pos"], *np.asarray(c["frame"]).ravel(), c["geom1"], c["geom2"], c["dim"]] for c in hc])
    for k in PARAM_FIELDS:
        if k in flat.arrays:
            out[f"e{e}_p_{k}"] = b.param_get(k, e, 1)[0]
    out[f"e{e}_p_opt"] = b.param_get("opt", e, 1)[0]
out["envs"] = np.array([r["env"] for r in ok[:nw] + ok[-1:]])
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "pp_dump.npz"), **out)
print("saved gpurun_out/pp_dump.npz")
