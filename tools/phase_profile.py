"""Per-phase cycle breakdown of the fused control-step kernel on the BASELINE config-2 workload (rsim_profile)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim"))
cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
tape = torch.tensor(lift.env_actions(np.arange(B), steps + skip), device="cuda")
for t in range(skip):
    env.step(tape[t])
env.batch.sync()
env.batch.profile(True)
t0 = time.perf_counter()
for t in range(steps):
    env.step(tape[skip + t])
env.batch.sync()
dt = time.perf_counter() - t0
p = env.batch.profile(False)
nsub = max(1, p["n_sub"])
cyc = {k: v for k, v in p.items() if not k.startswith("n_") and k not in ("boxbox", "mpr", "plane")}
tot = sum(cyc.values())
print(f"B={B} steps={steps}: {1e3*dt/steps:.2f} ms/step -> {B*steps/dt:.0f} env-steps/s")
print(f"per env-substep: total {tot/nsub:.0f} cycles (s_memtime ticks, 100 MHz => {tot/nsub/100:.1f} us)")
for k, v in cyc.items():
    print(f"  {k:8s} {v/nsub:10.1f}  {100*v/tot:5.1f}%")
print(f"  inside narrow: boxbox {p['boxbox']/nsub:.0f} mpr {p['mpr']/nsub:.0f} other {p['plane']/nsub:.0f} cycles/substep")
for k in ("n_cand", "n_con", "n_efc", "n_newton", "n_ls", "n_boxbox", "n_mpr", "n_support"):
    print(f"  {k:8s} {p[k]/nsub:8.3f} per substep")
