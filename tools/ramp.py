"""Per-launch duration over a long back-to-back run: shows the DVFS clock ramp after idle."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B = 4096; N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
if os.environ.get("RSIM_NOSCHED"): env.batch.set_schedule(False)
tape = torch.tensor(lift.env_actions(np.arange(B), 50), device="cuda")
stream = torch.cuda.ExternalStream(env.batch.stream())
time.sleep(2.0)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for t in range(N):
    ev[t][0].record(stream); env.step(tape[t % 50]); ev[t][1].record(stream)
env.batch.sync(); torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
print("per-launch ms:", " ".join(f"{x:.2f}" for x in ms[:12]), "...", " ".join(f"{x:.2f}" for x in ms[-6:]))
cum = np.cumsum(ms)
for k in (10, 25, 50, 100, 200, N):
    print(f"first {k:4d} launches: mean {ms[:k].mean():.2f} ms -> {B/ms[:k].mean()*1e3:.0f} env-steps/s   (elapsed {cum[k-1]:.0f} ms)")
print(f"last 50: mean {ms[-50:].mean():.2f} ms -> {B/ms[-50:].mean()*1e3:.0f} env-steps/s; diverged {int((~torch.isfinite(env.batch.tensor('qpos')).all(dim=1)).sum())}")
