"""Separates the two reasons a long run slows down: clock settling (DVFS) vs the episode getting heavier (more contacts / MPR / Newton work).
Runs N launches, then resets every env to its initial state WITHOUT idling and runs 30 more: those 30 are early-episode work at settled clocks."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B = 4096; N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
env = lift.LiftBatch(flat, cfg, np.arange(B), seed0=0)
tape = torch.tensor(lift.env_actions(np.arange(B), N + 40), device="cuda")
stream = torch.cuda.ExternalStream(env.batch.stream())
time.sleep(2.0)
def run(t0, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for t in range(n):
        ev[t][0].record(stream); env.step(tape[t0 + t]); ev[t][1].record(stream)
    env.batch.sync(); torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ev])
a = run(0, N)
env.reset()          # host reset: a few ms, the GPU does not get to idle long
b = run(0, 30)
print(f"cold start, steps 0-9:      {a[:10].mean():.2f} ms")
print(f"steps {N-30}-{N-1} (late, settled): {a[-30:].mean():.2f} ms")
print(f"after reset, steps 0-9:     {b[:10].mean():.2f} ms   steps 20-29: {b[20:30].mean():.2f} ms")
print("late per-launch:", " ".join(f"{x:.2f}" for x in a[-8:]), "| after reset:", " ".join(f"{x:.2f}" for x in b[:12]))
