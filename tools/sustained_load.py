"""Does a stretch of sustained load slow the SAME work down?  (Round 5, session 16 / 17: the secondary regions of the default bench command read ~5 % lower behind the
25 s Lift headline than behind a 25-step one; the parent process was ruled out.)  Three identical Stack batches (same env ids, seeds and action tape = the identical
sequence of control steps) are stepped P + K times each: A after two seconds of idle, B right behind A, then L seconds of back-to-back Lift control steps, then C.
Per-launch durations from HIP events on the batch's stream.  Usage (GPU box): python tools/sustained_load.py [load seconds, default 12]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from robosuite_amd import factory, lift  # noqa: E402

L = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
P, K = 150, 50
dev = torch.device("cuda", 0); torch.cuda.set_device(0)


def make(config):
    label, stem, B, dr, which = bench.CONFIGS[config]
    flat, cfg = factory.load_shipped(stem)
    ids = np.arange(B)
    env = bench.build_env(config, flat, cfg, ids, 0, 3)
    env.batch.set("ep_step", ((197 * ids) % bench.HORIZON).astype(np.int32))
    tape = torch.tensor(lift.env_actions(ids, P + K, action_dim=env.model.action_dim), device=dev)
    return env, tape, torch.cuda.ExternalStream(env.batch.stream(), device=dev)


def run(env, tape, stream, n0, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for t in range(n):
        ev[t][0].record(stream); env.step(tape[n0 + t]); ev[t][1].record(stream)
    env.batch.sync(); torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ev])


envs = [make("stack") for _ in range(3)]
load_env, load_tape, load_stream = make("lift")
for e, _, _ in envs:
    e.bank_quiesce()
time.sleep(2.0)
out = {}
for name, (e, tape, s) in zip("AB", envs[:2]):
    d = run(e, tape, s, 0, P + K); out[name] = d
t0 = time.perf_counter(); n_load = 0
while time.perf_counter() - t0 < L:
    run(load_env, load_tape, load_stream, 0, P + K); n_load += P + K
d = run(envs[2][0], envs[2][1], envs[2][2], 0, P + K); out["C"] = d
for k, what in (("A", "after 2 s of idle"), ("B", "right behind A"), ("C", f"behind {L:.0f} s of Lift control steps ({n_load} launches)")):
    d = out[k]
    print(f"Stack batch {k} ({what}): steps 0-9 {d[:10].mean():.3f} ms, the {K} steps after {P}: {d[P:].mean():.3f} ms (median {np.median(d[P:]):.3f})")
same = all(np.array_equal(envs[0][0].batch.get("qpos"), e.batch.get("qpos")) for e, _, _ in envs[1:])
print("identical work (final qpos of A, B, C bitwise equal):", same)
print(f"C / B: {out['C'][P:].mean() / out['B'][P:].mean():.4f}   A / B: {out['A'][P:].mean() / out['B'][P:].mean():.4f}")
