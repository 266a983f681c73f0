"""Stream groups (rsim_set_stream_groups): throughput of the bench workload by group count, and that the reached state does not depend on it.
Usage (GPU box): [GPU_MAX_HW_QUEUES=8] python tools/groups_sweep.py [steps] [G ...]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosuite_amd import lift, mjcf
B, H = 4096, 500
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Gs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16, 32]
adir = os.path.join(ROOT, "robosuite_amd", "assets")
flat = mjcf.load_model(os.path.join(adir, "lift_panda.rsim")); cfg = json.load(open(os.path.join(adir, "lift_panda.cfg.json")))
ids = np.arange(B)
tape = torch.tensor(lift.env_actions(ids, H + 20 + K), device="cuda")
ref = None
for G in Gs:
    env = lift.LiftBatch(flat, cfg, ids, seed0=0, horizon=H, bank_episodes=2 + (H + 20 + K) // H)
    env.batch.set_stream_groups(G)
    env.batch.set("ep_step", ((197 * ids) % H).astype(np.int32))
    for t in range(H + 20): env.step(tape[t])
    env.batch.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(K): env.step(tape[H + 20 + t])
    env.batch.sync(); dt = time.perf_counter() - t0
    q = env.batch.get("qpos"); r = env.batch.get("reward")
    if ref is None: ref = (q, r)
    same = np.array_equal(q, ref[0]) and np.array_equal(r, ref[1])
    print(f"groups {G:2d}: {1e3 * dt / K:.3f} ms/step -> {B * K / dt:,.0f} env-steps/s   state identical to the first run: {same}   bank_stale {int(env.batch.get('bank_stale').sum())}", flush=True)
    del env
