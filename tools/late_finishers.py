"""Which envs finish last in a lockstep launch of the bench mix, and what the dispatch order (previous step's duration, longest first) had predicted for them:
per step the correlation of consecutive durations, the span against the work per slot, and the last finishers as [env, start us, duration, previous duration, start
rank, tier].  Usage (GPU box): python tools/late_finishers.py [lift|stack|peg]   -> profiles/r06_u_late_finishers_stack.txt"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from robosuite_amd import factory, lift, shard
cfgname = sys.argv[1] if len(sys.argv) > 1 else "stack"
label, stem, B, dr, which = bench.CONFIGS[cfgname]
flat, cfg = factory.load_shipped(stem)
ids = shard.env_block(B, 0, 1)
P = int(sys.argv[2]) if len(sys.argv) > 2 else 520
env = bench.build_env(cfgname, flat, cfg, ids, 0, 6)
tape = torch.tensor(lift.env_actions(ids, P + 8, action_dim=env.model.action_dim), device="cuda")
env.batch.set("ep_step", ((197 * ids) % 500).astype(np.int32))
for t in range(P): env.step(tape[t])
env.batch.sync(); env.batch.profile(True); env.batch.profile_env(-2)
prev = None
for t in range(P, P + 6):
    tier0 = env.batch.tier_snapshot()
    env.step(tape[t]); env.batch.sync()
    w = env.batch.wavelog()
    t0, t1 = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
    dur = (t1 - t0) / 100.0; st = (t0 - t0.min()) / 100.0; en = (t1 - t0.min()) / 100.0
    ok = dur > 600   # (the reset-observation pass that follows the step rewrites the records of the envs it touches: 45 - 400 us entries)
    if prev is not None:
        both = ok & (prev > 600)
        c = np.corrcoef(prev[both], dur[both])[0, 1]
        last = np.argsort(-np.where(ok, en, 0))[:12]
        rk = np.argsort(np.argsort(st, kind="stable"), kind="stable")
        print(f"step {t}: span {en[ok].max():.0f} us  corr(prev dur, dur) {c:.3f}  slowest {dur[ok].max():.0f}  sum/slots {dur[ok].sum() / {'lift': 2048, 'stack': 2048, 'peg': 1280, 'pickplace': 1024}[cfgname]:.0f}")
        print("   last finishers [env, start, dur, prev dur, start rank, tier before]:", [[int(e), int(st[e]), int(dur[e]), int(prev[e]), int(rk[e]), int(tier0[e])] for e in last])
    prev = np.where(ok, dur, 0)
